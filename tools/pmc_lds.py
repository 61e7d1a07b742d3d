"""Derived view of the round-4 counter passes (tools/round4_profile.sh C -> pmc_gemm.py json): is the LDS pipe or the
texture path what the GEMM's MFMA waves wait for?

    python tools/pmc_lds.py gpurun_out/r4/pmc_lds.json > profiles/r04_pmc_lds.txt

Normalisation (MI355X_MICROARCH.md): GRBM_GUI_ACTIVE is summed over the 8 XCDs, so cycles of the launch = GUI / 8;
SQ_LDS_IDX_ACTIVE counts LDS-array cycles summed over all CUs (checked below against the analytic count of the kernel's
ds_read_b128: 4 array cycles per wave-instruction), TA_/TD_/TCP_*_sum are summed over the 256 per-CU instances, so
x / 256 / (GUI / 8) is the share of the launch during which a CU's unit was in that state.
"""
import json
import sys


def main():
    d = json.load(open(sys.argv[1]))
    cols = ('lds_array', 'lds_data_fifo_full', 'lds_cmd_fifo_full', 'ta_busy', 'td_busy', 'td_stalled_by_tc', 'tcp_pending',
            'ta_addr_stalled_by_tc', 'ta_data_stalled_by_tc', 'mfma_busy', 'wait_any', 'wait_inst_any')
    print('%-48s %8s %8s ' % ('kernel, workgroups', 'us', 'rounds') + ' '.join('%10s' % c[:10] for c in cols))
    out = {}
    for k, e in sorted(d.items(), key=lambda kv: -kv[1]['avg_us_under_pmc'] * kv[1]['launches_seen']):
        if 'k_linear' not in k:
            continue
        p = e['per_launch']
        cyc = p['GRBM_GUI_ACTIVE'] / 8.0
        wc = p.get('SQ_WAVE_CYCLES')

        def cu(n):
            return p[n] / 256.0 / cyc if n in p else float('nan')
        row = {'lds_array': cu('SQ_LDS_IDX_ACTIVE'),
               'lds_data_fifo_full': p.get('SQ_LDS_DATA_FIFO_FULL', float('nan')) / wc, 'lds_cmd_fifo_full': p.get('SQ_LDS_CMD_FIFO_FULL', float('nan')) / wc,
               'ta_busy': cu('TA_TA_BUSY_sum'), 'td_busy': cu('TD_TD_BUSY_sum'), 'td_stalled_by_tc': cu('TD_TC_STALL_sum'),
               'tcp_pending': cu('TCP_PENDING_STALL_CYCLES_sum'), 'ta_addr_stalled_by_tc': cu('TA_ADDR_STALLED_BY_TC_CYCLES_sum'),
               'ta_data_stalled_by_tc': cu('TA_DATA_STALLED_BY_TC_CYCLES_sum'), 'mfma_busy': e.get('mfma_busy', float('nan')),
               'wait_any': e.get('wait_any', float('nan')), 'wait_inst_any': e.get('wait_inst_any', float('nan')),
               'lds_bank_conflict_cycles': p.get('SQ_LDS_BANK_CONFLICT'), 'lds_idx_active_cycles': p.get('SQ_LDS_IDX_ACTIVE'),
               'ds_read_wave_insts': p.get('SQ_INSTS_LDS_LOAD'), 'lds_dma_wave_insts': p.get('TA_FLAT_READ_LDS_WAVEFRONTS_sum')}
        out[k] = row
        print('%-48s %8.1f %8.2f ' % (k[:48], e['avg_us_under_pmc'], e['rounds']) + ' '.join('%10.4f' % row[c] for c in cols))
    print()
    print('check of the unit: SQ_LDS_IDX_ACTIVE / SQ_INSTS_LDS_LOAD (array cycles per ds_read wave-instruction; ds_read_b128 = 4):')
    for k, r in out.items():
        if r['ds_read_wave_insts']:
            print('  %-48s %.3f' % (k[:48], r['lds_idx_active_cycles'] / r['ds_read_wave_insts']))
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
