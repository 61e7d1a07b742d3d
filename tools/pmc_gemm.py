"""rocprofv3 SQ-counter passes of the default bench -> stall / issue attribution of the GEMM launches.

    python tools/pmc_gemm.py OUT.json PASS1_counter_collection.csv [PASS2.csv ...]

Rows are keyed by (kernel instantiation, grid size): the eight MLP launches of a step share one
instantiation and differ in their grids.  All SQ_* wait / active / wave-cycle counters are in
quad-cycles summed over the waves (MI355X_MICROARCH.md, cycle-constants table), so their ratios
are dimensionless shares of the waves' lifetime:

    wait_any        SQ_WAIT_ANY / SQ_WAVE_CYCLES        parked in s_waitcnt / s_barrier
    wait_inst_any   SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES   issue stalls (MFMA RAW / pipe busy)
    wait_inst_lds   SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES   of which LDS-issue stalls
    active_any      SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES issuing
    mfma_busy       SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)   (share of the cycles the launch really had)
    eff_clock_ghz   GRBM_GUI_ACTIVE / 8 / duration: the clock the chip held during the launch (reads high on short launches)
    rounds          tiles / (256 CUs x workgroups per CU the kernel's LDS and registers allow)
"""
import collections
import csv
import json
import re
import sys


def short(name):
    m = re.match(r'(?:void )?(?:[A-Za-z0-9_]+::)*([A-Za-z0-9_]+(?:<[^>]*>)?)', name)   # any namespaces (mpe::, mpe::sb::)
    return m.group(1) if m else name[:60]


def main():
    out_path, files = sys.argv[1], sys.argv[2:]
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(lambda: collections.defaultdict(set))
    meta = {}
    dur = collections.defaultdict(dict)
    for f in files:
        for r in csv.DictReader(open(f)):
            name = short(r['Kernel_Name'])
            if 'k_linear' not in name and 'k_gat_fused' not in name and 'k_mlp_rows' not in name:
                continue
            key = '%s grid=%d' % (name, int(r['Grid_Size']) // max(1, int(r['Workgroup_Size'])))
            tot[key][r['Counter_Name']] += float(r['Counter_Value'])
            launches[key][r['Counter_Name']].add((f, r['Dispatch_Id']))
            meta[key] = {'workgroups': int(r['Grid_Size']) // max(1, int(r['Workgroup_Size'])),
                         'lds_bytes': int(r['LDS_Block_Size']), 'vgpr': int(r['VGPR_Count']),
                         'agpr': int(r['Accum_VGPR_Count']), 'sgpr': int(r['SGPR_Count'])}
            dur[key][(f, r['Dispatch_Id'])] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    out = {}
    for key, c in tot.items():
        e = dict(meta[key])
        per = {k: v / max(1, len(launches[key][k])) for k, v in c.items()}
        e['launches_seen'] = max(len(s) for s in launches[key].values())
        e['avg_us_under_pmc'] = sum(dur[key].values()) / len(dur[key]) / 1e3
        e['per_launch'] = per
        wc = per.get('SQ_WAVE_CYCLES')
        if wc:
            for nm, cn in (('wait_any', 'SQ_WAIT_ANY'), ('wait_inst_any', 'SQ_WAIT_INST_ANY'),
                           ('wait_inst_lds', 'SQ_WAIT_INST_LDS'), ('active_any', 'SQ_ACTIVE_INST_ANY'),
                           ('active_valu', 'SQ_ACTIVE_INST_VALU'), ('active_lds', 'SQ_ACTIVE_INST_LDS'),
                           ('active_vmem', 'SQ_ACTIVE_INST_VMEM')):
                if cn in per:
                    e[nm] = per[cn] / wc
        if per.get('GRBM_GUI_ACTIVE'):
            # effective clock of the dispatch (MI355X_MICROARCH.md, DVFS give-back): GRBM_GUI_ACTIVE is summed over the 8 XCDs; the
            # quotient reads HIGH on dispatches shorter than ~0.3 ms (the in-kernel stamps of tools/sb_clock_probe.py are the reference)
            e['eff_clock_ghz'] = per['GRBM_GUI_ACTIVE'] / 8.0 / (e['avg_us_under_pmc'] * 1e3)
        if per.get('GRBM_GUI_ACTIVE') and 'SQ_VALU_MFMA_BUSY_CYCLES' in per:
            e['mfma_busy'] = per['SQ_VALU_MFMA_BUSY_CYCLES'] / (per['GRBM_GUI_ACTIVE'] / 8.0 * 1024.0)
        if per.get('SQ_VALU_MFMA_BUSY_CYCLES') and 'SQ_VALU_MFMA_COEXEC_CYCLES' in per:
            e['valu_coexec_of_mfma_busy'] = per['SQ_VALU_MFMA_COEXEC_CYCLES'] / per['SQ_VALU_MFMA_BUSY_CYCLES']
        if per.get('SQ_WAVES'):
            w = per['SQ_WAVES']
            for cn in ('SQ_INSTS_VALU', 'SQ_INSTS_MFMA', 'SQ_INSTS_LDS', 'SQ_INSTS_SALU', 'SQ_INSTS_VMEM_RD',
                       'SQ_INSTS_VMEM_WR', 'SQ_INSTS_VALU_MFMA_MOPS_F32', 'SQ_INSTS_VALU_ADD_F64', 'SQ_INSTS_VALU_CVT'):
                if cn in per:
                    e[cn.lower() + '_per_wave'] = per[cn] / w
        # occupancy the launch can have: LDS (160 KiB / CU) and registers (512 per SIMD lane, 4-wave workgroups)
        regs = (e['vgpr'] + e['agpr'] + 7) // 8 * 8
        by_regs = min(8, 512 // max(8, regs))
        by_lds = 8 if not e['lds_bytes'] else (160 * 1024) // e['lds_bytes']
        e['wg_per_cu'] = min(by_regs, by_lds, 8)
        e['rounds'] = e['workgroups'] / (256.0 * e['wg_per_cu'])
        out[key] = e
    json.dump(out, open(out_path, 'w'), indent=1, sort_keys=True)
    cols = ('rounds', 'avg_us_under_pmc', 'eff_clock_ghz', 'mfma_busy', 'wait_any', 'wait_inst_any', 'wait_inst_lds', 'active_any')
    print('%-46s %s' % ('kernel, grid', ' '.join('%13s' % c for c in cols)))
    for key in sorted(out, key=lambda k: -out[k]['avg_us_under_pmc'] * out[k]['launches_seen']):
        e = out[key]
        print('%-46s %s' % (key[:46], ' '.join('%13s' % (('%.3f' % e[c]) if c in e else '-') for c in cols)))


if __name__ == '__main__':
    main()
