"""rocprofv3 SQ-counter pass -> per-kernel summary (LDS bank-conflict share, MFMA busy share).

    rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE \
              --output-format csv -d DIR -o sq -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-io
    python tools/pmc_sq.py DIR/sq_counter_collection.csv out.json

lds_conflict_share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (extra LDS-array cycles / all LDS-array
cycles, MI355X_MICROARCH.md §LDS); mfma_busy_share = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8
XCDs x 1024 SIMDs)."""
import collections, csv, json, re, sys

tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r'\(.*', '', r['Kernel_Name']).strip()
    tot[name][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
        cnt[name] += 1
out = {}
for k, c in tot.items():
    n = max(1, cnt[k])
    e = {'launches': n}
    for name, v in c.items():
        e[name + '_per_launch'] = v / n
    if c.get('SQ_LDS_IDX_ACTIVE'):
        e['lds_conflict_share'] = c.get('SQ_LDS_BANK_CONFLICT', 0.0) / c['SQ_LDS_IDX_ACTIVE']
    if c.get('GRBM_GUI_ACTIVE') and 'SQ_VALU_MFMA_BUSY_CYCLES' in c:
        e['mfma_busy_share'] = c['SQ_VALU_MFMA_BUSY_CYCLES'] / (c['GRBM_GUI_ACTIVE'] / 8.0 * 1024.0)
    out[k] = e
json.dump(out, open(sys.argv[2], 'w'), indent=1)
for k in sorted(out, key=lambda k: -out[k].get('GRBM_GUI_ACTIVE_per_launch', 0) * out[k]['launches'])[:12]:
    e = out[k]
    print('%-62s n=%3d lds_conflict %s mfma_busy %s' % (k[:62], e['launches'], ('%.3f' % e['lds_conflict_share']) if 'lds_conflict_share' in e else '  -  ',
                                                     ('%.3f' % e['mfma_busy_share']) if 'mfma_busy_share' in e else '  -  '))
