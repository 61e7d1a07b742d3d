"""rocprofv3 PMC passes -> profiles/r01_pmc_traffic.json (fabric bytes per launch and kernel).

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d DIR -o fetch -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d DIR -o write -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0
    python tools/pmc_traffic.py DIR/fetch_counter_collection.csv DIR/write_counter_collection.csv out.json

Units and corrections as MI355X_MICROARCH.md prescribes: FETCH_SIZE / WRITE_SIZE count KiB;
FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950; Infinity-Cache hits are included,
so the figure is fabric traffic, an upper bound on HBM traffic."""
import collections, csv, json, re, sys


def per_kernel(path, counter):
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        name = re.sub(r'\(.*', '', r['Kernel_Name'].replace('(anonymous namespace)::', '')).strip()
        tot[name] += float(r['Counter_Value'])
        cnt[name] += 1
    return tot, cnt


fetch, n_f = per_kernel(sys.argv[1], 'FETCH_SIZE')
write, n_w = per_kernel(sys.argv[2], 'WRITE_SIZE')
out = {'command': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0',
       'note': 'FETCH_SIZE/WRITE_SIZE are KiB; corrected bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE x2); Infinity-Cache hits are counted: fabric traffic, an upper bound on HBM traffic',
       'kernels': {}}
lin_bytes, lin_n = 0.0, 0
for k in fetch:
    n = n_f[k]
    f = fetch[k] / n
    w = write.get(k, 0.0) / max(1, n_w.get(k, 0))
    b = (2 * f + w) * 1024
    out['kernels'][k] = {'launches': n, 'fetch_kib_raw_per_launch': f, 'write_kib_per_launch': w, 'bytes_corrected_per_launch': b}
    if 'k_linear' in k:
        lin_bytes += b * n
        lin_n += n
# the dominant kernel since round 4: the split-bf16 tile kernel (csrc/gemm_sb16.hip)
sb = [(k, v) for k, v in out['kernels'].items() if 'k_linear_sb<' in k]
if sb:
    out['k_linear_sb_bytes_per_launch'] = sum(v['bytes_corrected_per_launch'] * v['launches'] for _, v in sb) / sum(v['launches'] for _, v in sb)
out['k_linear_bytes_per_launch'] = lin_bytes / max(1, lin_n)
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print('k_linear launches %d, corrected bytes per launch %.1f MB' % (lin_n, out['k_linear_bytes_per_launch'] / 1e6))
