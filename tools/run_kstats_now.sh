# kernel stats of the default bench, one stream (so that per-kernel durations add up to the step): -> gpurun_out/ks/
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ks; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $R/bench.py --contexts 1 --streams 1 --steps 100 --warmup 10 --cpu-sample 0 --no-io --json-steps 0 > $O/bench_s1_under_rocprof.json 2> $O/stats.err || { tail -5 $O/stats.err; exit 1; }
rm -f $O/stats/run_kernel_trace.csv
python3 - <<PY
import csv, json
d = json.load(open('$O/bench_s1_under_rocprof.json'))
print('value', round(d['value'], 1), 'ms/step', round(d['ms_per_step'], 4), 'gemm frac', round(d['roofline']['frac'], 4))
rows = list(csv.DictReader(open('$O/stats/run_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:22]:
    print('%6.2f%% %8d calls %9.1f us avg  %s' % (100 * float(r['TotalDurationNs']) / tot, int(r['Calls']), float(r['TotalDurationNs']) / int(r['Calls']) / 1e3, r['Name'][:150]))
PY
