set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/def; mkdir -p $O
cd $R
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/err.log || exit 1
timeout -k 10 300 python bench.py --profile-every 1 --cpu-sample 0 > $O/bench_every1.json 2>> $O/err.log || exit 1
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $R/bench.py > $O/bench_under_rocprof.json 2> $O/stats.err; echo "stats rc $?"
rm -f $O/stats/run_kernel_trace.csv
python3 - <<PY
import csv,json
for f in ('bench_default','bench_every1','bench_under_rocprof'):
    d=json.load(open('$O/%s.json'%f)); r=d['roofline']
    print(f, round(d['value'],1), round(d['ms_per_step'],4), 'io', round(d['io_inclusive']['value'],1) if d['io_inclusive'] else None, 'gemm', round(r['achieved'],2), round(r['frac'],4), r['launches'], round(r['avg_launch_ms'],5), r.get('sampled_steps'), round(r['gemm_share_of_step'],3))
rows=list(csv.DictReader(open('$O/stats/run_kernel_stats.csv')))
g=[r for r in rows if 'k_linear_dma' in r['Name']]
print('rocprof gemm launches', sum(int(r['Calls']) for r in g), 'avg ms', sum(float(r['TotalDurationNs']) for r in g)/sum(int(r['Calls']) for r in g)/1e6)
PY
