set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/gen; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/gputest.log 2>&1; echo "pytest rc $?"; tail -3 $O/gputest.log
timeout -k 10 300 python bench.py --persons 10 --frames 500 --cpu-sample 0 --steps 30 --no-io > $O/bench_5x10.json 2>> $O/err.log || exit 1
timeout -k 10 300 python bench.py --preset RING23 --persons 10 --frames 24 --cpu-sample 0 --steps 10 --warmup 2 --no-io > $O/bench_ring24.json 2>> $O/err.log || exit 1
timeout -k 10 300 python bench.py --preset RING23 --persons 10 --frames 96 --cpu-sample 0 --steps 6 --warmup 2 --no-io > $O/bench_ring96.json 2>> $O/err.log || exit 1
for f in 5x10 ring24 ring96; do python3 -c "
import json
d=json.load(open('$O/bench_$f.json')); print('$f', round(d['value'],1), round(d['ms_per_step'],3))"; done
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $R/bench.py --preset RING23 --persons 10 --frames 24 --cpu-sample 0 --steps 8 --warmup 1 --no-io > $O/bench_r.json 2> $O/stats.err; echo "stats rc $?"
rm -f $O/stats/run_kernel_trace.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$O/stats/run_kernel_stats.csv')))
for r in rows[:16]:
    print(r['Name'][:90].ljust(90), r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
PY
