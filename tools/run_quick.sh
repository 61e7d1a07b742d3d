set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/quick; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "${1:-tri or mlp or golden or dropin}" > $O/gputest.log 2>&1; echo "pytest rc $?"; tail -3 $O/gputest.log
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $R/bench.py --steps 60 --warmup 10 --cpu-sample 0 --no-io > $O/bench.json 2> $O/stats.err; echo "stats rc $?"
rm -f $O/stats/run_kernel_trace.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$O/stats/run_kernel_stats.csv')))
for r in rows[:16]:
    print(r['Name'][:90].ljust(90), r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
PY
python3 -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'])"
