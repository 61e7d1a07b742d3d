set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s5x10; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $R/bench.py --persons 10 --frames 500 --cpu-sample 0 --steps 11 --warmup 1 --no-io > $O/bench.json 2> $O/stats.err; echo "stats rc $?"
rm -f $O/stats/run_kernel_trace.csv
python3 - <<PY
import csv,json
rows=list(csv.DictReader(open('$O/stats/run_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
with open('$O/kernel_stats.txt','w') as f:
    f.write('rocprofv3 --kernel-trace --stats -- python3 bench.py --persons 10 --frames 500 --cpu-sample 0 --steps 11 --warmup 1 --no-io  (13 steps incl. initialisation and warm-up)\n')
    f.write('%-72s %7s %10s %10s %6s\n' % ('kernel','calls','avg us','min us','%'))
    for r in rows[:22]:
        f.write('%-72s %7s %10.1f %10.1f %6.1f\n' % (r['Name'][:72], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['Percentage'])))
    f.write('kernel time per step: %.3f ms\n' % (tot/1e6/13))
print(open('$O/kernel_stats.txt').read())
d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'])
PY
