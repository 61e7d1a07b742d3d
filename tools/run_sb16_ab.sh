# A/B of the MLP modes inside the default bench on one board, then the per-kernel times of the split mode
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
X="--json-steps 0 --dropin-frames 0 --no-io --cpu-sample 0"
for m in "" "--mlp-split" "" "--mlp-split"; do
  timeout -k 10 300 python bench.py $X $m > $O/ab_mlp.json 2> $O/ab_mlp.err || { tail -5 $O/ab_mlp.err; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/ab_mlp.json')); print('mode [$m]', round(d['value'],1), 'frames/s', round(d['ms_per_step'],4), 'ms')"
done
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_split -o run -- python3 $R/bench.py --contexts 1 --streams 1 $X --mlp-split > /dev/null 2> $O/stats_split.err; echo "stats rc $?"
rm -f $O/stats_split/run_kernel_trace.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$O/stats_split/run_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:12]:
    print('%-64s calls %5s avg %8.1f us %5.2f%%' % (r['Name'][:64].replace('void mpe::',''), r['Calls'], float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
PY
