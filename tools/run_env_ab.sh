# per-kernel times of the one-stream step under environment settings, one board:  bash tools/run_env_ab.sh "A=1" "B=2 C=3" ...
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/envab; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
i=0
for e in "" "$@"; do i=$((i+1))
  ( [ -n "$e" ] && export $e
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r$i -o run -- python3 $R/bench.py --contexts 1 --streams 1 --steps 60 --warmup 10 --cpu-sample 0 --no-io --json-steps 0 --dropin-frames 0 --no-profile --no-accuracy-modes > $O/r$i.json 2> $O/r$i.err ) || { tail -5 $O/r$i.err; exit 1; }
  rm -f $O/r$i/run_kernel_trace.csv
  python3 - <<PY
import csv, json
d = json.load(open('$O/r$i.json'))
rows = list(csv.DictReader(open('$O/r$i/run_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('== [$e]: %.1f frames/s %.4f ms' % (d['value'], d['ms_per_step']))
for r in rows[:${TOP:-5}]:
    print('   %-58s calls %5s avg %8.1f us %5.2f%%' % (r['Name'][:58].replace('void mpe::', ''), r['Calls'], float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
PY
done
