# per-kernel average durations of the default bench under rocprofv3, once per given environment setting:
#   bash tools/run_kstats_env.sh "MPE_GEMM_LOADER=0" "MPE_GEMM_LOADER=4"
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/kstats; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
i=0
for E in "$@"; do i=$((i+1))
  rm -rf $O/s$i
  env $E timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s$i -o run -- python3 $R/bench.py --steps 40 --warmup 5 --cpu-sample 0 --no-io --no-profile > $O/bench$i.json 2> $O/s$i.err || { tail -3 $O/s$i.err; exit 1; }
  rm -f $O/s$i/run_kernel_trace.csv
  echo "== $E"
  python3 - <<PY
import csv, json, re
rows=list(csv.DictReader(open('$O/s$i/run_kernel_stats.csv')))
tot=0
for r in rows[:14]:
    n=re.sub(r'\(.*','',r['Name']).replace('void mpe::','').replace('mpe::','')
    print('  %-46s calls %5s avg %8.1f us  %5s %%' % (n[:46], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
d=json.load(open('$O/bench$i.json')); print('  bench under rocprof: %.1f frames/s %.3f ms' % (d['value'], d['ms_per_step']))
PY
done
