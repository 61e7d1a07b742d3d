# Per-kernel times of the one-stream step for the product library and experiment builds of it (csrc: make exp EXPFLAGS=-DMPE_EXP=n,
# copied to ../libmpe_hip_<name>.so), one after the other on one board:
#   bash tools/run_variants.sh "'' e31 e32" [extra bench args]
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/var; mkdir -p $O
V="$1"; shift
cd /tmp; export TMPDIR=/tmp
for v in $V; do
  [ "$v" = "''" ] && v=""
  export MPE_LIB_VARIANT=$v
  [ -z "$v" ] && unset MPE_LIB_VARIANT
  n=${v:-product}
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -o run -- python3 $R/bench.py --contexts 1 --streams 1 --steps 60 --warmup 10 --cpu-sample 0 --no-io --json-steps 0 --dropin-frames 0 --no-profile --no-accuracy-modes "$@" > $O/$n.json 2> $O/$n.err || { tail -5 $O/$n.err; exit 1; }
  python3 - <<PY
import csv, collections
# per (kernel, grid): the launches of one instantiation differ in their shapes (GRID=1 to see them)
if '${GRID:-0}' == '1':
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open('$O/$n/run_kernel_trace.csv')):
        if 'k_linear' in r['Kernel_Name']:
            acc[(r['Kernel_Name'].split('(')[0].replace('void mpe::', ''), int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])))].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    for (k, g), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        v = v[len(v) // 5:]
        print('   grid %6d  %-50s launches %4d avg %8.1f us min %8.1f' % (g, k[:50], len(v), sum(v) / len(v) / 1e3, min(v) / 1e3))
PY
  rm -f $O/$n/run_kernel_trace.csv
  python3 - <<PY
import csv, json
d = json.load(open('$O/$n.json'))
rows = list(csv.DictReader(open('$O/$n/run_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('== $n: %.1f frames/s %.4f ms' % (d['value'], d['ms_per_step']))
for r in rows[:${TOP:-9}]:
    print('   %-58s calls %5s avg %8.1f us %5.2f%%' % (r['Name'][:58].replace('void mpe::', ''), r['Calls'], float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
PY
done
