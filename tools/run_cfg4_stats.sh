# kernel stats of the 23 x 10 shape in the configs[4] precision next to fp32 (one stream), and the JSON region's timeline inside bench.py
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c4; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for tag in fp32 cfg4; do
  extra=""; [ $tag = cfg4 ] && extra="--cfg4"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -o run -- python3 $R/bench.py --contexts 1 --streams 1 --preset RING23 --persons 10 --frames 96 --cpu-sample 0 --steps 6 --warmup 2 --json-steps 0 --no-io $extra > $O/bench_$tag.json 2> $O/$tag.err || { tail -3 $O/$tag.err; exit 1; }
  rm -f $O/$tag/run_kernel_trace.csv
done
python3 - <<PY
import csv
def load(tag):
    rows = list(csv.DictReader(open('$O/%s/run_kernel_stats.csv' % tag)))
    return {r['Name'][:90]: (int(r['Calls']), float(r['TotalDurationNs']) / 1e6) for r in rows}
a, b = load('fp32'), load('cfg4')
print('total ms: fp32 %.1f  cfg4 %.1f' % (sum(v[1] for v in a.values()), sum(v[1] for v in b.values())))
for k in sorted(set(a) | set(b), key=lambda k: -(a.get(k, (0, 0))[1] + b.get(k, (0, 0))[1]))[:18]:
    print('%-92s fp32 %4d x %8.1f us | cfg4 %4d x %8.1f us' % (k, a.get(k, (0, 0))[0], 1e3 * a.get(k, (0, 0))[1] / max(1, a.get(k, (0, 0))[0]), b.get(k, (0, 0))[0], 1e3 * b.get(k, (0, 0))[1] / max(1, b.get(k, (0, 0))[0])))
PY
cd $R
MPE_JSON_TIMING=1 timeout -k 10 300 python bench.py --steps 60 --warmup 10 --cpu-sample 0 --profile-steps 0 > $O/bench_json_timing.json 2> $O/bench_json_timing.err
grep -v amdgpu $O/bench_json_timing.err | tail -12
python3 -c "
import json; d=json.load(open('$O/bench_json_timing.json')); print('value', round(d['value']), 'json', round(d['json_inclusive']['value']))"
