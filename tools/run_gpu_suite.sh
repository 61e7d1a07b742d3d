# the whole GPU suite, then the default bench (what the driver runs at round end):  bash tools/run_gpu_suite.sh [tag]
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/suite; mkdir -p $O
T=${1:-now}
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputest_$T.log 2>&1; rc=$?
tail -15 $O/gputest_$T.log
[ $rc = 0 ] || exit $rc
timeout -k 10 300 python bench.py > $O/bench_$T.json 2> $O/bench_$T.err || { tail -5 $O/bench_$T.err; exit 1; }
python3 - <<PY
import json
d = json.load(open('$O/bench_$T.json'))
r = d['roofline']; p = d['parity']
print('value', round(d['value']), 'host_to_host', d['value_host_to_host'] and round(d['value_host_to_host']), 'json cold', d['value_json_cold'] and round(d['value_json_cold']),
      'roofline', round(r['frac'], 4), 'step', round(r['step']['frac'], 4))
print('parity', {k: p[k] for k in ('clusters_exact_frac', 'max_abs_mm', 'gpu_vs_exact_mm', 'ref_vs_exact_mm', 'delta_mpjpe_mm')})
print('max accuracy', p.get('mlp_max_accuracy'))
print('dropin', {k: v for k, v in (d.get('dropin_loop') or {}).items() if k != 'what'})
PY
