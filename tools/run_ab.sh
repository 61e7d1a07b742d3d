# A/B of one environment switch on the default bench: bash tools/run_ab.sh VAR [extra bench args]
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab; mkdir -p $O
V=$1; shift
cd $R
for x in 0 1 0 1; do
  env $V=$x python bench.py --steps 150 --warmup 15 --cpu-sample 0 --no-io "$@" > $O/b$x.json 2>/dev/null || exit 1
  python3 -c "
import json
d=json.load(open('$O/b$x.json')); print('$V=$x', round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['achieved'],1), (d['parity'] or {}).get('max_abs_score_diff'))"
done
