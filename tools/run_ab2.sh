# A/B of VAR=a vs VAR=b on the default bench: bash tools/run_ab2.sh VAR a b [extra bench args]
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab; mkdir -p $O
V=$1; A=$2; B=$3; shift 3
cd $R
for x in $A $B $A $B; do
  env $V=$x python bench.py --steps 150 --warmup 15 --cpu-sample 0 --no-io "$@" > $O/b$x.json 2>/dev/null || exit 1
  python3 -c "
import json
d=json.load(open('$O/b$x.json')); print('$V=$x', round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline']['achieved'],2))"
done
