# A/B of the product library against an experiment build (csrc: make exp EXPFLAGS=...), interleaved
# rounds in separate processes on one board: bash tools/run_ab_lib.sh [extra bench args]
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab; mkdir -p $O
cd $R
for x in "" exp "" exp; do
  env MPE_LIB_VARIANT=$x python bench.py --steps 150 --warmup 15 --cpu-sample 0 --no-io "$@" > $O/lib_$x.json 2>$O/lib_$x.err || { tail -3 $O/lib_$x.err; exit 1; }
  python3 -c "
import json
d=json.load(open('$O/lib_$x.json')); print('lib=${x:-product}', round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline']['achieved'],2))"
done
