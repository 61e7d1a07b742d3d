# single frame per call (online use) for the product library and variants of it, one context one stream (latency) and default
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/f1; mkdir -p $O; cd $R
X="--json-steps 0 --dropin-frames 0 --no-io --cpu-sample 0 --no-profile --frames 1 --steps 400 --warmup 40"
for rep in 1 2; do for v in "" $1; do
  for m in "--contexts 1 --streams 1" ""; do
    ( [ -n "$v" ] && export MPE_LIB_VARIANT=$v; timeout -k 10 200 python bench.py $X $m > $O/b.json 2> $O/b.err ) || { tail -3 $O/b.err; exit 1; }
    python3 -c "
import json; d=json.load(open('$O/b.json')); print('lib=${v:-product} [$m]', round(d['ms_per_step'],4), 'ms per step')"
  done; done; done
