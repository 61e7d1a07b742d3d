# engine modes of the default bench under an environment setting, one board: bash tools/run_modes_env.sh "A=1" "A=0"
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
X="--json-steps 0 --dropin-frames 0 --no-io --cpu-sample 0 --no-profile"
for rep in 1 2; do for e in "$@"; do
for m in "--contexts 1 --streams 1" "--contexts 2" "--contexts 3"; do
  ( export $e; timeout -k 10 300 python bench.py $X $m > $O/modes.json 2> $O/modes.err ) || { tail -5 $O/modes.err; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/modes.json')); print('[$e] [$m]', round(d['value'],1), 'frames/s', round(d['ms_per_step'],4), 'ms')"
done; done; done
