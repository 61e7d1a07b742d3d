# A/B of two environment settings on the default bench, interleaved rounds on one board:
#   bash tools/run_ab_env.sh "A=1 B=2" "A=0" [rounds]
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab; mkdir -p $O
cd $R
N=${3:-2}
for r in $(seq $N); do for E in "$1" "$2"; do
  env $E python bench.py --steps 150 --warmup 15 --cpu-sample 0 --no-io $BENCH_ARGS > $O/env.json 2>$O/env.err || { tail -3 $O/env.err; exit 1; }
  python3 -c "
import json
d=json.load(open('$O/env.json')); print('%-40s' % '$E', round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline']['achieved'],2))"
done; done
