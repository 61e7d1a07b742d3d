# throughput against the batch size (frames per step) on one board: is the 1000-frame batch still the best cut with the round-4 GEMMs?
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/bs; mkdir -p $O; cd $R
X="--json-steps 0 --dropin-frames 0 --no-io --cpu-sample 0 --no-profile"
for f in 125 250 500 1000 2000 4000; do
  for c in 1 2; do
    timeout -k 10 300 python bench.py $X --frames $f --contexts $c --streams 1 --steps $((300000 / f)) --warmup $((20000 / f)) > $O/b.json 2> $O/b.err || { tail -5 $O/b.err; exit 1; }
    python3 -c "
import json; d=json.load(open('$O/b.json')); print('frames $f contexts $c', round(d['value'],1), 'frames/s', round(d['ms_per_step'],4), 'ms')"
  done
done
