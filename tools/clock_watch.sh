# shader clock and socket power while the default bench runs (3000 steps = 17 s): is the GEMM power-limited?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/clk; mkdir -p $O
cd $R
python bench.py --steps 3000 --warmup 20 --cpu-sample 0 --no-io > $O/bench.json 2> $O/bench.err &
BP=$!
for i in $(seq 1 40); do
  printf "%s " "$(date +%S.%N | cut -c1-5)"
  /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket" | sed 's/.*: //' | tr '\n' ' '; echo
  kill -0 $BP 2>/dev/null || break
  sleep 0.7
done
wait $BP
python3 -c "
import json
d=json.load(open('$O/bench.json')); print(round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['achieved'],1))"
