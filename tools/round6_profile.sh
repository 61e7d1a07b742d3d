# The GPU-box command list behind profiles/r06_*.
#   bash tools/round6_profile.sh T    the GPU test suite + smoke
#   bash tools/round6_profile.sh F1 [tag]  the one-frame step: kernel stats + launch gaps of `bench.py --frames 1 --contexts 1 --streams 1`
#   bash tools/round6_profile.sh A    default bench plain, one stream, under rocprofv3 --kernel-trace --stats, the two PMC traffic passes
#   bash tools/round6_profile.sh B    the other shapes (tri, 5x10, configs[3] shard, 23x10, one frame)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O
cd $R
if [ "$1" = T ]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $O/gputest.log 2>&1; rc=$?; echo "pytest rc $rc" >> $O/gputest.log; grep -E "passed|failed|FAILED|rc" $O/gputest.log | tail -6
  [ $rc = 0 ] || exit 1
  timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
fi
show() { python3 -c "
import json,sys
d=json.load(open('$O/bench_$1.json')); r=d.get('roofline') or {}; s=r.get('step') or {}; dl=d.get('dropin_loop') or {}; j=d.get('json_inclusive') or {}; h=r.get('hbm') or {}
print('$1', round(d['value'],1), 'frames/s', round(d['ms_per_step'],4),'ms', 'min/max', d.get('value_min') and round(d['value_min'],1), d.get('value_max') and round(d['value_max'],1), 'host-to-host', d.get('value_host_to_host') and round(d['value_host_to_host'],1), 'json cold', j.get('value') and round(j['value'],1), 'gemm', r.get('frac') and round(r['frac'],4), 'step', s.get('frac') and round(s['frac'],4), 'hbm', h.get('frac') and round(h['frac'],4), 'dropin ms/frame', dl.get('ms_per_frame') and round(dl['ms_per_frame'],3), 'inside', dl.get('inside_mirrors_ms') and round(dl['inside_mirrors_ms'],3))
"; }
if [ "$1" = F1 ]; then
  TAG=${2:-base}
  X="--json-steps 0 --dropin-frames 0 --cpu-sample 0 --no-accuracy-modes"
  timeout -k 10 300 python bench.py --frames 1 --steps 400 --warmup 40 $X > $O/bench_1frame_$TAG.json 2> $O/bench_1frame_$TAG.err || { tail -5 $O/bench_1frame_$TAG.err; exit 1; }; show 1frame_$TAG
  timeout -k 10 300 python bench.py --frames 1 --contexts 1 --streams 1 --steps 400 --warmup 40 $X > $O/bench_1frame_s1_$TAG.json 2>> $O/bench_1frame_$TAG.err || exit 1; show 1frame_s1_$TAG
  timeout -k 10 300 python bench.py --frames 8 --contexts 1 --streams 1 --steps 400 --warmup 40 $X > $O/bench_8frame_s1_$TAG.json 2>> $O/bench_1frame_$TAG.err || exit 1; show 8frame_s1_$TAG
  cd /tmp; export TMPDIR=/tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_1f_$TAG -o run -- python3 $R/bench.py --frames 1 --contexts 1 --streams 1 --steps 200 --warmup 20 $X --no-io --no-profile > $O/bench_1frame_s1_${TAG}_under_rocprof.json 2> $O/stats_1f_$TAG.err; echo "stats rc $?"
  python3 $R/tools/launch_gaps.py $O/stats_1f_$TAG/run_kernel_trace.csv 200 > $O/1frame_${TAG}_launch_gaps.txt 2>&1; cat $O/1frame_${TAG}_launch_gaps.txt
  rm -f $O/stats_1f_$TAG/run_kernel_trace.csv
fi
if [ "$1" = A ]; then
  cd $R
  timeout -k 10 500 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -5 $O/bench_default.err; exit 1; }; show default
  timeout -k 10 300 python bench.py --contexts 1 --streams 1 --json-steps 0 --dropin-frames 0 > $O/bench_streams1.json 2>> $O/bench_default.err || exit 1; show streams1
  cd /tmp; export TMPDIR=/tmp
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -o run -- python3 $R/bench.py --contexts 1 --streams 1 --json-steps 0 --no-io --cpu-sample 0 --dropin-frames 0 --no-accuracy-modes > $O/bench_streams1_under_rocprof.json 2> $O/stats1.err; echo "stats (one stream) rc $?"
  rm -f $O/stats*/run_kernel_trace.csv
  for C in FETCH_SIZE WRITE_SIZE; do timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc -o $C -- python3 $R/bench.py --contexts 1 --streams 1 --steps 3 --warmup 1 --cpu-sample 0 --no-io --json-steps 0 --no-profile --dropin-frames 0 > /dev/null 2> $O/$C.err; echo "$C rc $?"; done
  python3 $R/tools/pmc_traffic.py $O/pmc/FETCH_SIZE_counter_collection.csv $O/pmc/WRITE_SIZE_counter_collection.csv $O/pmc_traffic.json
  rm -f $O/*/*_kernel_trace.csv
  python3 - <<PY
import csv, json
d = json.load(open('$O/bench_streams1_under_rocprof.json')); r = d['roofline']
rows = list(csv.DictReader(open('$O/stats1/run_kernel_stats.csv')))
g = [x for x in rows if 'k_linear_sb' in x['Name']]
print('under rocprof: value', round(d['value'], 1), '| live HIP events: split-bf16 GEMM avg launch', round(r['avg_launch_ms'], 5), 'ms | rocprof k_linear_sb* avg',
      round(sum(float(x['TotalDurationNs']) for x in g) / sum(int(x['Calls']) for x in g) / 1e6, 5), 'ms over', sum(int(x['Calls']) for x in g), 'launches')
PY
fi
if [ "$1" = B ]; then
  cd $R
  X="--json-steps 0 --dropin-frames 0"
  timeout -k 10 300 python bench.py --mode tri --cpu-sample 20 $X > $O/bench_tri.json 2> $O/bench_b.err; show tri
  timeout -k 10 300 python bench.py --persons 10 --frames 500 --cpu-sample 0 --steps 30 $X > $O/bench_5x10.json 2>> $O/bench_b.err; show 5x10
  timeout -k 10 300 python bench.py --persons 10 --total-frames 12500 --cpu-sample 0 --steps 5 --warmup 1 $X > $O/bench_c4_shard.json 2>> $O/bench_b.err; show c4_shard
  timeout -k 10 300 python bench.py --preset RING23 --persons 10 --frames 96 --cpu-sample 0 --steps 6 --warmup 2 $X > $O/bench_ring96.json 2>> $O/bench_b.err; show ring96
  timeout -k 10 300 python bench.py --frames 1 --cpu-sample 0 --steps 400 --warmup 40 $X > $O/bench_1frame.json 2>> $O/bench_b.err; show 1frame
fi
