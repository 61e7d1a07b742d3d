# The GPU-box command list behind profiles/r03_*.  Two calls (gpurun's limit is 20 minutes each):
#   bash tools/round3_profile.sh A   tests, smoke, default bench (two contexts / one context on one stream) plain and under rocprofv3
#                                    --kernel-trace --stats, the two PMC traffic passes, the SQ passes of the GEMM
#   bash tools/round3_profile.sh B   the other shapes (tri, 5x10, configs[3] shard, 23x10 fp32 / as worded / reduced, one
#                                    frame), the JSON path, GEMM per-shape and K sweep, the checkers
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3f; mkdir -p $O
cd $R
show() { python3 -c "
import json,sys
d=json.load(open('$O/bench_$1.json')); r=d.get('roofline') or {}; s=r.get('step') or {}
print('$1', round(d['value'],1), 'frames/s', round(d['ms_per_step'],3),'ms', 'io', d.get('io_inclusive') and round(d['io_inclusive']['value'],1), 'json', d.get('json_inclusive') and round(d['json_inclusive']['value'],1), 'gemm', r.get('frac') and round(r['frac'],4), 'step', s.get('frac') and round(s['frac'],4))
"; }
if [ "$1" = A ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc $?" >> $O/gputest.log; grep -E "passed|failed|FAILED|rc" $O/gputest.log | tail -6
  grep -q "Memory access fault" $O/gputest.log && exit 1
  timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
  timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1; show default
  timeout -k 10 300 python bench.py --contexts 1 --streams 1 --json-steps 0 > $O/bench_streams1.json 2>> $O/bench_default.err || exit 1; show streams1
  timeout -k 10 300 python bench.py --contexts 1 --streams 2 --json-steps 0 > $O/bench_contexts1_streams2.json 2>> $O/bench_default.err || exit 1; show contexts1_streams2
  cd /tmp; export TMPDIR=/tmp
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats2 -o run -- python3 $R/bench.py --json-steps 0 > $O/bench_default_under_rocprof.json 2> $O/stats2.err; echo "stats (default: two contexts) rc $?"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -o run -- python3 $R/bench.py --contexts 1 --streams 1 --json-steps 0 --no-io --cpu-sample 0 > $O/bench_streams1_under_rocprof.json 2> $O/stats1.err; echo "stats (one stream) rc $?"
  rm -f $O/stats*/run_kernel_trace.csv
  for C in FETCH_SIZE WRITE_SIZE; do timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc -o $C -- python3 $R/bench.py --contexts 1 --streams 1 --steps 3 --warmup 1 --cpu-sample 0 --no-io --json-steps 0 --no-profile > /dev/null 2> $O/$C.err; echo "$C rc $?"; done
  python3 $R/tools/pmc_traffic.py $O/pmc/FETCH_SIZE_counter_collection.csv $O/pmc/WRITE_SIZE_counter_collection.csv $O/pmc_traffic.json
  B="python3 $R/bench.py --contexts 1 --streams 1 --steps 3 --warmup 1 --cpu-sample 0 --no-io --no-profile --json-steps 0"
  P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
  P2="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE"
  P3="SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT"
  i=0
  for P in "$P1" "$P2" "$P3"; do i=$((i+1))
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/p$i -o d -- $B > /dev/null 2> $O/p$i.err; echo "SQ pass $i rc $?"
  done
  python3 $R/tools/pmc_gemm.py $O/pmc_gemm.json $O/p1/d_counter_collection.csv $O/p2/d_counter_collection.csv $O/p3/d_counter_collection.csv > $O/pmc_gemm.txt; head -12 $O/pmc_gemm.txt
  rm -f $O/*/*_kernel_trace.csv
  python3 - <<PY
import csv, json
for tag in ('default', 'streams1'):
    d = json.load(open('$O/bench_%s_under_rocprof.json' % tag)); r = d['roofline']
    rows = list(csv.DictReader(open('$O/stats%s/run_kernel_stats.csv' % ('2' if tag == 'default' else '1'))))
    g = [x for x in rows if 'k_linear' in x['Name']]
    print(tag, 'under rocprof: value', round(d['value'], 1), '| live HIP events: GEMM avg launch', round(r['avg_launch_ms'], 5), 'ms | rocprof k_linear_* avg',
          round(sum(float(x['TotalDurationNs']) for x in g) / sum(int(x['Calls']) for x in g) / 1e6, 5), 'ms over', sum(int(x['Calls']) for x in g), 'launches')
PY
fi
if [ "$1" = B ]; then
  timeout -k 10 300 python bench.py --mode tri --cpu-sample 20 --json-steps 0 > $O/bench_tri.json 2> $O/bench_b.err; show tri
  timeout -k 10 300 python bench.py --persons 10 --frames 500 --cpu-sample 0 --steps 30 --json-steps 0 > $O/bench_5x10.json 2>> $O/bench_b.err; show 5x10
  timeout -k 10 300 python bench.py --persons 10 --total-frames 12500 --cpu-sample 0 --steps 5 --warmup 1 --json-steps 0 > $O/bench_c4_shard.json 2>> $O/bench_b.err; show c4_shard
  timeout -k 10 300 python bench.py --preset RING23 --persons 10 --frames 24 --cpu-sample 0 --steps 10 --warmup 2 --json-steps 0 > $O/bench_ring24.json 2>> $O/bench_b.err; show ring24
  timeout -k 10 300 python bench.py --preset RING23 --persons 10 --frames 96 --cpu-sample 0 --steps 6 --warmup 2 --json-steps 0 > $O/bench_ring96.json 2>> $O/bench_b.err; show ring96
  timeout -k 10 300 python bench.py --preset RING23 --persons 10 --frames 96 --cpu-sample 0 --steps 6 --warmup 2 --json-steps 0 --cfg4 > $O/bench_ring96_cfg4.json 2>> $O/bench_b.err; show ring96_cfg4
  timeout -k 10 300 python bench.py --preset RING23 --persons 10 --frames 96 --cpu-sample 0 --steps 6 --warmup 2 --json-steps 0 --reduced > $O/bench_ring96_reduced.json 2>> $O/bench_b.err; show ring96_reduced
  timeout -k 10 300 python bench.py --frames 1 --cpu-sample 0 --steps 200 --warmup 20 --json-steps 0 > $O/bench_1frame.json 2>> $O/bench_b.err; show 1frame
  cd /tmp; export TMPDIR=/tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_ring96 -o run -- python3 $R/bench.py --contexts 1 --streams 1 --preset RING23 --persons 10 --frames 96 --cpu-sample 0 --steps 6 --warmup 2 --json-steps 0 --no-io > /dev/null 2> $O/stats_ring96.err; echo "ring96 stats rc $?"
  rm -f $O/stats_ring96/run_kernel_trace.csv
  cd $R
  for i in 1 2; do MPE_JSON_TIMING=1 timeout -k 10 300 python tools/json_stream_probe.py 48000 1000 device >> $O/json_stream_probe.txt 2>&1; done
  timeout -k 10 300 python tools/json_stream_probe.py 48000 1000 host >> $O/json_stream_probe.txt 2>&1
  grep -E "parser:|first window" $O/json_stream_probe.txt
  timeout -k 10 300 python tools/gemm_bench.py > $O/gemm_bench.txt 2>&1; echo "gemm_bench rc $?"
  timeout -k 10 300 python tools/gemm_ksweep.py > $O/gemm_ksweep.txt 2>&1; echo "ksweep rc $?"
  MPE_GEMM_LOADER=0 timeout -k 10 300 python tools/gemm_ksweep.py > $O/gemm_ksweep_no_loader_waves.txt 2>&1
  timeout -k 10 900 python tests/checkers/parity_rate.py 1000 > $O/parity_rate.log 2>&1; tail -2 $O/parity_rate.log | cut -c1-300; cp gpurun_out/parity_rate.json $O/ 2>/dev/null
  for pr in PANOPTIC ARPLAB RING23; do timeout -k 10 600 python tests/checkers/shape_fuzz.py $([ $pr = RING23 ] && echo 60 || echo 300) 21 $pr > $O/shape_fuzz_$pr.log 2>&1; cp gpurun_out/shape_fuzz.json $O/shape_fuzz_$pr.json; tail -1 $O/shape_fuzz_$pr.log | cut -c1-200; done
fi
