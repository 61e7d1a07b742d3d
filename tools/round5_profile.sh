# The GPU-box command list behind profiles/r05_* (parts T, A, B as in round 4; C = the SQ / TA counter passes of the split-bf16 launches).
#   bash tools/round4_profile.sh C   LDS-pipe / texture-path counters of the GEMM launches: which unit do the
#                                    MFMA waves of k_linear_dma wait for -- the LDS pipe (DMA writes + fragment reads) or the loader side?
#   bash tools/round4_profile.sh T   the GPU test suite + smoke
#   bash tools/round4_profile.sh A   default bench plain and under rocprofv3 --kernel-trace --stats (two contexts / one stream), the two PMC
#                                    traffic passes
#   bash tools/round4_profile.sh B   the other shapes (tri, 5x10, configs[3] shard, 23x10 fp32 / as worded / reduced, one frame) + ring96 stats
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd $R
if [ "$1" = T ]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $O/gputest.log 2>&1; rc=$?; echo "pytest rc $rc" >> $O/gputest.log; grep -E "passed|failed|FAILED|rc" $O/gputest.log | tail -6
  [ $rc = 0 ] || exit 1
  timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
fi
if [ "$1" = C ]; then
  cd /tmp; export TMPDIR=/tmp
  B="python3 $R/bench.py --contexts 1 --streams 1 --steps 3 --warmup 1 --cpu-sample 0 --no-io --no-profile --json-steps 0"
  P4="SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT"
  P5="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE"
  P6="TA_TA_BUSY_sum TA_FLAT_READ_LDS_WAVEFRONTS_sum TD_TD_BUSY_sum TD_TC_STALL_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE"
  P7="TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum GRBM_GUI_ACTIVE"
  P8="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
  i=3
  for P in "$P4" "$P5" "$P6" "$P7" "$P8"; do i=$((i+1))
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/p$i -o d -- $B > /dev/null 2> $O/p$i.err; rc=$?; echo "pass $i rc $rc"
    [ $rc = 124 ] && exit 1
    tail -2 $O/p$i.err | cut -c1-300
  done
  python3 $R/tools/pmc_gemm.py $O/pmc_lds.json $O/p4/d_counter_collection.csv $O/p5/d_counter_collection.csv $O/p6/d_counter_collection.csv $O/p7/d_counter_collection.csv $O/p8/d_counter_collection.csv > $O/pmc_lds.txt 2>&1; head -14 $O/pmc_lds.txt
  rm -f $O/*/*_kernel_trace.csv $O/p?/d_counter_collection.csv       # (the merged json / txt are the record; gpurun_out/ is capped at 64 MiB)
fi
show() { python3 -c "
import json,sys
d=json.load(open('$O/bench_$1.json')); r=d.get('roofline') or {}; s=r.get('step') or {}; dl=d.get('dropin_loop') or {}; j=d.get('json_inclusive') or {}
pa=d.get('parity') or {}; ma=pa.get('mlp_max_accuracy') or {}; fx=pa.get('mlp_f64_exact') or {}
print('$1', round(d['value'],1), 'frames/s', round(d['ms_per_step'],3),'ms', 'host-to-host', d.get('value_host_to_host') and round(d['value_host_to_host'],1), 'json cold/warm', j.get('value') and round(j['value'],1), j.get('value_warm') and round(j['value_warm'],1), 'gemm', r.get('frac') and round(r['frac'],4), 'step', s.get('frac') and round(s['frac'],4), 'dropin ms/frame', dl.get('ms_per_frame') and round(dl['ms_per_frame'],3), 'inside', dl.get('inside_mirrors_ms') and round(dl['inside_mirrors_ms'],3),
      '| gpu-exact mm default/max/f64', pa.get('gpu_vs_exact_mm'), ma.get('gpu_vs_exact_mm'), fx.get('gpu_vs_exact_mm'), 'fps max/f64', ma.get('frames_per_s') and round(ma['frames_per_s']), fx.get('frames_per_s') and round(fx['frames_per_s']))
"; }
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
  timeout -k 10 300 python bench.py --preset RING23 --persons 10 --frames 96 --cpu-sample 0 --steps 6 --warmup 2 $X --cfg4 > $O/bench_ring96_cfg4.json 2>> $O/bench_b.err; show ring96_cfg4
  timeout -k 10 300 python bench.py --preset RING23 --persons 10 --frames 96 --cpu-sample 0 --steps 6 --warmup 2 $X --reduced > $O/bench_ring96_reduced.json 2>> $O/bench_b.err; show ring96_reduced
  timeout -k 10 300 python bench.py --frames 1 --cpu-sample 0 --steps 200 --warmup 20 $X > $O/bench_1frame.json 2>> $O/bench_b.err; show 1frame
  cd /tmp; export TMPDIR=/tmp
  for M in "" "--cfg4"; do
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_ring96$M -o run -- python3 $R/bench.py --contexts 1 --streams 1 --preset RING23 --persons 10 --frames 96 --cpu-sample 0 --steps 6 --warmup 2 $X --no-io $M > /dev/null 2> $O/stats_ring96$M.err; echo "ring96 $M stats rc $?"
  done
  rm -f $O/stats_ring96*/run_kernel_trace.csv
fi
