# The GPU-box command list behind profiles/r04_*.
#   bash tools/round4_profile.sh C   LDS-pipe / texture-path counters of the GEMM launches: which unit do the
#                                    MFMA waves of k_linear_dma wait for -- the LDS pipe (DMA writes + fragment reads) or the loader side?
#   bash tools/round4_profile.sh T   the GPU test suite + smoke
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4; mkdir -p $O
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
  rm -f $O/*/*_kernel_trace.csv
fi
