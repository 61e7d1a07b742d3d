# Round 3: where does the fp32-MFMA GEMM lose its 30 %?  Per-shape rates (fp32 chain vs f64 running
# sums), SQ stall / issue counters per (instantiation, grid), and the same with plain fp32 sums in
# the MLP (the ACC64 ablation).  Output under gpurun_out/r3a; summaries are copied to profiles/.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r3a}; mkdir -p $O
cd $R
timeout -k 10 300 python tools/gemm_bench.py > $O/gemm_bench.txt 2>&1; echo "gemm_bench rc $?"; cat $O/gemm_bench.txt
cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-io --no-profile"
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
P2="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE"
P3="SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_VALU_ADD_F64"
i=0
for P in "$P1" "$P2" "$P3"; do i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/p$i -o d -- $B > /dev/null 2> $O/p$i.err; echo "default pass $i rc $?"
done
i=0
for P in "$P1" "$P2"; do i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/f$i -o f -- $B --fast-mlp > /dev/null 2> $O/f$i.err; echo "fast-mlp pass $i rc $?"
done
python3 $R/tools/pmc_gemm.py $O/pmc_gemm_default.json $O/p1/d_counter_collection.csv $O/p2/d_counter_collection.csv $O/p3/d_counter_collection.csv | tee $O/pmc_gemm_default.txt
python3 $R/tools/pmc_gemm.py $O/pmc_gemm_fast_mlp.json $O/f1/f_counter_collection.csv $O/f2/f_counter_collection.csv | tee $O/pmc_gemm_fast_mlp.txt
rm -f $O/*/*_kernel_trace.csv
cd $R
timeout -k 10 300 python bench.py --cpu-sample 0 --no-io > $O/bench_default.json 2> $O/bench.err; echo "bench rc $?"
timeout -k 10 300 python bench.py --cpu-sample 0 --no-io --fast-mlp > $O/bench_fast_mlp.json 2>> $O/bench.err
timeout -k 10 300 python bench.py --cpu-sample 0 --no-io --streams 2 > $O/bench_streams2.json 2>> $O/bench.err
for f in default fast_mlp streams2; do python3 -c "
import json
d=json.load(open('$O/bench_$f.json')); r=d.get('roofline') or {}
print('$f', round(d['value'],1), 'frames/s', round(d['ms_per_step'],3), 'ms  gemm frac', r.get('frac'))"; done
