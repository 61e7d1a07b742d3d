# one SQ / GRBM counter pass of the one-stream step: MFMA busy and effective clock per kernel (profiles/r05_pmc_sb_kernels.txt)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --contexts 1 --streams 1 --steps 6 --warmup 2 --cpu-sample 0 --no-io --no-profile --json-steps 0 --dropin-frames 0 --no-accuracy-modes"
P8="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $P8 --output-format csv -d $O/p8 -o d -- $B > /dev/null 2> $O/p8.err; rc=$?; echo "pass rc $rc"
[ $rc = 0 ] || { tail -3 $O/p8.err; exit 1; }
python3 $R/tools/pmc_gemm.py $O/pmc_sb_kernels.json $O/p8/d_counter_collection.csv > $O/pmc_sb_kernels.txt 2>&1; cat $O/pmc_sb_kernels.txt
rm -f $O/p8/*_kernel_trace.csv $O/p8/d_counter_collection.csv
