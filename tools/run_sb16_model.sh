# timing-model variants of the split-bf16 GEMM (tools/sb16_gemm.hip built with -DSB_ORDER / -DSB_RANDOM), one board
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sbm; mkdir -p $O
for b in $R/tools/sb16_gemm_o*; do timeout -k 10 120 $b 2>&1 | grep -E "SB_ORDER|6 products" | tee -a $O/model.txt; done
