# the GPU test-suite once per diagnostic switch of include/mpe.h's frozen list (fallback paths must hold every parity test too)
#   bash tools/run_switch_matrix.sh 1 | 2 | 3 | 4     (parts: a gpurun call is limited to 20 minutes)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/matrix; mkdir -p $O
cd $R
if [ "${1:-1}" = 1 ]; then SW="MPE_NO_COEF_EPILOGUE=1 MPE_FUSED_NO_OVERLAP=1 MPE_CLUSTER_KERNEL=block MPE_L0_GROUPED=0"
elif [ "$1" = 2 ]; then SW="MPE_NO_HEAD_SRC_TABLE=1 MPE_NO_FUSED_ATTENTION=1 MPE_GEMM_NARROW=0 MPE_JSON_WGS=7"
elif [ "$1" = 4 ]; then SW="MPE_LATENCY_PATH=0"      # round 6: small batches on the batch path's own small-batch kernels
else SW="MPE_SKINNY_WAVES=0 MPE_HALF_VEC=4"; fi      # (MPE_GAT_ACC64_MINK=0 removes an accuracy feature: two score tests fail under it by design, r05_switch_matrix.txt)
for sw in $SW; do
  env $sw timeout -k 10 600 python -m pytest tests -m gpu -q > $O/$sw.log 2>&1
  echo "$sw: $(tail -1 $O/$sw.log)"
  grep -q "Memory access fault" $O/$sw.log && { echo "GPU fault under $sw"; exit 1; }
  grep -E "^FAILED" $O/$sw.log | head -5
done
