# the GPU test-suite once per diagnostic switch of include/mpe.h's frozen list (fallback paths must hold every parity test too)
#   bash tools/run_switch_matrix.sh 1 | 2 | 3 | 4     (parts of three switches: a gpurun call is limited to 20 minutes)
# tests/test_gpu_bench.py (bench.py's own regions, several minutes, no parity content of its own) is left out of the matrix.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/matrix; mkdir -p $O
cd $R
if [ "${1:-1}" = 1 ]; then SW="MPE_NO_COEF_EPILOGUE=1 MPE_FUSED_NO_OVERLAP=1 MPE_CLUSTER_KERNEL=block"
elif [ "$1" = 2 ]; then SW="MPE_L0_GROUPED=0 MPE_NO_HEAD_SRC_TABLE=1 MPE_NO_FUSED_ATTENTION=1"
elif [ "$1" = 3 ]; then SW="MPE_GEMM_NARROW=0 MPE_JSON_WGS=7 MPE_SKINNY_WAVES=0"
else SW="MPE_HALF_VEC=4 MPE_LATENCY_PATH=0 MPE_DROPIN_PREFETCH=0"; fi      # round 6: small batches on the batch path's own small-batch kernels; the per-frame mirrors step by step
# (MPE_GAT_ACC64_MINK=0 removes an accuracy feature: two score tests fail under it by design, r05_switch_matrix.txt)
for sw in $SW; do
  env $sw timeout -k 10 380 python -m pytest tests -m gpu -q --ignore=tests/test_gpu_bench.py > $O/$sw.log 2>&1
  echo "$sw: $(tail -1 $O/$sw.log)"
  grep -q "Memory access fault" $O/$sw.log && { echo "GPU fault under $sw"; exit 1; }
  grep -E "^FAILED" $O/$sw.log | head -5
done
