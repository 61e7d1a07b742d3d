# the GPU test-suite once per diagnostic switch (fallback paths must hold every parity test too)
#   bash tools/run_switch_matrix.sh 1 | 2 | 3 | 4     (parts: a gpurun call is limited to 20 minutes; 3 and 4 = round 4)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/matrix; mkdir -p $O
cd $R
if [ "${1:-1}" = 1 ]; then SW="MPE_NO_COEF_EPILOGUE=1 MPE_FUSED_NO_OVERLAP=1 MPE_CLUSTER_KERNEL=block MPE_L0_GROUPED=0 MPE_GEMM_TUNE=8"
elif [ "$1" = 2 ]; then SW="MPE_NO_HEAD_SRC_TABLE=1 MPE_NO_FUSED_ATTENTION=1 MPE_GEMM_LOADER=0 MPE_GEMM_NARROW=0 MPE_JSON_WGS=7"
# round 4 (split-bf16 tile kernel): both tile forms for every GAT launch, the tile kernel at every batch size, and the legs whose
# paths the split GEMMs touch
elif [ "$1" = 3 ]; then SW="MPE_SB_GAT_MW=4 MPE_SB_GAT_MW=8 MPE_SKINNY_WAVES=0 MPE_NO_COEF_EPILOGUE=1 MPE_SB_PERS=0"
else SW="MPE_L0_GROUPED=0 MPE_NO_FUSED_ATTENTION=1 MPE_GEMM_NARROW=0"; fi
for sw in $SW; do
  env $sw timeout -k 10 600 python -m pytest tests -m gpu -q > $O/$sw.log 2>&1
  echo "$sw: $(tail -1 $O/$sw.log)"
  grep -q "Memory access fault" $O/$sw.log && { echo "GPU fault under $sw"; exit 1; }
  grep -E "^FAILED" $O/$sw.log | head -5
done
