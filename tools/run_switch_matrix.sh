# the GPU test-suite once per diagnostic switch (fallback paths must hold every parity test too)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/matrix; mkdir -p $O
cd $R
for sw in MPE_NO_COEF_EPILOGUE=1 MPE_FUSED_NO_OVERLAP=1 MPE_CLUSTER_KERNEL=block MPE_L0_GROUPED=0 MPE_GEMM_TUNE=8 MPE_NO_HEAD_SRC_TABLE=1 MPE_NO_FUSED_ATTENTION=1; do
  env $sw timeout -k 10 600 python -m pytest tests -m gpu -q > $O/$sw.log 2>&1
  echo "$sw: $(tail -1 $O/$sw.log)"
  grep -E "^FAILED" $O/$sw.log | head -5
done
