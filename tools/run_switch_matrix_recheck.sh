# the three legs of the switch matrix whose assertions were restated: run again
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/matrix2; mkdir -p $O
cd $R
for sw in MPE_NO_COEF_EPILOGUE=1 MPE_L0_GROUPED=0 MPE_GEMM_TUNE=8; do
  env $sw timeout -k 10 600 python -m pytest tests -m gpu -q > $O/$sw.log 2>&1
  echo "$sw: $(tail -1 $O/$sw.log)"
  grep -q "Memory access fault" $O/$sw.log && { echo "GPU fault under $sw"; exit 1; }
  grep -E "^FAILED" $O/$sw.log | head -5
done
