# K sweep of mpe_linear for the product library and two TIMING-ONLY ablation builds of k_linear_dma (wrong data):
#   exp4: stages are requested as usual but nobody waits for them to land (s_barrier without vmcnt(0))
#   exp8: no staging at all
#   exp32: the same requests as plain register loads (same bytes through TA / TCP / L2, nothing written into LDS); exp36 = exp32 + exp4
# Built with: make -C 3d_multi_pose_estimator_amd/csrc exp EXPFLAGS=-DMPE_EXP=4 && cp ../libmpe_hip_exp.so ../libmpe_hip_exp4.so (and =8)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
for v in ${VARIANTS:-product exp4 exp8 product exp4 exp8}; do
  lv=$v; [ $v = product ] && lv=""
  echo "== lib $v" >> $O/ablation_ksweep.txt
  MPE_LIB_VARIANT=$lv timeout -k 10 200 python tools/gemm_ksweep.py >> $O/ablation_ksweep.txt 2>&1 || { tail -5 $O/ablation_ksweep.txt; exit 1; }
done
grep -E "==|K= 416|K=3328" $O/ablation_ksweep.txt
