# in-kernel clocks of the split-bf16 launches (diagnostic build: make -C 3d_multi_pose_estimator_amd/csrc exp EXPFLAGS=-DMPE_SB_CLOCK)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sb32; mkdir -p $O
cd $R
export MPE_LIB_VARIANT=${1:-exp}
: > $O/clock2.txt
for w in "mlp 3" "mlp 3 zero" "gat 3" "step 3"; do
  timeout -k 10 150 python3 tools/sb_clock_probe.py $w >> $O/clock2.txt 2>> $O/clock2.err || { tail -5 $O/clock2.err; exit 1; }
done
cat $O/clock2.txt
