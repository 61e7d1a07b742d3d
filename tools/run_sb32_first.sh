# round 5, first GPU call: numerics of the 32x32x16 order, library kernel accuracy in both forms, per-kernel times under the switch
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sb32; mkdir -p $O
cd $R
timeout -k 10 200 tools/sb32_numerics > $O/numerics.txt 2>&1 || { tail -5 $O/numerics.txt; exit 1; }
for m in 0 3; do MPE_SB_M32=$m timeout -k 10 200 python3 tools/sb32_check.py > $O/check_m32_$m.txt 2>&1 || { tail -20 $O/check_m32_$m.txt; exit 1; }; done
MPE_SB_M32=1 MPE_SB_FL1=1 timeout -k 10 200 python3 tools/sb32_check.py > $O/check_m32_1_fl1.txt 2>&1 || { tail -20 $O/check_m32_1_fl1.txt; exit 1; }
cat $O/check_m32_*.txt
TOP=12 bash tools/run_env_ab.sh "MPE_SB_M32=1" "MPE_SB_M32=3" "MPE_SB_M32=1 MPE_SB_FL1=1" > $O/envab.txt 2>&1 || { tail -20 $O/envab.txt; exit 1; }
cat $O/envab.txt
