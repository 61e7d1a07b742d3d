set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sb32; mkdir -p $O
cd $R
MPE_SB_M32=4 timeout -k 10 200 python3 tools/sb32_check.py > $O/check3_m32_4.txt 2>&1 || { tail -20 $O/check3_m32_4.txt; exit 1; }
cat $O/check3_m32_4.txt
MPE_SB_M32=4 MPE_SB_FL1=1 timeout -k 10 200 python3 tools/sb32_check.py > $O/check3_m32_4_fl1.txt 2>&1 || { tail -20 $O/check3_m32_4_fl1.txt; exit 1; }
cat $O/check3_m32_4_fl1.txt
TOP=8 bash tools/run_env_ab.sh "MPE_SB_M32=4" "MPE_SB_M32=4 MPE_SB_FL1=1" > $O/envab3.txt 2>&1 || { tail -20 $O/envab3.txt; exit 1; }
cat $O/envab3.txt
