set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sb32; mkdir -p $O
cd $R
MPE_SB_M32=3 MPE_SB_M32_NC=2 timeout -k 10 200 python3 tools/sb32_check.py > $O/check2_m32_3.txt 2>&1 || { tail -20 $O/check2_m32_3.txt; exit 1; }
cat $O/check2_m32_3.txt
TOP=8 bash tools/run_env_ab.sh "MPE_SB_M32=1" "MPE_SB_M32=3 MPE_SB_M32_NC=2" "MPE_SB_M32=1 MPE_SB_FL1=1" > $O/envab2.txt 2>&1 || { tail -20 $O/envab2.txt; exit 1; }
cat $O/envab2.txt
