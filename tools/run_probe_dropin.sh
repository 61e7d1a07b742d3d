set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/dropin; mkdir -p $O
cd $R
timeout -k 10 300 python tools/dropin_loop_probe.py 200 4 > $O/probe.txt 2>&1 || { tail -20 $O/probe.txt; exit 1; }
head -70 $O/probe.txt
