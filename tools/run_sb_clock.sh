set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sb32; mkdir -p $O
cd $R
export MPE_LIB_VARIANT=exp
: > $O/clock.txt
for m in 0 1 5; do
  MPE_SB_M32=$m timeout -k 10 120 python3 tools/sb_clock_probe.py 3 >> $O/clock.txt 2>> $O/clock.err || { tail -5 $O/clock.err; exit 1; }
done
MPE_SB_M32=0 timeout -k 10 120 python3 tools/sb_clock_probe.py 3 zero >> $O/clock.txt 2>> $O/clock.err || { tail -5 $O/clock.err; exit 1; }
MPE_SB_M32=5 timeout -k 10 120 python3 tools/sb_clock_probe.py 3 zero >> $O/clock.txt 2>> $O/clock.err || { tail -5 $O/clock.err; exit 1; }
cat $O/clock.txt
