# launch-by-launch trace of the one-frame step: bash tools/run_1frame_trace.sh TAG [ENV=VALUE ...]   (FRAMES=8: eight frames per call)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
X="--json-steps 0 --dropin-frames 0 --cpu-sample 0 --no-accuracy-modes"
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$TAG -o run -- python3 $R/bench.py --frames ${FRAMES:-1} --contexts 1 --streams 1 --steps 200 --warmup 20 $X --no-io --no-profile > $O/trace_$TAG.json 2> $O/trace_$TAG.err || { tail -3 $O/trace_$TAG.err; exit 1; }
python3 $R/tools/launch_gaps.py $O/trace_$TAG/run_kernel_trace.csv 150 > $O/1frame_${TAG}_launches.txt 2>&1
rm -rf $O/trace_$TAG
python3 -c "
import json; d=json.load(open('$O/trace_$TAG.json')); print('$TAG under rocprof:', round(d['ms_per_step']*1e3,1), 'us per step of ${FRAMES:-1} frame(s)')"
awk '/by position/{f=1} f' $O/1frame_${TAG}_launches.txt
