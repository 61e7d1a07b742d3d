# round 5: the tests this round added or touched, then the default bench
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/suite; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "within_1e_3 or f64_matrix_pipe or flush_per_stage or attn_fp16 or reduced_mode or cfg4 or frame_loop_calls or dropin or split_bf16 or mlp_default_mode or bench" > $O/quick.log 2>&1; rc=$?
tail -12 $O/quick.log
[ $rc = 0 ] || exit $rc
