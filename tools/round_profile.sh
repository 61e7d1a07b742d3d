set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2n; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc $?" >> $O/gputest.log; grep -E "passed|failed|FAILED" $O/gputest.log | tail -5
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $R/bench.py > $O/bench_under_rocprof.json 2> $O/stats.err; echo "stats rc $?"
rm -f $O/stats/run_kernel_trace.csv
for C in FETCH_SIZE WRITE_SIZE; do timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc -o $C -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-io > /dev/null 2> $O/$C.err; echo "$C rc $?"; done
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc -o sq -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-io > /dev/null 2> $O/sq.err; echo "sq rc $?"
rm -f $O/pmc/*_kernel_trace.csv
python3 $R/tools/pmc_traffic.py $O/pmc/FETCH_SIZE_counter_collection.csv $O/pmc/WRITE_SIZE_counter_collection.csv $O/pmc_traffic.json
python3 $R/tools/pmc_sq.py $O/pmc/sq_counter_collection.csv $O/pmc_sq.json | head -8
cd $R
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc $?"
timeout -k 10 300 python bench.py --mode tri --cpu-sample 20 > $O/bench_tri.json 2>> $O/bench_default.err; echo "tri rc $?"
timeout -k 10 300 python bench.py --streams 2 --cpu-sample 0 > $O/bench_streams2.json 2>> $O/bench_default.err
timeout -k 10 300 python bench.py --persons 10 --frames 500 --cpu-sample 0 --steps 30 > $O/bench_5x10.json 2>> $O/bench_default.err
timeout -k 10 300 python bench.py --persons 10 --total-frames 12500 --cpu-sample 0 --steps 5 --warmup 1 > $O/bench_c4_shard.json 2>> $O/bench_default.err; echo "c4 shard rc $?"
timeout -k 10 300 python bench.py --preset RING23 --persons 10 --frames 24 --cpu-sample 0 --steps 10 --warmup 2 > $O/bench_ring24.json 2>> $O/bench_default.err
timeout -k 10 300 python bench.py --preset RING23 --persons 10 --frames 96 --cpu-sample 0 --steps 6 --warmup 2 > $O/bench_ring96.json 2>> $O/bench_default.err
timeout -k 10 300 python bench.py --preset RING23 --persons 10 --frames 96 --cpu-sample 0 --steps 6 --warmup 2 --reduced > $O/bench_ring96_reduced.json 2>> $O/bench_default.err
timeout -k 10 300 python bench.py --frames 1 --cpu-sample 0 --steps 200 --warmup 20 > $O/bench_1frame.json 2>> $O/bench_default.err
timeout -k 10 600 python tools/json_to_poses.py 16000 1000 > $O/json_to_poses.log 2>&1; tail -3 $O/json_to_poses.log
timeout -k 10 900 python tests/checkers/parity_rate.py 1000 > $O/parity_rate.log 2>&1; tail -2 $O/parity_rate.log | cut -c1-300
for pr in PANOPTIC ARPLAB RING23; do timeout -k 10 600 python tests/checkers/shape_fuzz.py $([ $pr = RING23 ] && echo 60 || echo 300) 21 $pr > $O/shape_fuzz_$pr.log 2>&1; cp gpurun_out/shape_fuzz.json $O/shape_fuzz_$pr.json; tail -1 $O/shape_fuzz_$pr.log | cut -c1-200; done
for f in default tri streams2 5x10 c4_shard ring24 ring96 ring96_reduced 1frame; do python3 -c "
import json,sys
d=json.load(open('$O/bench_$f.json'))
print('$f', round(d['value'],1), 'frames/s', round(d['ms_per_step'],3),'ms', 'io', d['io_inclusive'] and round(d['io_inclusive']['value'],1), 'roof', d['roofline'] and round(d['roofline']['frac'],3))
"; done
