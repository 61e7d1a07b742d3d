# the two PMC traffic passes of the default bench -> gpurun_out/pmc2/pmc_traffic.json
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc2; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc -o $C -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-io > /dev/null 2> $O/$C.err; echo "$C rc $?"; done
rm -f $O/pmc/*_kernel_trace.csv
python3 $R/tools/pmc_traffic.py $O/pmc/FETCH_SIZE_counter_collection.csv $O/pmc/WRITE_SIZE_counter_collection.csv $O/pmc_traffic.json
python3 -c "
import json
t=json.load(open('$O/pmc_traffic.json'))['kernels']
for k,v in t.items():
    if 'mlp_rows' in k or 'gat_fused' in k: print(k, v['launches'], round(v['bytes_corrected_per_launch']/1e6,1), 'MB')"
