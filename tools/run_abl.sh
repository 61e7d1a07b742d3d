set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/abl; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for A in 0 1 2 4 8 12 15; do
export MPE_FUSED_ABLATE=$A
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s$A -o run -- python3 $R/bench.py --steps 40 --warmup 5 --cpu-sample 0 --no-io > $O/bench$A.json 2> $O/stats$A.err || { echo "rc fail $A"; exit 1; }
rm -f $O/s$A/run_kernel_trace.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$O/s$A/run_kernel_stats.csv')))
print('ablate $A:', '  '.join('%s %.1f' % (r['Name'][5:26], float(r['AverageNs'])/1e3) for r in rows if 'k_gat_fused' in r['Name']))
PY
done
