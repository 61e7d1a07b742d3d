set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_shapes -o run -- python3 $R/tools/sb16_shapes.py > $O/sb16_shapes.txt 2> $O/sb16_shapes.err; echo "rc $?"
cat $O/sb16_shapes.txt
python3 - <<PY
import csv
for r in csv.DictReader(open('$O/stats_shapes/run_kernel_trace.csv')):
    n = r['Kernel_Name']
    if 'k_linear_sb' in n or 'k_linear_dma' in n:
        print('%-70s grid %8s  %9.1f us' % (n[:70].replace('void mpe::','').replace('(anonymous namespace)::',''), r['Grid_Size'], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
PY
rm -f $O/stats_shapes/run_kernel_trace.csv
