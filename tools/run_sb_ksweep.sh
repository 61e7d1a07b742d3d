R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ksw; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/k -o run -- python3 $R/tools/sb_ksweep.py > $O/out.txt 2> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
python3 - <<PY
import csv, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open('$O/k/run_kernel_trace.csv')):
    if 'k_linear_sb<' in r['Kernel_Name']:
        acc[(r['Kernel_Name'].split('(')[0][-40:], int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), r['Dispatch_Id'])] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
# launches in order: 8 per (shape, K)
vals = [v for k, v in sorted(acc.items(), key=lambda kv: int(kv[0][2]))]
names = [k for k, v in sorted(acc.items(), key=lambda kv: int(kv[0][2]))]
shapes = [(180000, 400, k) for k in (416, 832, 1664, 3328)] + [(4004, 3072, k) for k in (416, 832, 1664, 3328)]
for i, (m, n, k) in enumerate(shapes):
    v = vals[i * 8:(i + 1) * 8]
    t = min(v) / 1e3
    print('M=%6d N=%4d K=%4d  %-40s %8.1f us  %6.1f fp32-equivalent TFLOP/s' % (m, n, k, names[i * 8][0], t, 2.0 * m * n * k / t / 1e6))
PY
rm -f $O/k/run_kernel_trace.csv
