# SQ counters of the attention kernel: what do its waves spend their cycles on?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sq2; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $O/p1 -o a -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-io > /dev/null 2> $O/a.err; echo "pass1 rc $?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/p2 -o b -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-io > /dev/null 2> $O/b.err; echo "pass2 rc $?"
python3 - <<PY
import csv,collections,glob
for f in glob.glob('$O/p*/*counter_collection.csv'):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0]
        if 'gat_fused<4' in k or 'k_linear_dma<true, false, 5' in k or 'k_mlp_rows' in k:
            acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
    for k,v in acc.items():
        print(f.split('/')[-2], k[:40], {c: '%.3g' % x for c,x in v.items()})
PY
