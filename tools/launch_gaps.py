"""Launch-by-launch picture of a short dependent chain from a rocprofv3 kernel trace (csv): per kernel name the calls per step, the
average duration and the average idle time in front of it (start - end of the previous kernel on the device), over the last
`steps` repetitions of the chain.  usage: launch_gaps.py run_kernel_trace.csv STEPS"""
import csv
import sys
from collections import OrderedDict

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = len(rows)
# the timed region: the last `steps` repetitions of the chain; a chain starts at the kernel named by MARK (the first launch of
# mpe_match_batch: the topology / front kernel)
import os
mark = os.environ.get('MARK', 'k_topology,k_lat_l0a').split(',')
idx = [i for i, r in enumerate(rows) if any(m in r['Kernel_Name'] for m in mark)]
if len(idx) > steps + 1:
    use = rows[idx[-steps - 1]:idx[-1]]
    per = len(use) / float(steps)
else:
    use, per = rows, float(n)
stat = OrderedDict()
prev_end = None
t_busy = t_gap = 0
for r in use:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].split('(')[0][:90]
    d = stat.setdefault(name, [0, 0, 0])
    d[0] += 1
    d[1] += e - s
    t_busy += e - s
    if prev_end is not None:
        d[2] += max(0, s - prev_end)
        t_gap += max(0, s - prev_end)
    prev_end = e
k = len(use) / float(per)
print('launches per step: %.1f   kernel time per step %.1f us   idle between kernels per step %.1f us   (%d steps)' % (per, t_busy / k / 1e3, t_gap / k / 1e3, int(k)))
print('%-92s %6s %9s %9s' % ('kernel', 'calls', 'avg us', 'gap us'))
for name, (c, dur, gap) in sorted(stat.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print('%-92s %6.1f %9.2f %9.2f' % (name, c / k, dur / c / 1e3, gap / c / 1e3))

# per position in the chain (only when every step has the same launch count): what each launch of a step takes
if abs(per - round(per)) < 1e-9 and k >= 2:
    per_i = int(round(per))
    pos = [[0, 0, None] for _ in range(per_i)]
    prev_end = None
    for j, r in enumerate(use):
        s0, e0 = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        q = pos[j % per_i]
        q[0] += e0 - s0
        if prev_end is not None:
            q[1] += max(0, s0 - prev_end)
        q[2] = r['Kernel_Name'].split('(')[0][:70]
        prev_end = e0
    print('\nby position in the step:  #  avg us  gap us  kernel')
    for j, q in enumerate(pos):
        print('%3d %7.2f %6.2f  %s' % (j, q[0] / k / 1e3, q[1] / k / 1e3, q[2]))
