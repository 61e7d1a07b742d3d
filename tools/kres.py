#!/usr/bin/env python3
"""Per-kernel resource usage of a .hip file as the compiler reports it (-Rpass-analysis=kernel-resource-usage):
   python tools/kres.py 3d_multi_pose_estimator_amd/csrc/gemm_sb16.hip [name filter] [extra hipcc flags ...]"""
import re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
extra = sys.argv[3:]
cmd = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-Rpass-analysis=kernel-resource-usage',
       '-c', src, '-o', '/dev/null'] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r'remark:\s+(.*?)\s*\[-Rpass', line)
    if not m:
        continue
    t = m.group(1)
    if t.startswith('Function Name:'):
        cur = t.split(':', 1)[1].strip()
        rows[cur] = {}
    elif cur and ':' in t:
        k, v = t.split(':', 1)
        rows[cur][k.strip()] = v.strip()
for name, r in rows.items():
    dem = subprocess.run(['/usr/bin/c++filt', name], capture_output=True, text=True).stdout.strip()
    dem = dem.split('(')[0]
    if flt and flt not in dem:
        continue
    print('%-70s VGPR %3s AGPR %3s spill %3s scratch %4s occ %s' % (dem[-70:], r.get('VGPRs'), r.get('AGPRs'), r.get('VGPRs Spill'),
          r.get('ScratchSize [bytes/lane]'), r.get('Occupancy [waves/SIMD]')))
