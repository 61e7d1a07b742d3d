"""In-kernel shader clock of the split-bf16 tile kernel's launches (MI355X_MICROARCH.md, DVFS give-back item 6): a diagnostic build
(make -C 3d_multi_pose_estimator_amd/csrc exp EXPFLAGS=-DMPE_SB_CLOCK; MPE_LIB_VARIANT=exp) stamps s_memtime / s_memrealtime around the
tile loop of the first MFMA wave of every workgroup.  An MLP of 3072-wide layers runs back to back for SECONDS, then the stamps of the
last tile-kernel launch (4004 x 3072 x 3072) are read: clock = d(memtime) / d(memrealtime) x 100 MHz, median over the workgroups.
   MPE_LIB_VARIANT=exp [MPE_SB_M32=5] python tools/sb_clock_probe.py [seconds] [zero]"""
import ctypes as C, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import env, pkg
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
zero = len(sys.argv) > 2 and sys.argv[2] == 'zero'
e = env('panoptic')
eng = pkg('pipeline').Engine(e.params, e.calib, max_frames=1001, max_persons_per_camera=4)
g = torch.Generator().manual_seed(3)
dims = [1260, 3072, 3072, 3072, 3072, 54]
sd = {}
for i in range(len(dims) - 1):
    w = torch.randn(dims[i + 1], dims[i], generator=g) / np.sqrt(dims[i])
    sd['layers.%d.weight' % (2 * i + 1)] = (w * 0 if zero else w).numpy()
    sd['layers.%d.bias' % (2 * i + 1)] = torch.randn(dims[i + 1], generator=g).numpy() * 0.1
eng.load_mlp(sd)
x = torch.randn(4004, 1260, generator=g).cuda()
for _ in range(3):
    eng.mlp_forward(x)
torch.cuda.synchronize()
t0 = time.time(); n = 0
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
while time.time() - t0 < secs:
    for _ in range(20):
        eng.mlp_forward(x)
    n += 20
    torch.cuda.synchronize()
ev1.record(); torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1) / n
out = (C.c_ulonglong * (256 * 4))()
rc = eng.lib.mpe_debug_sb_stamps(out, 256)
a = np.frombuffer(out, dtype=np.uint64).reshape(256, 4).astype(np.float64)
dt, dr = a[:, 2] - a[:, 0], a[:, 3] - a[:, 1]
ok = (dr > 0) & (dt > 0)
clk = dt[ok] / dr[ok] * 100e6
flop = 2.0 * 4004 * sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1))
print('MPE_SB_M32=%s zero_weights=%s: %d passes, %.3f ms per MLP pass (%.1f fp32-equivalent TFLOP/s incl. gaps); stamps rc %d, %d workgroups; '
      'in-kernel clock median %.3f GHz (min %.3f max %.3f); tile loop %.1f us median' % (
          os.environ.get('MPE_SB_M32', '0'), zero, n, ms, flop / ms / 1e9, rc, int(ok.sum()), np.median(clk) / 1e9, clk.min() / 1e9, clk.max() / 1e9,
          np.median(dr[ok]) / 100.0))
eng.close()
