"""Per-tile fixed cost of the split-bf16 tile kernel: fp32-equivalent TFLOP/s of mpe_linear (split form) against K at the GAT shape
(180 000 x 400, no f64 sums: twelve-wave form) and at the MLP shape (4004 x 3072, f64 sums).  python tools/sb_ksweep.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import env, pkg
e = env('panoptic')
eng = pkg('pipeline').Engine(e.params, e.calib, max_frames=8, max_persons_per_camera=4)
g = torch.Generator().manual_seed(5)
for (m, n, f64) in ((180000, 400, False), (4004, 3072, True)):
    for k in (416, 832, 1664, 3328):
        x = torch.randn(m, k, generator=g).cuda()
        w = (torch.randn(n, k, generator=g) / np.sqrt(k)).numpy(); b = torch.randn(n, generator=g).numpy()
        for _ in range(3):
            eng.linear(x, w, b, 0.1, split=True, split_f64=f64)
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # (engine.linear uploads the weights and makes the planes in every call: time the kernel by rocprofv3 instead)
        t0.record()
        for _ in range(5):
            eng.linear(x, w, b, 0.1, split=True, split_f64=f64)
        t1.record(); torch.cuda.synchronize()
        print('M=%d N=%d K=%d f64=%s: %.3f ms per call (incl. upload + split)' % (m, n, k, f64, t0.elapsed_time(t1) / 5))
        del x
eng.close()
