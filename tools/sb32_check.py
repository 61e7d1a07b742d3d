"""Accuracy of the library's split-bf16 launches against the layer evaluated in float64, at the shapes of the path, and whether five
rows keep their bits in a batch of five.  Errors in fp32 ulps of the output scale (max |y|), as tests/test_gpu_parity.py measures them.
(Round 5 ran it under MPE_SB_M32 = 1 | 3 | 5 on commit 80a9985, whose library held the 32 x 32 x 16 kernels: profiles/r05_sb32_forms.txt;
the switches it prints are gone from the library since.)
   python tools/sb32_check.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import env, pkg
e = env('panoptic')
eng = pkg('pipeline').Engine(e.params, e.calib, max_frames=8, max_persons_per_camera=4)
print('MPE_SB_M32=%s MPE_SB_FL1=%s' % (os.environ.get('MPE_SB_M32', '0'), os.environ.get('MPE_SB_FL1', '0')))
g = torch.Generator().manual_seed(11)
shapes = [(4004, 3072, 1260, True, 0.1), (4004, 3072, 3072, True, 0.1), (4004, 2048, 3072, True, 0.1), (4004, 1024, 2048, True, 0.1),
          (3000, 1024, 1024, True, None), (9000, 400, 400, False, 0.15), (9000, 320, 400, False, None), (9000, 320, 320, False, 0.15),
          (2999, 150, 320, False, None), (700, 400, 902, True, None)]
for (m, n, k, f64, slope) in shapes:
    x = torch.randn(m, k, generator=g)
    x = torch.where(x > 0, x, 0.1 * x)
    w = torch.randn(n, k, generator=g) / np.sqrt(k)
    b = torch.randn(n, generator=g)
    ex = x.cuda().double() @ w.cuda().double().T + b.cuda().double()
    if slope is not None:
        ex = torch.where(ex > 0, ex, slope * ex)
    y = eng.linear(x.cuda(), w.numpy(), b.numpy(), slope, split=True, split_f64=f64)
    y1 = eng.linear(x[:5].cuda(), w.numpy(), b.numpy(), slope, split=True, split_f64=f64)
    scale = ex.abs().max().item()
    ulp = 2.0 ** (np.floor(np.log2(scale)) - 23)
    err = (y.double() - ex) / ulp
    print('M=%5d N=%4d K=%4d f64=%d: rms %.3f max %.2f ulp; rows 0-4 equal to a batch of five: %s (max diff %.2f ulp); finite %s' % (
        m, n, k, f64, err.pow(2).mean().sqrt().item(), err.abs().max().item(), torch.equal(y1, y[:5]),
        ((y1 - y[:5]).abs().max().item() / ulp), bool(torch.isfinite(y).all())))
eng.close()
