"""GAT-shaped GEMMs on the fp32 MFMA (k_linear_dma) and in the split-bf16 form without f64 sums (k_linear_sb<.., F64 = false>):
run under rocprofv3 --kernel-trace --stats and compare the kernels' average durations.  Also prints both forms' error against
float64 on the same operands."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import env, pkg
e = env('panoptic')
eng = pkg('pipeline').Engine(e.params, e.calib, max_frames=8, max_persons_per_camera=4)
g = torch.Generator().manual_seed(3)
for (m, k, n, slope) in ((180000, 416, 400, 0.15), (180000, 416, 320, None), (180000, 320, 150, None), (20000, 902, 400, None)):
    x = torch.randn(m, k, generator=g); x = torch.where(x > 0, x, 0.15 * x).cuda()
    w = (torch.randn(n, k, generator=g) / np.sqrt(k)).numpy(); b = torch.randn(n, generator=g).numpy()
    ex = None
    for rep in range(3):
        y0 = eng.linear(x, w, b, slope)
        y1 = eng.linear(x, w, b, slope, split=True, split_f64=False)
    sub = slice(0, 4096)
    ex = x[sub].double().cpu() @ torch.from_numpy(w).double().T + torch.from_numpy(b).double()
    if slope is not None: ex = torch.where(ex > 0, ex, slope * ex)
    ulp = 2.0 ** (np.floor(np.log2(ex.abs().max().item())) - 23)
    r0 = ((y0[sub].cpu().double() - ex) ** 2).mean().sqrt().item() / ulp
    r1 = ((y1[sub].cpu().double() - ex) ** 2).mean().sqrt().item() / ulp
    print('%dx%dx%d: rms error fp32 chain %.3f ulp, split-bf16 (no flush) %.3f ulp; max %.2f / %.2f' % (m, k, n, r0, r1, (y0[sub].cpu().double() - ex).abs().max().item() / ulp, (y1[sub].cpu().double() - ex).abs().max().item() / ulp))
eng.close()
