"""Split-bf16 MLP mode against the default (fp32 MFMA + f64 running sums per stage): accuracy on MLP rows against the exactly
evaluated network, bit-identity across batch sizes, and the time of mpe_mlp_forward at the headline's 4000 rows."""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import env, load_case, pkg, ALL_CASES
e = env('panoptic')
pipeline = pkg('pipeline')
eng = pipeline.Engine(e.params, e.calib, max_frames=1024, max_persons_per_camera=4)
eng.load_mlp(e.mlp_room)
xs = []
for v, name in ALL_CASES:
    if v != 'panoptic': continue
    arr, frames = load_case(name, v)
    for n in range(len(frames)):
        if 'f%d_mlp_in' % n in arr: xs.append(arr['f%d_mlp_in' % n])
x = torch.from_numpy(np.concatenate(xs))
sd = e.mlp_room
def exact(x):
    h = x.double()
    keys = sorted({int(k.split('.')[1]) for k in sd})
    for i, k in enumerate(keys):
        h = h @ torch.from_numpy(sd['layers.%d.weight' % k]).double().T + torch.from_numpy(sd['layers.%d.bias' % k]).double()
        if i < len(keys) - 1:
            h = torch.where(h > 0, h, 0.1 * h).float().double()
    return h
ex = exact(x)
def cpu(x):
    h = x
    keys = sorted({int(k.split('.')[1]) for k in sd})
    for i, k in enumerate(keys):
        h = torch.nn.functional.linear(h, torch.from_numpy(sd['layers.%d.weight' % k]), torch.from_numpy(sd['layers.%d.bias' % k]))
        if i < len(keys) - 1: h = torch.nn.functional.leaky_relu(h, 0.1)
    return h
ref = cpu(x).double()
big = x.repeat((4000 + x.shape[0] - 1) // x.shape[0], 1)[:4000].cuda()
out = {}
for mode in ('acc64', 'split'):
    eng.set_precision(False, True, mlp_split=(mode == 'split'))
    y_small = eng.mlp_forward(x.cuda()).cpu()
    y_one = torch.cat([eng.mlp_forward(x[i:i + 1].cuda()).cpu() for i in range(min(6, x.shape[0]))])
    y_big = eng.mlp_forward(big).cpu()
    same = torch.equal(y_big[:x.shape[0]], y_small) and torch.equal(y_one, y_small[:y_one.shape[0]])
    torch.cuda.synchronize()
    for _ in range(3): eng.mlp_forward(big)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): eng.mlp_forward(big)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
    e_gpu = (y_small.double() - ex).abs().max().item()
    out[mode] = y_small
    print('%-6s rows %d: |gpu-exact| %.3e  |ref-exact| %.3e  |gpu-ref| %.3e (MLP units; x1e4 = mm)  rows bit-identical at M=1/%d/4000: %s  mlp_forward(4000 rows) %.3f ms'
          % (mode, x.shape[0], e_gpu, (ref - ex).abs().max().item(), (y_small.double() - ref).abs().max().item(), x.shape[0], same, dt * 1e3))
print('split vs acc64 max diff %.3e' % (out['split'] - out['acc64']).abs().max().item())
# raw GEMM shapes through mpe_linear: small-batch kernels against the tile kernel
g = torch.Generator().manual_seed(5)
for (m, k, n) in ((2100, 1260, 3072), (2100, 3072, 2048), (2100, 1024, 54), (300, 416, 400)):
    w = ((torch.rand(n, k, generator=g) - 0.5) * 0.1).numpy(); b = torch.rand(n, generator=g).numpy()
    xx = (torch.rand(m, k, generator=g) - 0.3)
    yt = eng.linear(xx.cuda(), w, b, slope=0.1, split=True).cpu()
    y1 = torch.cat([eng.linear(xx[i:i + 3].cuda(), w, b, slope=0.1, split=True).cpu() for i in (0, 700, m - 3)])
    want = torch.cat([yt[0:3], yt[700:703], yt[m - 3:m]])
    exd = torch.nn.functional.leaky_relu(xx.double() @ torch.from_numpy(w).double().T + torch.from_numpy(b).double(), 0.1)
    ya = eng.linear(xx.cuda(), w, b, slope=0.1, acc64=True).cpu()
    print('linear %dx%dx%d: small-batch rows == tile rows: %s | max err split %.3e acc64 %.3e (scale %.2f)' % (m, k, n, torch.equal(y1, want), (yt.double() - exd).abs().max().item(), (ya.double() - exd).abs().max().item(), exd.abs().max().item()))
eng.close()
