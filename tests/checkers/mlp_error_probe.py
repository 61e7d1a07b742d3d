"""Diagnostic (not a test): error of the MLP chain vs a per-layer-rounded fp64 reference for
torch-CPU fp32 (the oracle), the GPU fp32-chain GEMM and the GPU f64-running-sum GEMM."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import oracle_np as onp
PKG = '3d_multi_pose_estimator_amd'
syn = importlib.import_module(PKG + '.synthetic'); cal = importlib.import_module(PKG + '.calibration')
par = importlib.import_module(PKG + '.parameters'); pipeline = importlib.import_module(PKG + '.pipeline')
arr = np.load(os.path.join(ROOT, 'tests/golden/c4_5x10.npz'))
sd = syn.mlp_state_dict(11, 1260)
x = torch.from_numpy(arr['f0_mlp_in'])
keys = sorted({int(k.split('.')[1]) for k in sd})
def exact_chain(x):
    h = x.double(); outs = []
    for n, k in enumerate(keys):
        W = torch.from_numpy(sd['layers.%d.weight' % k]).double(); b = torch.from_numpy(sd['layers.%d.bias' % k]).double()
        h = h @ W.T + b
        if n != len(keys) - 1: h = torch.nn.functional.leaky_relu(h, 0.1)
        h = h.float().double(); outs.append(h)
    return outs
R = exact_chain(x)[-1]
y_cpu = onp.mlp_forward(sd, x).double()
calib = cal.Calibration(par.parameters)
eng = pipeline.Engine(par.parameters, calib, max_frames=8, max_persons_per_camera=10)
eng.load_mlp(sd)
res = {}
for name, acc in (('gpu_f32chain', False), ('gpu_acc64', True)):
    eng.set_precision(False, acc)
    res[name] = eng.mlp_forward(x.cuda()).cpu().double()
print('rows', x.shape[0], 'out scale', R.abs().max().item())
print('cpu(torch f32) vs exact :', (y_cpu - R).abs().max().item())
for k, v in res.items():
    print('%-14s vs exact : %.3e   vs cpu: %.3e' % (k, (v - R).abs().max().item(), (v - y_cpu).abs().max().item()))
# single-layer error (layer 1 only, K=1260 and layer 2, K=3072)
for li, kk in enumerate(keys[:2]):
    W = sd['layers.%d.weight' % kk]; b = sd['layers.%d.bias' % kk]
    xin = x if li == 0 else torch.randn(x.shape[0], W.shape[1])
    ref = (xin.double() @ torch.from_numpy(W).double().T + torch.from_numpy(b).double())
    cpu = torch.nn.functional.linear(xin, torch.from_numpy(W), torch.from_numpy(b)).double()
    g0 = eng.linear(xin.cuda(), W, b, None, acc64=False).cpu().double()
    g1 = eng.linear(xin.cuda(), W, b, None, acc64=True).cpu().double()
    u = ref.abs().mean().item()
    print('layer %d K=%d: mean|y|=%.3f  err/mean|y| cpu %.2e  gpu chain %.2e  gpu acc64 %.2e (eps=6e-8)' % (
        li, W.shape[1], u, (cpu - ref).abs().max().item() / u, (g0 - ref).abs().max().item() / u, (g1 - ref).abs().max().item() / u))
    print('           rms: cpu %.2e  chain %.2e  acc64 %.2e' % ((cpu-ref).pow(2).mean().sqrt().item()/u, (g0-ref).pow(2).mean().sqrt().item()/u, (g1-ref).pow(2).mean().sqrt().item()/u))
