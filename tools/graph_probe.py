"""Diagnostic: single-frame step latency with the launches replayed from a captured HIP graph
(torch.cuda.CUDAGraph around the C-ABI calls) against plain stream launches."""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = '3d_multi_pose_estimator_amd'
syn = importlib.import_module(PKG + '.synthetic'); cal = importlib.import_module(PKG + '.calibration')
par = importlib.import_module(PKG + '.parameters'); pipeline = importlib.import_module(PKG + '.pipeline')
calib = cal.Calibration(par.parameters)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
eng = pipeline.Engine(par.parameters, calib, max_frames=B, max_persons_per_camera=4)
eng.load_gat(syn.gat_state_dict(7, 902, logit_gain=25.0, logit_shift=0.698), syn.gat_params(902))
eng.load_mlp(syn.mlp_state_dict(11, 1260))
frames = []
for i in range(B):
    f = syn.make_frame(calib, 100 + i)[0]
    frames.append({c: [f[c][0], f[c][1]] for c in f})
db = eng.to_device(eng.pack(frames))

def step():
    _, persons, n_persons = eng.match(db, want_scores=False)
    return eng.mlp3d(db, persons, n_persons)

def timeit(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

ref = step()[0].clone()
print('stream launches: %.3f ms/step' % timeit(step))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    out = step()
g.replay(); torch.cuda.synchronize()
print('graph replay equals stream result:', bool(torch.equal(out[0], ref)))
print('graph replay:    %.3f ms/step' % timeit(g.replay))
