"""Diagnostic: throughput when every step also uploads its packed batch from host memory
(pageable numpy -> HBM through torch), versus the resident-input figure bench.py reports."""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = '3d_multi_pose_estimator_amd'
syn = importlib.import_module(PKG + '.synthetic'); cal = importlib.import_module(PKG + '.calibration')
par = importlib.import_module(PKG + '.parameters'); pipeline = importlib.import_module(PKG + '.pipeline')
calib = cal.Calibration(par.parameters)
eng = pipeline.Engine(par.parameters, calib, max_frames=1000, max_persons_per_camera=4)
eng.load_gat(syn.gat_state_dict(7, 902, logit_gain=25.0, logit_shift=0.948), syn.gat_params(902))
eng.load_mlp(syn.mlp_state_dict(11, 1260))
frames = [{c: [f[c][0], f[c][1]] for c in f} for f in (syn.make_frame(calib, i)[0] for i in range(250))] * 4
pb = eng.pack(frames)
nbytes = sum(getattr(pb, k).nbytes for k in ('frame_head_off', 'frame_en_off', 'slot_cam', 'slot_n', 'head_cam', 'joint_mask', 'tri_mask', 'xy', 'vp'))
db = eng.to_device(pb)
def step(d):
    _, p, n = eng.match(d, want_scores=False)
    return eng.mlp3d(d, p, n)
for _ in range(3): step(db)
torch.cuda.synchronize()
K = 20
t = time.perf_counter()
for _ in range(K): step(db)
torch.cuda.synchronize(); t_res = (time.perf_counter() - t) / K
t = time.perf_counter()
for _ in range(K):
    d = eng.to_device(pb); out = step(d)
    poses = out[0].cpu()
torch.cuda.synchronize(); t_pcie = (time.perf_counter() - t) / K
print('packed batch: %.2f MB for %d frames (%.1f KB/frame); poses out %.2f MB' % (nbytes / 1e6, len(frames), nbytes / len(frames) / 1e3, out[0].numel() * 4 / 1e6))
print('resident inputs : %.3f ms/step  %.0f frames/s' % (t_res * 1e3, len(frames) / t_res))
print('H2D + D2H/step  : %.3f ms/step  %.0f frames/s (synchronous upload, pageable memory)' % (t_pcie * 1e3, len(frames) / t_pcie))
