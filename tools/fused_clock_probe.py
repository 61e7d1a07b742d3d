"""Where a workgroup of k_gat_fused spends its life (diagnostic build: make -C 3d_multi_pose_estimator_amd/csrc exp EXPFLAGS=-DMPE_FUSED_CLOCK;
MPE_LIB_VARIANT=exp python tools/fused_clock_probe.py): the GAT forward of 1000 frames of 5 x 4, per workgroup the time between its
phase boundaries (100 MHz clock, thread 0)."""
import ctypes as C, importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = '3d_multi_pose_estimator_amd'
syn = importlib.import_module(PKG + '.synthetic'); cal = importlib.import_module(PKG + '.calibration')
par = importlib.import_module(PKG + '.parameters'); pipeline = importlib.import_module(PKG + '.pipeline'); L = importlib.import_module(PKG + '.lib')
calib = cal.Calibration(par.parameters)
B = 1000
eng = pipeline.Engine(par.parameters, calib, max_frames=B, max_persons_per_camera=4)
eng.load_gat(syn.gat_state_dict(7, 902, logit_gain=25.0, logit_shift=0.698), syn.gat_params(902))
uniq = []
for i in range(50):
    f = syn.make_frame(calib, 100 + i)[0]
    uniq.append({c: [f[c][0], f[c][1]] for c in f})
db = eng.to_device(eng.pack([uniq[i % 50] for i in range(B)]))
lib = eng.lib
buf = (C.c_uint * (16384 * 4))()
names = ['issue of loads + table values landed', 'softmax phase (image landing underneath)', 'rest of the landing + barrier', 'phase 3 (weighted sums, stores issued)']
# one attention layer at a time (mpe_gat_layer runs fc1, fc2 and the attention stage of ONE layer): the 40-wide layer 1 is the production shape
x = torch.randn(db.n_heads + db.n_edge_nodes, 400, device='cuda') * 0.1
for layer, width in ((1, 400), (3, 320)):
    xin = x[:, :width].contiguous()
    for _ in range(3):
        eng.gat_layer(db, layer, xin)
    torch.cuda.synchronize()
    assert lib.mpe_debug_fused_stamps(buf, 1) == 0
    eng.gat_layer(db, layer, xin)
    torch.cuda.synchronize()
    assert lib.mpe_debug_fused_stamps(buf, 0) == 0
    v = np.frombuffer(buf, dtype=np.uint32).reshape(16384, 4).astype(np.float64)
    v = v[v.sum(axis=1) > 0]
    print('layer %d: %d workgroups sampled' % (layer, len(v)))
    tot = v.sum(axis=1).mean()
    for i, nm in enumerate(names):
        print('  %-44s %7.2f us per workgroup  %5.1f %%' % (nm, v[:, i].mean() / 100.0, 100.0 * v[:, i].mean() / tot))
    print('  %-44s %7.2f us per workgroup' % ('sum (workgroup lifetime seen by thread 0)', tot / 100.0))
