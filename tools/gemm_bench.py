"""Micro-benchmark of mpe_linear on the shapes of the path (diagnostic tool, not a test)."""
import ctypes as C, importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = '3d_multi_pose_estimator_amd'
cal = importlib.import_module(PKG + '.calibration'); par = importlib.import_module(PKG + '.parameters')
pipeline = importlib.import_module(PKG + '.pipeline'); L = importlib.import_module(PKG + '.lib')
eng = pipeline.Engine(par.parameters, cal.Calibration(par.parameters), max_frames=8, max_persons_per_camera=4)
shapes = [(180000, 400, 400, 'gat fc1 400'), (180000, 400, 320, 'gat fc2 320'), (180000, 320, 320, 'gat 320'),
          (180000, 320, 150, 'gat fc2 150'), (180000, 150, 150, 'gat 150'), (180000, 150, 1, 'gat fc2 1'),
          (20000, 902, 902, 'gat L0 fc1'), (20000, 902, 400, 'gat L0 fc2'),
          (4004, 1260, 3072, 'mlp 1'), (4004, 3072, 3072, 'mlp 2'), (4004, 3072, 2048, 'mlp 3'),
          (4004, 2048, 2048, 'mlp 4'), (4004, 2048, 1024, 'mlp 5'), (4004, 1024, 1024, 'mlp 6'), (4004, 1024, 54, 'mlp 9')]
if len(sys.argv) > 1:
    shapes = [s for s in shapes if any(a in s[3] for a in sys.argv[1:])]
for m, k, n, name in shapes:
    w = (np.random.rand(n, k).astype(np.float32) - 0.5); b = np.random.rand(n).astype(np.float32)
    dw, db, ldw = C.c_void_p(), C.c_void_p(), C.c_int32()
    eng._chk(eng.lib.mpe_upload_linear(eng.ctx, w.ctypes.data_as(L.c_f32p), b.ctypes.data_as(L.c_f32p), n, k, C.byref(dw), C.byref(db), C.byref(ldw)))
    x = torch.rand(m, ldw.value, device='cuda') - 0.5
    ldc = (n + 31) // 32 * 32
    y = torch.empty(m, ldc, device='cuda')
    for flags in (1, 3):
        def run():
            eng._chk(eng.lib.mpe_linear(eng.ctx, eng._stream(), C.c_void_p(x.data_ptr()), ldw.value, dw, ldw.value, db,
                                        C.c_void_p(y.data_ptr()), ldc, m, None, n, k, flags, 0.1))
        for _ in range(3 if flags & 2 else 40): run()          # the first launches of a process see the clock still ramping
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print('%-12s M=%6d K=%4d N=%4d %s  %.3f ms  %.1f TFLOP/s' % (name, m, k, n, 'acc64' if flags & 2 else 'f32  ', ms, 2.0 * m * n * k / ms / 1e9))
    eng.lib.mpe_free_device(eng.ctx, dw); eng.lib.mpe_free_device(eng.ctx, db)
