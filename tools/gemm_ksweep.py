"""mpe_linear rate against K at fixed M x N (diagnostic): what a tile's fixed cost (prologue, first
stage in flight, epilogue) takes out of the steady-state loop.  python tools/gemm_ksweep.py [M] [N]"""
import ctypes as C, importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = '3d_multi_pose_estimator_amd'
cal = importlib.import_module(PKG + '.calibration'); par = importlib.import_module(PKG + '.parameters')
pipeline = importlib.import_module(PKG + '.pipeline'); L = importlib.import_module(PKG + '.lib')
eng = pipeline.Engine(par.parameters, cal.Calibration(par.parameters), max_frames=8, max_persons_per_camera=4)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 180000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 400
for k in (96, 192, 416, 832, 1664, 3328):
    w = ((np.random.rand(N, k).astype(np.float32) - 0.5) * 0.1); b = np.random.rand(N).astype(np.float32)
    dw, db, ldw = C.c_void_p(), C.c_void_p(), C.c_int32()
    eng._chk(eng.lib.mpe_upload_linear(eng.ctx, w.ctypes.data_as(L.c_f32p), b.ctypes.data_as(L.c_f32p), N, k, C.byref(dw), C.byref(db), C.byref(ldw)))
    x = torch.rand(M, ldw.value, device='cuda') - 0.5
    ldc = (N + 31) // 32 * 32
    y = torch.empty(M, ldc, device='cuda')
    for flags in (1, 3):
        def run():
            eng._chk(eng.lib.mpe_linear(eng.ctx, eng._stream(), C.c_void_p(x.data_ptr()), ldw.value, dw, ldw.value, db,
                                        C.c_void_p(y.data_ptr()), ldc, M, None, N, k, flags, 0.1))
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        stages = (k + 31) // 32
        print('M=%d N=%d K=%4d (%3d stages) %s  %.3f ms  %.1f TFLOP/s (padded K: %.1f)' % (M, N, k, stages, 'acc64' if flags & 2 else 'f32  ', ms, 2.0 * M * N * k / ms / 1e9, 2.0 * M * N * stages * 32 / ms / 1e9))
    eng.lib.mpe_free_device(eng.ctx, dw); eng.lib.mpe_free_device(eng.ctx, db)
