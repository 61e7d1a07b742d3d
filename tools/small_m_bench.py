"""Kernel time of mpe_linear at small M (diagnostic): run once with MPE_SKINNY_WAVES=0 (tile kernel
only) and once with MPE_SKINNY_WAVES=1000000 (wave-per-tile kernel only) to place the switch-over."""
import ctypes as C, importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = '3d_multi_pose_estimator_amd'
cal = importlib.import_module(PKG + '.calibration'); par = importlib.import_module(PKG + '.parameters')
pipeline = importlib.import_module(PKG + '.pipeline'); L = importlib.import_module(PKG + '.lib')
eng = pipeline.Engine(par.parameters, cal.Calibration(par.parameters), max_frames=8, max_persons_per_camera=4)
shapes = [(400, 400, 1, 'gat 400'), (902, 400, 1, 'gat L0'), (3072, 3072, 3, 'mlp 3072'), (1024, 1024, 3, 'mlp 1024'),
          (1024, 54, 2, 'mlp 54')]
ms_ = [4, 16, 64, 180, 360, 720, 1440, 2880, 5000]
print('MPE_SKINNY_WAVES=%s' % os.environ.get('MPE_SKINNY_WAVES'))
for k, n, flags, name in shapes:
    w = (np.random.rand(n, k).astype(np.float32) - 0.5); b = np.random.rand(n).astype(np.float32)
    dw, db, ldw = C.c_void_p(), C.c_void_p(), C.c_int32()
    eng._chk(eng.lib.mpe_upload_linear(eng.ctx, w.ctypes.data_as(L.c_f32p), b.ctypes.data_as(L.c_f32p), n, k, C.byref(dw), C.byref(db), C.byref(ldw)))
    out = []
    for m in ms_:
        x = torch.rand(m, ldw.value, device='cuda') - 0.5
        ldc = (n + 31) // 32 * 32
        y = torch.empty(m, ldc, device='cuda')
        def run():
            eng._chk(eng.lib.mpe_linear(eng.ctx, eng._stream(), C.c_void_p(x.data_ptr()), ldw.value, dw, ldw.value, db,
                                        C.c_void_p(y.data_ptr()), ldc, m, None, n, k, flags, 0.1))
        for _ in range(5): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        e0.record()
        for _ in range(reps): run()
        e1.record(); torch.cuda.synchronize()
        out.append('%d:%.1f' % (m, e0.elapsed_time(e1) / reps * 1e3))
    print('%-9s us per launch by M  ' % name + '  '.join(out))
    eng.lib.mpe_free_device(eng.ctx, dw); eng.lib.mpe_free_device(eng.ctx, db)
