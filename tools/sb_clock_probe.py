"""In-kernel shader clock of the split-bf16 tile kernel's launches (MI355X_MICROARCH.md, DVFS give-back item 6): a diagnostic build
(make -C 3d_multi_pose_estimator_amd/csrc exp EXPFLAGS=-DMPE_SB_CLOCK; MPE_LIB_VARIANT=exp) stamps s_memtime / s_memrealtime around the
tile loop of the first MFMA wave of every workgroup, one bucket per launch shape.  A workload runs back to back for SECONDS, then the
stamps of the last pass are read: clock = d(memtime) / d(memrealtime) x 100 MHz, median over the workgroups.
   MPE_LIB_VARIANT=exp python tools/sb_clock_probe.py mlp [seconds] [zero]     an MLP of 4004 x 3072-wide layers
   MPE_LIB_VARIANT=exp python tools/sb_clock_probe.py gat [seconds]            the GAT forward of 1000 frames of 5 x 4
   MPE_LIB_VARIANT=exp python tools/sb_clock_probe.py step [seconds]           the whole step (match + MLP 3D), one stream"""
import ctypes as C, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import env, pkg
what = sys.argv[1] if len(sys.argv) > 1 else 'mlp'
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
zero = len(sys.argv) > 3 and sys.argv[3] == 'zero'
e = env('panoptic')
syn = pkg('synthetic')
eng = pkg('pipeline').Engine(e.params, e.calib, max_frames=1001, max_persons_per_camera=4)
g = torch.Generator().manual_seed(3)
shapes = {}
if what == 'mlp':
    dims = [1260, 3072, 3072, 3072, 3072, 54]
    sd = {}
    for i in range(len(dims) - 1):
        w = torch.randn(dims[i + 1], dims[i], generator=g) / np.sqrt(dims[i])
        sd['layers.%d.weight' % (2 * i + 1)] = (w * 0 if zero else w).numpy()
        sd['layers.%d.bias' % (2 * i + 1)] = torch.randn(dims[i + 1], generator=g).numpy() * 0.1
        shapes[(dims[i + 1], (dims[i] + 31) // 32 * 32)] = 'MLP %d -> %d (4004 rows)' % (dims[i], dims[i + 1])
    eng.load_mlp(sd)
    x = torch.randn(4004, 1260, generator=g).cuda()
    run = lambda: eng.mlp_forward(x)
else:
    V, J = 5, 18
    nf = 2 + V * J * 10
    eng.load_gat(syn.gat_state_dict(7, nf, logit_gain=25.0, logit_shift=0.698 + 0.25), syn.gat_params(nf))
    eng.load_mlp(syn.mlp_state_dict(11, V * J * 14))
    frames = []
    for i in range(250):
        f, _ = syn.make_frame(e.calib, i, syn.FrameSpec(persons=4))
        frames.append({c: [f[c][0], f[c][1]] for c in f})
    db = eng.to_device(eng.pack([frames[i % 250] for i in range(1000)]))
    for d, h, o in syn.gat_layer_dims(nf)[1:]:
        shapes[(d, (d + 31) // 32 * 32)] = 'GAT fc1 %d -> %d (180 000 rows)' % (d, d)
        shapes[(h * o, (d + 31) // 32 * 32)] = 'GAT fc2 %d -> %d (180 000 rows)' % (d, h * o)
    d0, h0, o0 = syn.gat_layer_dims(nf)[0]
    shapes[(h0 * o0, (d0 + 31) // 32 * 32)] = 'GAT layer-0 fc2 %d -> %d (20 000 head rows, f64 sums)' % (d0, h0 * o0)
    for i, o in syn.mlp_layer_dims(V * J * 14):
        shapes[(o, (i + 31) // 32 * 32)] = 'MLP %d -> %d (4000 rows)' % (i, o)
    if what == 'gat':
        run = lambda: eng.gat_scores(db)
    else:
        def run():
            _, persons, n_persons = eng.match(db, want_scores=False)
            return eng.mlp3d(db, persons, n_persons)
for _ in range(3):
    run()
torch.cuda.synchronize()
t0 = time.time(); n = 0
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
while time.time() - t0 < secs:
    for _ in range(10):
        run()
    n += 10
    torch.cuda.synchronize()
ev1.record(); torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1) / n
out = (C.c_ulonglong * (16 * 256 * 4))()
rc = eng.lib.mpe_debug_sb_stamps(out, 0)
a = np.frombuffer(out, dtype=np.uint64).reshape(16, 256, 4).astype(np.float64)
print('%s%s: %d passes, %.3f ms per pass (stamps rc %d)' % (what, ' zero weights' if zero else '', n, ms, rc))
bucket_of = {}
for (nn, kp), name in shapes.items():
    bucket_of.setdefault((nn ^ (kp >> 5)) & 15, []).append(name)
for b in range(16):
    dt, dr = a[b, :, 2] - a[b, :, 0], a[b, :, 3] - a[b, :, 1]
    ok = (dr > 0) & (dt > 0)
    if not ok.any():
        continue
    clk = dt[ok] / dr[ok] * 100e6
    print('  bucket %2d  %-70s %3d workgroups  clock median %.3f GHz (min %.3f max %.3f)  tile loop %.1f us median'
          % (b, ' | '.join(bucket_of.get(b, ['?'])), int(ok.sum()), np.median(clk) / 1e9, clk.min() / 1e9, clk.max() / 1e9, np.median(dr[ok]) / 100.0))
eng.close()
