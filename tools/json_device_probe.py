"""Where the time of the device-side JSON parse goes (diagnostic): host staging, H2D, the three kernels.
    python tools/json_device_probe.py [frames] [chunk]"""
import importlib, json, os, sys, time
import os as _os; _os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')      # before HIP initialises: one hardware queue per busy stream (lib.py leaves the environment alone)
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = '3d_multi_pose_estimator_amd'
syn = importlib.import_module(PKG + '.synthetic'); cal = importlib.import_module(PKG + '.calibration')
par = importlib.import_module(PKG + '.parameters'); pipeline = importlib.import_module(PKG + '.pipeline')
packing = importlib.import_module(PKG + '.packing')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
calib = cal.Calibration(par.parameters)
uniq = [syn.make_frame(calib, i, syn.FrameSpec(persons=4))[0] for i in range(200)]
text = json.dumps([uniq[i % 200] for i in range(n)]).encode()
eng = pipeline.Engine(par.parameters, calib, max_frames=chunk, max_persons_per_camera=4)
bufs = eng.json_device_buffers(chunk)
index = packing.JsonIndex(text)
time.sleep(1.0)                      # let the background scan finish: staging time alone
for rep in range(2):
    for w in range(n // chunk):
        t0 = time.perf_counter()
        nf, ne, used = packing.stage_json_window(index, par.parameters, bufs['host'], frame_start=w * chunk, max_frames=chunk)
        t1 = time.perf_counter()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        bufs['dev'].buf[:used].copy_(bufs['host'].buf[:used], non_blocking=True)
        e1.record()
        eng.parse_json_device(bufs, nf, ne, used)
        e2.record()
        pd = eng.finish_parse(bufs)
        t2 = time.perf_counter()
        if rep:
            print('window %d: stage %.2f ms (%d entries, %.1f MB staged), H2D %.2f ms, parse kernels %.2f ms, wall after stage %.2f ms, heads %d'
                  % (w, 1e3 * (t1 - t0), ne, used / 1e6, e0.elapsed_time(e1), e1.elapsed_time(e2), 1e3 * (t2 - t1), pd.n_heads))
t0 = time.perf_counter()
ix2 = packing.JsonIndex(text)
import ctypes as C
bufs2 = eng.json_device_buffers(chunk)
packing.stage_json_window(ix2, par.parameters, bufs2['host'], frame_start=n - chunk, max_frames=chunk)   # waits for the whole scan
print('frame scan of %.0f MB: %.1f ms' % (len(text) / 1e6, 1e3 * (time.perf_counter() - t0)))

# ---- the same parse on a side stream while the default stream computes ----
eng.load_gat(syn.gat_state_dict(7, 902, logit_gain=25.0, logit_shift=0.948), syn.gat_params(902))
eng.load_mlp(syn.mlp_state_dict(11, 1260))
pd = eng.finish_parse(bufs) if False else None
nf, ne, used = packing.stage_json_window(index, par.parameters, bufs['host'], frame_start=0, max_frames=chunk)
eng.parse_json_device(bufs, nf, ne, used)
db = eng.finish_parse(bufs)
for _ in range(3):
    _, persons, n_persons = eng.match(db, want_scores=False); eng.mlp3d(db, persons, n_persons)
torch.cuda.synchronize()
bufs_b = eng.json_device_buffers(chunk)
packing.stage_json_window(index, par.parameters, bufs_b['host'], frame_start=chunk, max_frames=chunk)
for prio in (0, -1):
    side = torch.cuda.Stream(priority=prio)
    for rep in range(2):
        for _ in range(4):
            _, persons, n_persons = eng.match(db, want_scores=False); eng.mlp3d(db, persons, n_persons)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        with torch.cuda.stream(side):
            ev[0].record()
            bufs_b['dev'].buf[:used].copy_(bufs_b['host'].buf[:used], non_blocking=True)
            ev[1].record()
            eng.parse_json_device(bufs_b, nf, ne, used)
            ev[2].record()
        torch.cuda.synchronize()
    print('side stream priority %d, default stream busy with 4 compute steps: H2D %.2f ms, parse kernels %.2f ms' % (prio, ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])))
