"""Throughput from frame-JSON BYTES to 3D poses in host memory: the reference
parses every frame twice in Python (json.load + json.loads per camera) before any numeric work.

    python tools/json_to_poses.py [frames] [chunk]

Three forms on the same synthetic document (5 views x 4 persons, reference wire format incl. the
bodies_3D ground truth the path does not need):
  python   json.load + Python packer + pageable upload + device path        (what a port would do)
  native   mpe_pack_json (C++, all cores) + pageable upload + device path
  stream   Engine.stream_json: mpe_pack_json_into a page-locked arena, one H2D copy per chunk,
           the parse of chunk i+1 overlapped with the device work of chunk i
Writes gpurun_out/json_to_poses.json."""
import importlib, json, os, sys, time
import os as _os; _os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')      # before HIP initialises: one hardware queue per busy stream (lib.py leaves the environment alone)
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = '3d_multi_pose_estimator_amd'
syn = importlib.import_module(PKG + '.synthetic'); cal = importlib.import_module(PKG + '.calibration')
par = importlib.import_module(PKG + '.parameters'); pipeline = importlib.import_module(PKG + '.pipeline')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
calib = cal.Calibration(par.parameters)
uniq = [syn.make_frame(calib, i, syn.FrameSpec(persons=4))[0] for i in range(200)]
text = json.dumps([uniq[i % 200] for i in range(n)]).encode()
eng = pipeline.Engine(par.parameters, calib, max_frames=chunk, max_persons_per_camera=4)
eng.load_gat(syn.gat_state_dict(7, 902, logit_gain=25.0, logit_shift=0.948), syn.gat_params(902))
eng.load_mlp(syn.mlp_state_dict(11, 1260))


def device_path(pb):
    db = eng.to_device(pb)
    _, persons, n_persons = eng.match(db, want_scores=False)
    poses = eng.mlp3d(db, persons, n_persons)[0]
    return poses.cpu().numpy(), n_persons.cpu().numpy()


def run_python():
    frames = json.loads(text)
    tot = 0
    for s in range(0, n, chunk):
        fr = [{c: [f[c][0], f[c][1]] for c in f} for f in frames[s:s + chunk]]
        tot += device_path(eng.pack(fr))[1].shape[0]
    return tot


def run_native():
    tot = 0
    for s in range(0, n, chunk):
        tot += device_path(eng.pack_json(text, frame_start=s, max_frames=chunk))[1].shape[0]
    return tot


def run_stream():
    return sum(len(nn) for _, _, nn in eng.stream_json(text, chunk_frames=chunk))


out = {'frames': n, 'chunk': chunk, 'json_mb': len(text) / 1e6, 'host_threads': os.cpu_count()}
for name, fn in (('python', run_python), ('native', run_native), ('stream', run_stream)):
    if name == 'python' and n > 2000:
        reps = 1
    else:
        fn()                                   # warm up
        reps = 2
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        assert fn() == n
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    out[name] = {'frames_per_s': n / dt, 'json_mb_per_s': len(text) / 1e6 / dt}
    print(name, out[name])
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'json_to_poses.json'), 'w'), indent=1)
