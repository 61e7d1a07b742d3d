import importlib, json, os, sys, time
import os as _os; _os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')      # before HIP initialises: one hardware queue per busy stream (lib.py leaves the environment alone)
import torch
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT)
PKG = '3d_multi_pose_estimator_amd'
syn = importlib.import_module(PKG + '.synthetic'); cal = importlib.import_module(PKG + '.calibration')
par = importlib.import_module(PKG + '.parameters'); pipeline = importlib.import_module(PKG + '.pipeline')
n, chunk = 16000, 1000
calib = cal.Calibration(par.parameters)
uniq = [syn.make_frame(calib, i, syn.FrameSpec(persons=4))[0] for i in range(200)]
text = json.dumps([uniq[i % 200] for i in range(n)]).encode()
eng = pipeline.Engine(par.parameters, calib, max_frames=chunk, max_persons_per_camera=4)
eng.load_gat(syn.gat_state_dict(7, 902, logit_gain=25.0, logit_shift=0.948), syn.gat_params(902))
eng.load_mlp(syn.mlp_state_dict(11, 1260))
print(open('/sys/fs/cgroup/cpu.max').read().strip() if os.path.exists('/sys/fs/cgroup/cpu.max') else 'no cpu.max', len(os.sched_getaffinity(0)))
for nt in (8, 12, 16, 24, 32, 64, 0):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tot = sum(len(nn) for _, _, nn in eng.stream_json(text, chunk_frames=chunk, n_threads=nt))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('threads', nt, round(n / dt), 'frames/s', round(len(text) / 1e6 / dt), 'MB/s')
