"""One 5 x 4 frame per engine call with K contexts in flight: plain stream launches against the replay of one captured HIP graph per
context (torch.cuda.CUDAGraph around mpe_match_batch + mpe_mlp3d_batch on the context's own stream).  With the latency launches of
round 6 a frame is ~26 launches: one host thread issues them in ~110 us, which caps three or more contexts in flight; a replay costs
the host one call.  usage: python tools/graph_contexts.py [frames per call]"""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
PKG = '3d_multi_pose_estimator_amd'
syn = importlib.import_module(PKG + '.synthetic'); cal = importlib.import_module(PKG + '.calibration')
par = importlib.import_module(PKG + '.parameters'); pipeline = importlib.import_module(PKG + '.pipeline')
calib = cal.Calibration(par.parameters)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
eng0 = pipeline.Engine(par.parameters, calib, max_frames=B, max_persons_per_camera=4)
eng0.load_gat(syn.gat_state_dict(7, 902, logit_gain=25.0, logit_shift=0.698), syn.gat_params(902))
eng0.load_mlp(syn.mlp_state_dict(11, 1260))
frames = []
for i in range(B):
    f = syn.make_frame(calib, 100 + i)[0]
    frames.append({c: [f[c][0], f[c][1]] for c in f})

def bench(K, use_graph, n=600):
    engs = eng0.contexts(K)
    streams = [torch.cuda.Stream() for _ in engs]
    dbs = [e.to_device(e.pack(frames)) for e in engs]
    outs = [None] * K
    def step(k):
        _, persons, n_persons = engs[k].match(dbs[k], want_scores=False)
        return engs[k].mlp3d(dbs[k], persons, n_persons)
    graphs = []
    for k in range(K):
        with torch.cuda.stream(streams[k]):
            for _ in range(3):
                outs[k] = step(k)
        torch.cuda.synchronize()
        if use_graph:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=streams[k]):
                outs[k] = step(k)
            graphs.append(g)
    def run(i):
        k = i % K
        if use_graph:
            with torch.cuda.stream(streams[k]):
                graphs[k].replay()
        else:
            with torch.cuda.stream(streams[k]):
                outs[k] = step(k)
    for i in range(40): run(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): run(i)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    return dt, outs[0][0].clone()

ref = None
for K in (1, 2, 3, 4, 6, 8):
    for g in (False, True):
        dt, poses = bench(K, g)
        if ref is None:
            ref = poses
        print('%d frame(s) per call, %d context(s) in flight, %-14s %7.1f us per call  %8.0f frames/s  same poses: %s'
              % (B, K, 'graph replay:' if g else 'stream launches:', dt * 1e6, B / dt, bool(torch.equal(poses, ref))))
