"""Throughput with K engines (own context, workspace and stream each) working on K shares of the
1000-frame batch at once on ONE GPU: concurrent kernels fill the tails of the dependent launch chain
and put the memory-bound attention under the MFMA-bound GEMMs of the other share.

    python tools/two_engines.py [K ...]      (default: 1 2 3 4)
"""
import importlib, json, os, sys, time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = '3d_multi_pose_estimator_amd'
syn = importlib.import_module(PKG + '.synthetic'); cal = importlib.import_module(PKG + '.calibration')
par = importlib.import_module(PKG + '.parameters'); pipeline = importlib.import_module(PKG + '.pipeline')
calib = cal.Calibration(par.parameters)
total, steps = 1000, 200
gat = syn.gat_state_dict(7, 902, logit_gain=25.0, logit_shift=0.948); prm = syn.gat_params(902)
mlp = syn.mlp_state_dict(11, 1260)
uniq = []
for i in range(250):
    f = syn.make_frame(calib, i, syn.FrameSpec(persons=4))[0]
    uniq.append({c: [f[c][0], f[c][1]] for c in f})
frames = [uniq[i % 250] for i in range(total)]
out = {}
for K in [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4]:
    per = total // K
    engs, dbs, streams = [], [], []
    for k in range(K):
        e = pipeline.Engine(par.parameters, calib, max_frames=per, max_persons_per_camera=4)
        e.load_gat(gat, prm); e.load_mlp(mlp)
        engs.append(e); dbs.append(e.to_device(e.pack(frames[k * per:(k + 1) * per]))); streams.append(torch.cuda.Stream())

    two = bool(os.environ.get('MPE_TWO_STREAMS'))           # every engine with its own matching / 3D stream pair, as bench.py's default mode
    streams3d = [torch.cuda.Stream() for _ in range(K)] if two else streams

    def step():
        res = []
        for e, db, s, s3 in zip(engs, dbs, streams, streams3d):
            with torch.cuda.stream(s):
                _, persons, n_persons = e.match(db, want_scores=False)
                ev = torch.cuda.Event(); ev.record(s)
            with torch.cuda.stream(s3):
                s3.wait_event(ev)
                res.append(e.mlp3d(db, persons, n_persons)[0])
            for t_ in (persons, n_persons):
                t_.record_stream(s3)
        return res
    alt = bool(os.environ.get('MPE_ALTERNATE'))             # every engine holds the WHOLE batch; step i runs on engine i % K
    if alt:
        for e in engs:
            e.close()
        engs, dbs = [], []
        for k in range(K):
            e = pipeline.Engine(par.parameters, calib, max_frames=total, max_persons_per_camera=4)
            e.load_gat(gat, prm); e.load_mlp(mlp)
            engs.append(e); dbs.append(e.to_device(e.pack(frames)))
        per = total // K                                    # so that the frames/s formula below counts `total` frames per step
        counter = [0]

        def step():
            k = counter[0] % K
            counter[0] += 1
            if two:
                with torch.cuda.stream(streams[k]):
                    _, persons, n_persons = engs[k].match(dbs[k], want_scores=False)
                    ev = torch.cuda.Event(); ev.record(streams[k])
                with torch.cuda.stream(streams3d[k]):
                    streams3d[k].wait_event(ev)
                    out = [engs[k].mlp3d(dbs[k], persons, n_persons)[0]]
                for t_ in (persons, n_persons):
                    t_.record_stream(streams3d[k])
                return out
            with torch.cuda.stream(streams[k]):
                _, persons, n_persons = engs[k].match(dbs[k], want_scores=False)
                return [engs[k].mlp3d(dbs[k], persons, n_persons)[0]]
    for _ in range(10):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        keep = step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    out[K] = {'frames_per_s': per * K * steps / dt, 'ms_per_1000_frames': 1e3 * dt / steps * (1000.0 / (per * K))}
    print(K, 'engines x', per, 'frames:', round(out[K]['frames_per_s']), 'frames/s')
    for e in engs:
        e.close()
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'two_engines.json'), 'w'), indent=1)
