"""Soak of the two-context engine mode: Engine.stream_json(contexts=2) and Engine.run_pipelined(contexts=2) over and over,
every result compared bit for bit with the first one (a race between windows / contexts / buffer slots would show as a
difference).  python tools/soak_contexts.py [iterations] [frames per window / batch: 64]
(500-frame batches put every GEMM of the step on the tile kernels, twelve-wave form included, with two or three steps in flight)"""
import importlib, json, os, sys, time
import os as _os; _os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')      # before HIP initialises: one hardware queue per busy stream (lib.py leaves the environment alone)
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = '3d_multi_pose_estimator_amd'
syn = importlib.import_module(PKG + '.synthetic'); cal = importlib.import_module(PKG + '.calibration')
par = importlib.import_module(PKG + '.parameters'); pipeline = importlib.import_module(PKG + '.pipeline')
n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 200
WF = int(sys.argv[2]) if len(sys.argv) > 2 else 64
calib = cal.Calibration(par.parameters)
eng = pipeline.Engine(par.parameters, calib, max_frames=WF, max_persons_per_camera=6)
eng.load_gat(syn.gat_state_dict(7, 902, logit_gain=25.0, logit_shift=0.948), syn.gat_params(902))
eng.load_mlp(syn.mlp_state_dict(11, 1260))
frames = [syn.make_frame(calib, 500 + i, syn.FrameSpec(persons=1 + i % 5))[0] for i in range(WF * 9 + 17)]
text = json.dumps(frames).encode()
ref = [(p.copy(), n.copy()) for _, p, n in eng.stream_json(text, chunk_frames=WF, parser='host')]
t0 = time.perf_counter()
for it in range(n_it):
    got = [(p.copy(), n.copy()) for _, p, n in eng.stream_json(text, chunk_frames=WF, contexts=2)]
    assert len(got) == len(ref)
    for (p1, n1), (p2, n2) in zip(ref, got):
        assert np.array_equal(n1, n2), it
        for f in range(len(n1)):
            assert np.array_equal(p1[f, :n1[f]], p2[f, :n1[f]]), (it, f)
print('stream_json(contexts=2): %d passes over %d windows identical to the host-parsed reference, %.1f s' % (n_it, len(ref), time.perf_counter() - t0))
batches = [eng.to_device(eng.pack([{c: [f[c][0], f[c][1]] for c in f} for f in frames[i:i + WF * 3 // 4]])) for i in range(0, WF * 15 // 2, WF * 3 // 4)]
want = []
for db in batches:
    _, persons, n_persons = eng.match(db, want_scores=False)
    want.append((eng.mlp3d(db, persons, n_persons)[0].cpu().numpy(), n_persons.cpu().numpy()))
t0 = time.perf_counter()
for it in range(n_it):
    for (w, wn), (p, n, q, _) in zip(want, eng.run_pipelined(batches, contexts=2 + it % 2)):
        assert np.array_equal(wn, n.cpu().numpy()), it
        pn = p.cpu().numpy()
        for f in range(len(wn)):
            assert np.array_equal(w[f, :wn[f]], pn[f, :wn[f]]), (it, f)
print('run_pipelined(contexts=2|3): %d passes over %d batches identical to the plain call sequence, %.1f s' % (n_it, len(batches), time.perf_counter() - t0))
eng.close()
