"""The MLP's error budget on FRESH rows (the tests assert it on the reference's golden rows): MLP input rows of random frames (the engine's own
matching + row assembly), then per network -- the hash MLP of the fixtures and the capture-volume decoder MLP -- the distances of the HIP
path (default mode 3, and modes 4 / 5) and of torch-CPU fp32 (the reference's arithmetic, utils/mlp.py:8-28) from the network evaluated in
float64 with fp32 rounding between layers.  Millimetres after the x10 decode (metrics_from_model.py:281).
    python tests/checkers/mlp_budget_probe.py [frames: 200] [seed: 1] [PANOPTIC|ARPLAB]      -> one JSON line
"""
import importlib, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import oracle_np as onp          # the checker side: torch-CPU fp32 MLP = the reference's arithmetic
PKG = '3d_multi_pose_estimator_amd'
syn = importlib.import_module(PKG + '.synthetic'); cal = importlib.import_module(PKG + '.calibration')
par = importlib.import_module(PKG + '.parameters'); pipeline = importlib.import_module(PKG + '.pipeline')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
preset = sys.argv[3] if len(sys.argv) > 3 else 'PANOPTIC'
P = par.select(preset)
calib = cal.Calibration(P)
rng = np.random.RandomState(seed)
nf = 2 + len(P.used_cameras_skeleton_matching) * len(P.joint_list) * 10
in_dim = len(P.cameras) * len(P.joint_list) * P.numbers_per_joint
frames = [onp.processed_input(syn.make_frame(calib, 100000 * seed + i, syn.FrameSpec(persons=int(rng.randint(1, 6)), noise_px=float(rng.choice([0.0, 1.0, 3.0])),
                                                                                 joint_drop=float(rng.choice([0.0, 0.1, 0.3]))))[0]) for i in range(n)]
eng = pipeline.Engine(P, calib, max_frames=n, max_persons_per_camera=6)
eng.load_gat(syn.gat_state_dict(7, nf, logit_gain=25.0, logit_shift=0.698 + 0.25), syn.gat_params(nf))
nets = {'hash': syn.mlp_state_dict(11, in_dim),
        'room': syn.decoder_mlp_state_dict(len(P.used_cameras), len(P.joint_list), P.numbers_per_joint, noise_seed=3, noise_bound=0.01)}       # (the fixtures' room MLP: tests/golden/meta.json)
eng.load_mlp(nets['hash'])
db = eng.to_device(eng.pack(frames))
_, persons, n_persons = eng.match(db, want_scores=False)
rows, valid = eng.mlp_input_rows(db, persons, n_persons)
eng.sync_status()
x = rows[valid.bool()].contiguous().cpu()
eng.close()


def exact(x, w):
    keys = sorted({int(k.split('.')[1]) for k in w})
    h = x.double()
    for i, k in enumerate(keys):
        h = h @ torch.from_numpy(np.asarray(w['layers.%d.weight' % k])).double().T + torch.from_numpy(np.asarray(w['layers.%d.bias' % k])).double()
        if i != len(keys) - 1:
            h = torch.nn.functional.leaky_relu(h, 0.1)
        h = h.float().double()
    return h


rep = {'preset': preset, 'seed': seed, 'frames': n, 'rows': int(x.shape[0])}
for name, w in nets.items():
    ex = exact(x, w)
    ref = onp.mlp_forward(w, x).double()
    keep = (ex.abs().amax(dim=1) * 10.0 <= 5.0) if name == 'room' else torch.ones(x.shape[0], dtype=torch.bool)     # the capture volume: |pose| <= 5 m
    r = {'rows_kept': int(keep.sum()), 'output_scale': float(ex[keep].abs().max()) if keep.any() else 0.0,
         'ref_vs_exact_mm': float((ref - ex)[keep].abs().max()) * 1e4 if keep.any() else None}
    for mode, kw in (('default', {}), ('max_accuracy', {'mlp_max_accuracy': True}), ('f64', {'mlp_f64': True})):
        e2 = pipeline.Engine(P, calib, max_frames=max(8, (x.shape[0] + 5) // 6), max_persons_per_camera=6)
        e2.load_mlp(w)
        if kw:
            e2.set_precision(**kw)
        g = e2.mlp_forward(x.cuda()).cpu().double()
        e2.close()
        if keep.any():
            eg, er = (g - ex)[keep].abs().amax(dim=1), (ref - ex)[keep].abs().amax(dim=1)
            r[mode] = {'gpu_vs_exact_mm': float(eg.max()) * 1e4, 'gpu_vs_ref_mm': float((g - ref)[keep].abs().max()) * 1e4,
                       'rows_gpu_closer_or_equal': int((eg <= er).sum()), 'worst_row_ratio_gpu_over_ref': float((eg / er.clamp_min(1e-12)).max())}
    rep[name] = r
print(json.dumps(rep))
