"""GPU parity tests: the HIP path (through the C ABI) against the golden fixtures produced by
the reference and against the CPU oracle on the same seeded inputs."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import ALL_CASES, ALL_CASES_FZ, CASES, GOLDEN, ROOT, env, load_case, oracle, pkg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def engine(calib, gat_weights, mlp_weights):
    pipeline = pkg('pipeline')
    eng = pipeline.Engine(calib.params, calib, max_frames=64, max_persons_per_camera=10)
    sd, prm = gat_weights
    eng.load_gat(sd, prm)
    eng.load_mlp(mlp_weights)
    yield eng
    eng.close()


_engines = {}


def engine_for(variant):
    """One engine per fixture variant (PANOPTIC 5 cams, ARPLAB 6 cams, RING23 23 cams)."""
    if variant not in _engines:
        e = env(variant)
        eng = pkg('pipeline').Engine(e.params, e.calib, max_frames=4,
                                     max_persons_per_camera=10 if variant in ('panoptic', 'arplab') else 3)
        eng.load_gat(*e.gat)
        eng.load_mlp(e.mlp)
        _engines[variant] = eng
    return _engines[variant]


@pytest.fixture(scope='module', autouse=True)
def _close_engines():
    yield
    for eng in _engines.values():
        eng.close()
    _engines.clear()


def _pi(frame):
    return oracle().processed_input(frame)


def test_library_loaded():
    L = pkg('lib')
    assert os.path.exists(L.LIB_PATH)
    assert L.load().mpe_version().startswith(b'mpe-hip')


@pytest.mark.parametrize('m,k,n,slope', [(1, 32, 1, None), (5, 150, 150, 0.15), (180, 902, 400, None),
                                         (333, 400, 320, 0.15), (4, 1260, 3072, 0.1), (257, 1024, 54, None)])
def test_linear_vs_torch_fp32(engine, m, k, n, slope):
    g = torch.Generator().manual_seed(m * 1000 + n)
    x = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g) / np.sqrt(k)
    b = torch.randn(n, generator=g)
    ref = torch.nn.functional.linear(x, w, b)
    ref64 = torch.nn.functional.linear(x.double(), w.double(), b.double())
    if slope is not None:
        ref = torch.nn.functional.leaky_relu(ref, slope)
        ref64 = torch.nn.functional.leaky_relu(ref64, slope)
    y = engine.linear(x.cuda(), w.numpy(), b.numpy(), slope).cpu()
    # one fp32 MFMA chain of length k: error grows like sqrt(k)*eps
    tol = 4e-7 * np.sqrt(k) * 4
    assert (y - ref).abs().max().item() < tol
    assert (y.double() - ref64).abs().max().item() < tol
    # f64 running sums: about one rounding, i.e. at least as close to exact as torch's sgemm
    y2 = engine.linear(x.cuda(), w.numpy(), b.numpy(), slope, acc64=True).cpu()
    e_gpu = (y2.double() - ref64).abs().max().item()
    e_cpu = (ref.double() - ref64).abs().max().item()
    scale = ref64.abs().max().item()
    assert e_gpu <= max(2.0 * e_cpu, 2.5e-7 * scale), (e_gpu, e_cpu)


@pytest.mark.parametrize('k,n,slope', [(902, 400, None), (400, 320, 0.15), (1260, 3072, 0.1), (1024, 54, None)])
def test_linear_small_batch_rows_identical(engine, k, n, slope):
    """The latency kernel (one wave per 16x16 tile, chosen for small batches) keeps the k order,
    fp32 chain and f64 flush cadence of the throughput kernel: the same row gives the same bits
    whether it travels in a batch of 37 or of 3000."""
    g = torch.Generator().manual_seed(k + n)
    x = torch.randn(3000, k, generator=g)
    w = (torch.randn(n, k, generator=g) / np.sqrt(k)).numpy()
    b = torch.randn(n, generator=g).numpy()
    for acc64 in (False, True):
        big = engine.linear(x.cuda(), w, b, slope, acc64=acc64).cpu()
        for m in (1, 16, 37):
            small = engine.linear(x[:m].cuda(), w, b, slope, acc64=acc64).cpu()
            assert torch.equal(small, big[:m]), (acc64, m, (small - big[:m]).abs().max().item())


@pytest.mark.parametrize('variant,name', ALL_CASES_FZ)
def test_head_features_vs_golden(variant, name):
    engine = engine_for(variant)
    arr, frames = load_case(name, variant)
    for n, frame in enumerate(frames):
        p = 'f%d_' % n
        pb = engine.pack([_pi(frame)])
        feat = engine.head_features(engine.to_device(pb)).cpu().numpy()      # [H][J][10]
        H = pb.n_heads
        dense = np.zeros((H, env(variant).meta['num_feats']), np.float32)
        rc = arr[p + 'feat_rc']
        sel = rc[:, 0] < H
        dense[rc[sel, 0], rc[sel, 1]] = arr[p + 'feat_v'][sel]
        for h in range(H):
            c = pb.head_cam[h]
            want = dense[h, 2 + c * 180: 2 + (c + 1) * 180].reshape(18, 10)
            # rays are 3-term f32 dot products: allow 2 ulp of the largest term
            np.testing.assert_allclose(feat[h], want, rtol=0, atol=5e-7)
            assert dense[h, 0] == 1.0


@pytest.mark.parametrize('variant,name', ALL_CASES_FZ)
def test_dense_rows_vs_golden(variant, name):
    """mpe_dense_rows = graph.ndata['h'] of one frame as the reference builds it (graph_generator.py:444-508, 629-631): EVERY entry of
    the dense N x F matrix against the reference's (stored sparse in the fixtures): head rows (column 0, the camera's block),
    edge-node rows (one-hot at column 1), zeros everywhere else."""
    engine = engine_for(variant)
    arr, frames = load_case(name, variant)
    F = env(variant).meta['num_feats']
    for n, frame in enumerate(frames):
        p = 'f%d_' % n
        if (p + 'N') not in arr or int(arr[p + 'N']) == 0:
            continue
        db = engine.to_device(engine.pack([_pi(frame)]))
        if db.n_heads + db.n_edge_nodes == 0:
            continue
        got = engine.dense_rows(db).cpu().numpy()
        N = int(arr[p + 'N'])
        assert got.shape == (N, F)
        want = np.zeros((N, F), np.float32)
        rc = arr[p + 'feat_rc']
        want[rc[:, 0], rc[:, 1]] = arr[p + 'feat_v']
        np.testing.assert_allclose(got, want, rtol=0, atol=5e-7)
        assert np.array_equal(got == 0, want == 0) or np.abs(got[(got == 0) != (want == 0)]).max() < 5e-7


@pytest.mark.parametrize('variant,name', ALL_CASES_FZ)
def test_gat_scores_vs_golden(variant, name):
    engine = engine_for(variant)
    arr, frames = load_case(name, variant)
    for n, frame in enumerate(frames):
        p = 'f%d_' % n
        db = engine.to_device(engine.pack([_pi(frame)]))
        sc, sh = engine.gat_scores(db, heads=True)
        want = arr[p + 'scores']
        H = db.n_heads
        np.testing.assert_allclose(sc.cpu().numpy(), want[H:], rtol=0, atol=2e-5)
        np.testing.assert_allclose(sh.cpu().numpy(), want[:H], rtol=0, atol=2e-5)


@pytest.mark.parametrize('variant', ['wave', 'lds', 'big', 'block', 'ring23', 'ring23_wave', 'ring23_big'])
def test_cluster_known_answers_bit_exact(engine, calib, variant, monkeypatch):
    """400 (+120 with 23 cameras) known answers of the reference's
    get_person_proposal_from_network_output, through every clustering kernel: k_cluster_wave
    (<= 64 heads per frame: state in wave registers), k_cluster_block (larger frames: one
    workgroup per frame), and the two sequential variants kept as cross-checks, k_cluster_lds
    (work arrays in LDS) and k_cluster_big (global scratch + heapsort)."""
    arr = np.load(os.path.join(GOLDEN, 'ring23' if variant.startswith('ring23') else '', 'cluster_cases.npz'))
    packing = pkg('packing')
    kernel = {'wave': 'wave', 'lds': 'lds', 'big': 'big', 'block': 'block', 'ring23': None,
              'ring23_wave': 'wave', 'ring23_big': 'big'}[variant]
    if kernel:                      # the default choice follows the engine's capacity: wave <= 64 heads < block
        monkeypatch.setenv('MPE_CLUSTER_KERNEL', kernel)
    if variant.startswith('ring23'):  # 23 cameras: larger components, CPython set growth 8 -> 32 -> 128
        engine = engine_for('ring23')
    V = engine.V
    # assemble all cases into batches of <= 64 frames
    cases = range(int(arr['n']))
    step = min(64, engine.max_frames)
    for start in range(0, len(cases), step):
        chunk = list(cases)[start:start + step]
        pb = packing.PackedBatch(V, engine.J)
        B = len(chunk)
        pb.n_frames = B
        pb.slot_cam = np.full((B, V), -1, np.int32)
        pb.slot_n = np.zeros((B, V), np.int32)
        head_off = [0]
        en_off = [0]
        head_cam = []
        scores = []
        for f, i in enumerate(chunk):
            sc_, sn_ = arr['c%d_slot_cam' % i], arr['c%d_slot_n' % i]
            pb.slot_cam[f, :len(sc_)] = sc_
            pb.slot_n[f, :len(sn_)] = sn_
            tot = int(sn_.sum())
            head_off.append(head_off[-1] + tot)
            en_off.append(en_off[-1] + (tot * tot - int((sn_ * sn_).sum())) // 2)
            head_cam += [int(c) for c, k in zip(sc_, sn_) for _ in range(k)]
            scores.append(arr['c%d_scores' % i])
        n = head_off[-1]
        pb.frame_head_off = np.array(head_off, np.int32)
        pb.frame_en_off = np.array(en_off, np.int32)
        pb.head_cam = np.array(head_cam, np.int32)
        pb.joint_mask = np.ones(n, np.uint32)
        pb.tri_mask = np.ones(n, np.uint32)
        pb.xy = np.zeros((n, engine.J, 2))
        pb.vp = np.zeros((n, engine.J, 2), np.float32)
        db = engine.to_device(pb)
        persons, n_persons = engine.cluster(db, torch.from_numpy(np.concatenate(scores)))
        persons, n_persons = persons.cpu().numpy(), n_persons.cpu().numpy()
        for f, i in enumerate(chunk):
            want = arr['c%d_persons' % i]
            assert n_persons[f] == len(want), i
            assert np.array_equal(persons[f, :len(want)], want), i


@pytest.mark.parametrize('kernel', [None, 'wave', 'block', 'lds', 'big'])
def test_cluster_fresh_known_answers_bit_exact(calib, kernel, monkeypatch):
    """The 1500 further known answers of the reference's clustering function (tests/golden/cluster_cases_fuzz.npz, written by
    oracle/gen_cluster_fuzz.py from the reference's own file: near-threshold scores in float32 steps, saturated ties, confidently wrong
    links, up to ten skeletons per camera) through the default kernel choice and through each of the four clustering kernels: the
    reference's persons, bit for bit (a8)."""
    arr = np.load(os.path.join(GOLDEN, 'cluster_cases_fuzz.npz'))
    packing = pkg('packing')
    engine = pkg('pipeline').Engine(calib.params, calib, max_frames=64, max_persons_per_camera=10)
    if kernel:
        monkeypatch.setenv('MPE_CLUSTER_KERNEL', kernel)
    try:
        V, n_cases = engine.V, int(arr['n'])
        for start in range(0, n_cases, 64):
            chunk = list(range(start, min(start + 64, n_cases)))
            pb = packing.PackedBatch(V, engine.J)
            B = len(chunk)
            pb.n_frames = B
            pb.slot_cam = np.full((B, V), -1, np.int32)
            pb.slot_n = np.zeros((B, V), np.int32)
            head_off, en_off, head_cam, scores = [0], [0], [], []
            for f, i in enumerate(chunk):
                sc_, sn_ = arr['c%d_slot_cam' % i], arr['c%d_slot_n' % i]
                pb.slot_cam[f, :len(sc_)] = sc_
                pb.slot_n[f, :len(sn_)] = sn_
                tot = int(sn_.sum())
                head_off.append(head_off[-1] + tot)
                en_off.append(en_off[-1] + (tot * tot - int((sn_ * sn_).sum())) // 2)
                head_cam += [int(c) for c, k in zip(sc_, sn_) for _ in range(k)]
                scores.append(arr['c%d_scores' % i])
            n = head_off[-1]
            pb.frame_head_off = np.array(head_off, np.int32)
            pb.frame_en_off = np.array(en_off, np.int32)
            pb.head_cam = np.array(head_cam, np.int32)
            pb.joint_mask = np.ones(n, np.uint32)
            pb.tri_mask = np.ones(n, np.uint32)
            pb.xy = np.zeros((n, engine.J, 2))
            pb.vp = np.zeros((n, engine.J, 2), np.float32)
            db = engine.to_device(pb)
            persons, n_persons = engine.cluster(db, torch.from_numpy(np.concatenate(scores)))
            persons, n_persons = persons.cpu().numpy(), n_persons.cpu().numpy()
            for f, i in enumerate(chunk):
                want = arr['c%d_persons' % i]
                assert n_persons[f] == len(want), (kernel, i)
                assert np.array_equal(persons[f, :len(want)], want), (kernel, i)
    finally:
        engine.close()


def test_cluster_large_frames_vs_oracle(monkeypatch):
    """23 cameras x 10 skeletons = 230 heads, 25 300 edge-nodes per frame: more matchings above
    the threshold than the LDS sort holds (global-scratch sort of k_cluster_block), long chains of
    merges, components with repeated cameras.  Oracle = the restated reference rules."""
    onp = oracle()
    packing = pkg('packing')
    e = env('ring23')
    engine = pkg('pipeline').Engine(e.params, e.calib, max_frames=3, max_persons_per_camera=10)
    V, P = engine.V, 10
    rng = np.random.default_rng(5)
    slot_n = np.full(V, P, np.int32)
    pairs = packing.pairs_of_frame(slot_n)
    H, M = V * P, len(pairs)
    head_cam = np.repeat(np.arange(V, dtype=np.int32), P)
    person_of = np.tile(rng.permutation(P), V)                      # hidden identity of every head
    same = person_of[pairs[:, 0]] == person_of[pairs[:, 1]]
    cases = [rng.random(M).astype(np.float32),                                          # half above 0.5
             np.where(same, 0.97, 0.03).astype(np.float32) + rng.normal(0, 0.02, M).astype(np.float32),
             np.where(same ^ (rng.random(M) < 0.15), 0.9, 0.1).astype(np.float32) + rng.normal(0, 0.05, M).astype(np.float32)]
    B = len(cases)
    pb = packing.PackedBatch(V, engine.J)
    pb.n_frames = B
    pb.slot_cam = np.tile(np.arange(V, dtype=np.int32), (B, 1))
    pb.slot_n = np.tile(slot_n, (B, 1))
    pb.frame_head_off = np.arange(B + 1, dtype=np.int32) * H
    pb.frame_en_off = np.arange(B + 1, dtype=np.int32) * M
    pb.head_cam = np.tile(head_cam, B)
    pb.joint_mask = np.ones(B * H, np.uint32)
    pb.tri_mask = np.ones(B * H, np.uint32)
    pb.xy = np.zeros((B * H, engine.J, 2))
    pb.vp = np.zeros((B * H, engine.J, 2), np.float32)
    db = engine.to_device(pb)
    want = [np.array(onp.cluster(sc, pairs, H, list(head_cam), V), np.int32).reshape(-1, V) for sc in cases]
    assert (cases[0] > 0.5).sum() > 8192
    for kernel in ('block', 'big'):
        monkeypatch.setenv('MPE_CLUSTER_KERNEL', kernel)
        persons, n_persons = engine.cluster(db, torch.from_numpy(np.concatenate(cases)))
        persons, n_persons = persons.cpu().numpy(), n_persons.cpu().numpy()
        for f in range(B):
            keep = min(len(want[f]), engine.pcap)
            assert n_persons[f] == keep, (kernel, f)
            assert np.array_equal(persons[f, :keep], want[f][:keep]), (kernel, f)
    engine.close()


@pytest.mark.parametrize('variant,name', ALL_CASES_FZ)
def test_match_and_3d_vs_golden(variant, name):
    """End to end per golden frame: clusters bit-exact, MLP rows / poses / triangulation
    within tolerance (see test_mlp_error_budget for the 3D bound)."""
    engine = engine_for(variant)
    arr, frames = load_case(name, variant)
    for n, frame in enumerate(frames):
        p = 'f%d_' % n
        db = engine.to_device(engine.pack([_pi(frame)]))
        scores, persons, n_persons = engine.match(db)
        want = arr[p + 'persons']
        H = db.n_heads
        gs = np.sort(arr[p + 'scores'][H:])
        assert int(n_persons[0]) == len(want)
        assert np.array_equal(persons[0, :len(want)].cpu().numpy(), want)
        if len(want) == 0:
            continue
        rows, valid = engine.mlp_input_rows(db, persons, n_persons)
        np.testing.assert_allclose(rows[0, :len(want)].cpu().numpy(), arr[p + 'mlp_in'], rtol=0, atol=3e-7)
        assert valid[0, :len(want)].all()
        # MLP on IDENTICAL rows (the reference's): |gpu - ref| is bounded by the two sides' distances
        # to the exactly evaluated network, and the HIP side is the closer one.  No additive slack.
        onp = oracle()
        mlp_sd = env(variant).mlp
        x_ref = torch.from_numpy(arr[p + 'mlp_in'])
        y = engine.mlp_forward(x_ref.cuda()).cpu().numpy()
        exact = onp.mlp_exact(mlp_sd, x_ref).numpy()
        e_cpu = np.abs(arr[p + 'mlp_out'] - exact).max()
        e_gpu = np.abs(y - exact).max()
        assert e_gpu <= e_cpu, (e_gpu, e_cpu)
        assert np.abs(y - arr[p + 'mlp_out']).max() <= e_cpu + e_gpu
        # end to end the HIP path feeds its OWN rows (<= 3e-7 from the reference's, above).  Same
        # rule on those rows with torch-CPU (= the reference's MLP arithmetic) as the other side;
        # the batched path must give the bits of the stage call and the x10 decode must be exact fp32.
        x_gpu = rows[0, :len(want)]
        y_gpu_own = engine.mlp_forward(x_gpu.contiguous()).cpu().numpy()
        y_cpu_own = onp.mlp_forward(mlp_sd, x_gpu.cpu()).numpy()
        ex_own = onp.mlp_exact(mlp_sd, x_gpu.cpu()).numpy()
        e_gpu_own, e_cpu_own = np.abs(y_gpu_own - ex_own).max(), np.abs(y_cpu_own - ex_own).max()
        assert e_gpu_own <= e_cpu_own, (e_gpu_own, e_cpu_own)
        poses, pv = engine.mlp3d(db, persons, n_persons)
        got_pose = poses[0, :len(want)].cpu().numpy()
        assert np.array_equal(got_pose.reshape(len(want), -1), y_gpu_own * np.float32(10.0))
        # distance to the reference's poses = MLP budget + what the reference network itself makes of
        # the row difference (torch-CPU on both sets of rows) + one fp32 quantum of the decode
        drift = 10 * np.abs(y_cpu_own - arr[p + 'mlp_out']).max()
        q = float(np.spacing(np.float32(np.abs(arr[p + 'poses']).max())))
        d = np.abs(got_pose - arr[p + 'poses']).max()
        assert d <= 10 * (e_gpu_own + e_cpu_own) + drift + q, (d, e_gpu_own, e_cpu_own, drift)
        tri, jv = engine.triangulate(db, persons, n_persons)
        tv = arr[p + 'tri_valid'].astype(bool)
        has_id = any('ID' in sk for cam in frame for sk in json.loads(frame[cam][0]))
        if not has_id:
            assert np.array_equal(jv[0, :len(want)].cpu().numpy().astype(bool), tv)
            got = tri[0, :len(want)].cpu().numpy()
            np.testing.assert_allclose(got[tv], arr[p + 'tri'][tv], rtol=1e-9, atol=1e-9)


def _first_divergence(s_gpu, s_ref, thr=0.5):
    """Both score vectors drive the same integer logic (stable sort by descending score of the
    matchings above the threshold, then sequential rules), so two runs can only differ from the
    first position where the sorted sequences differ.  Returns (gap, allowed): the oracle-score gap
    that decided that position and the largest gap the measured score deviation can explain."""
    dev = float(np.abs(s_gpu - s_ref).max())
    og = [m for m in np.argsort(-s_gpu, kind='stable') if s_gpu[m] > thr]
    orf = [m for m in np.argsort(-s_ref, kind='stable') if s_ref[m] > thr]
    for k in range(max(len(og), len(orf))):
        a = og[k] if k < len(og) else None
        b = orf[k] if k < len(orf) else None
        if a == b:
            continue
        if a is None or b is None:                 # one list ended: a score crossed the threshold
            m = b if a is None else a
            return abs(float(s_ref[m]) - thr), dev
        return abs(float(s_ref[a]) - float(s_ref[b])), 2.0 * dev
    return None, dev


def _exact_mlp(x, weights):
    """The network evaluated in f64 with activations rounded to fp32 between layers (what both
    fp32 implementations approximate)."""
    keys = sorted({int(k.split('.')[1]) for k in weights})
    h = x.double()
    for n, k in enumerate(keys):
        W = torch.from_numpy(weights['layers.%d.weight' % k]).double()
        b = torch.from_numpy(weights['layers.%d.bias' % k]).double()
        h = h @ W.T + b
        if n != len(keys) - 1:
            h = torch.nn.functional.leaky_relu(h, 0.1)
        h = h.float().double()
    return h


def test_mlp_error_budget(engine, mlp_weights):
    """The north star asks for 3D joints within 1e-3 mm of the reference in fp32.  The
    reference's own fp32 MLP (torch-CPU sgemm) is ~4e-3 mm away from the exactly evaluated
    network, so that figure cannot be a bound on |gpu - reference|.  What we require instead:
    the HIP path (fp32 MFMA + f64 running sums) is at least as close to the exact per-layer
    result as torch-CPU is, and the two differ by no more than their combined noise."""
    arr, _ = load_case('c4_5x10')
    x = torch.from_numpy(arr['f0_mlp_in'])
    h = _exact_mlp(x, mlp_weights)
    cpu = torch.from_numpy(arr['f0_mlp_out']).double()
    gpu = engine.mlp_forward(x.cuda()).cpu().double()
    e_cpu = (cpu - h).abs().max().item()
    e_gpu = (gpu - h).abs().max().item()
    assert e_gpu <= e_cpu + 5e-8, (e_gpu, e_cpu)
    assert e_gpu * 10 < 3e-6            # < 3e-3 mm from the exact fp32-per-layer network
    assert (gpu - cpu).abs().max().item() <= e_cpu + e_gpu + 1e-8


def test_batch_vs_oracle(engine, calib, gat_weights, mlp_weights):
    """A 48-frame mixed batch: every stage against the CPU oracle on the same inputs."""
    onp = oracle()
    syn = pkg('synthetic')
    sd, prm = gat_weights
    specs = [syn.FrameSpec(persons=4), syn.FrameSpec(persons=3, joint_drop=0.2, noise_px=1.5),
             syn.FrameSpec(persons=5, cameras=['trackerb', 'trackera', 'trackerd', 'trackere']),
             syn.FrameSpec(persons=2, empty_cameras=('trackera',), spurious=1)]
    frames = []
    for i in range(48):
        f, _ = syn.make_frame(calib, 100 + i, specs[i % len(specs)])
        frames.append(onp.processed_input(f))
    db = engine.to_device(engine.pack(frames))
    scores, persons, n_persons = engine.match(db)
    poses, valid = engine.mlp3d(db, persons, n_persons)
    tri, jv = engine.triangulate(db, persons, n_persons)
    scores = scores.cpu().numpy()
    persons, n_persons = persons.cpu().numpy(), n_persons.cpu().numpy()
    poses, tri, jv = poses.cpu().numpy(), tri.cpu().numpy(), jv.cpu().numpy()
    sm = list(calib.params.used_cameras_skeleton_matching)
    exact, flagged = 0, 0
    worst_pose = 0.0
    gaps = []
    for f, frame in enumerate(frames):
        h0, H, e0, M = db.host.frame_counts(f)
        res = onp.run_frame(frame, calib, sd, prm, mlp_weights, mode='mlp')
        np.testing.assert_allclose(scores[e0:e0 + M], res['scores'], rtol=0, atol=2e-5)
        # clustering on the GPU's own scores must match the oracle exactly (integer logic)
        head_cam = [sm.index(c) for c in res['graph']['nodes_camera'][:H]]
        own = onp.cluster(scores[e0:e0 + M], res['graph']['pairs'], H, head_cam, len(sm))
        assert n_persons[f] == len(own)
        assert np.array_equal(persons[f, :len(own)], np.array(own, np.int32).reshape(-1, len(sm)))
        # against the oracle's scores the result can only differ when two scores that
        # decide an order are closer than the fp32 reordering noise
        if own == res['persons']:
            exact += 1
            if len(own):
                # mixed-up / spurious skeletons triangulate far away and drive MLP inputs and
                # outputs to tens of metres: the bound scales with the magnitude (4 m = volume)
                mag = max(1.0, float(np.abs(res['poses']).max()) / 4.0)
                worst_pose = max(worst_pose, np.abs(poses[f, :len(own)] - res['poses']).max() / mag)
                for k, person in enumerate(own):
                    sk = onp.person_skeletons(person, res['graph']['jsons_for_head'], sm)
                    t = onp.triangulate_person(sk, calib)
                    for j in range(18):
                        assert bool(jv[f, k, j]) == (j in t)
                        if j in t:
                            np.testing.assert_allclose(tri[f, k, j], t[j], rtol=1e-9, atol=1e-9)
        else:
            # a differing frame is excused only by the score gap at the FIRST diverging decision:
            # the two matchings that swap places there must be closer, in the oracle's scores, than
            # twice the measured score deviation of this frame (or one of them must sit within that
            # deviation of the threshold)
            flagged += 1
            gap, allowed = _first_divergence(scores[e0:e0 + M], res['scores'])
            assert gap is not None and gap <= allowed, (f, gap, allowed)
            gaps.append((f, gap, allowed))
    print('clusters equal to the oracle in %d of %d frames; excused by first-divergence gaps %s' % (exact, len(frames), gaps))
    assert exact >= 44, (exact, flagged)
    # end to end (HIP rows, <= 3e-7 from the oracle's rows, through a 9-layer MLP), relative to the pose magnitude
    assert worst_pose < 1.2e-5


def test_mpjpe_equal_to_reference_path(engine, calib, mlp_weights):
    """North star: "MPJPE equal to reference within 0.01 mm".  Same clusters (ground-truth
    pairing through the clustering kernel), both 3D stages, MPJPE against the synthetic ground
    truth computed for the HIP path and for the CPU oracle."""
    onp = oracle()
    syn = pkg('synthetic')
    common = pkg('harness.common')
    spec = syn.FrameSpec(persons=4, noise_px=1.0)
    frames, owners, gts = [], [], []
    for i in range(24):
        f, gt = syn.make_frame(calib, 500 + i, spec)
        frames.append(onp.processed_input(f))
        owners.append(gt['owner'])
        gts.append(gt['persons'])
    db = engine.to_device(engine.pack(frames, keep_json=True))
    persons, n_persons = engine.cluster(db, common.teacher_scores(db, owners))
    poses, valid = engine.mlp3d(db, persons, n_persons)
    tri, jv = engine.triangulate(db, persons, n_persons)
    persons, n_persons = persons.cpu().numpy(), n_persons.cpu().numpy()
    poses, tri, jv = poses.cpu().numpy(), tri.cpu().numpy(), jv.cpu().numpy()
    sm = list(calib.params.used_cameras_skeleton_matching)
    used = calib.params.used_joints

    def mpjpe(pred, gt_people):
        return min(float(np.mean([np.linalg.norm(pred[j] - g[j]) for j in used])) for g in gt_people)
    e = {'mlp_gpu': [], 'mlp_cpu': [], 'tri_gpu': [], 'tri_cpu': []}
    for f in range(len(frames)):
        assert n_persons[f] == 4
        for k in range(4):
            sk = onp.person_skeletons(list(persons[f, k]), db.host.jsons_for_head[f], sm)
            row, kept = onp.mlp_input_row(sk, calib)
            cpu_pose = onp.decode_pose(onp.mlp_forward(mlp_weights, row[None])[0], 18)
            cpu_tri = onp.triangulate_person(sk, calib)
            e['mlp_gpu'].append(mpjpe(poses[f, k], gts[f]))
            e['mlp_cpu'].append(mpjpe(cpu_pose, gts[f]))
            e['tri_gpu'].append(mpjpe(tri[f, k], gts[f]))
            e['tri_cpu'].append(mpjpe(np.stack([cpu_tri.get(j, np.zeros(3)) for j in range(18)]), gts[f]))
    m = {k: float(np.mean(v)) for k, v in e.items()}
    assert abs(m['mlp_gpu'] - m['mlp_cpu']) < 1e-5          # 0.01 mm
    assert abs(m['tri_gpu'] - m['tri_cpu']) < 1e-8
    assert m['tri_gpu'] < 0.01                               # 1 px of noise -> millimetres


def test_ragged_and_empty_frames(engine, calib, gat_weights, mlp_weights):
    """Edge cases of the reference's skip rules inside one batch: an empty frame, a frame with a
    single camera (heads but no cross-camera pair -> no graph, metrics_from_model.py:195-196), a
    camera whose skeletons have no joints, and normal frames around them."""
    onp = oracle()
    syn = pkg('synthetic')
    sd, prm = gat_weights
    normal = [onp.processed_input(syn.make_frame(calib, 900 + i)[0]) for i in range(3)]
    one_cam = {'trackerb': normal[0]['trackerb']}
    hollow = dict(normal[1])
    hollow['trackera'] = ['[{"ID": 1}, {}]', 0]
    frames = [normal[0], {}, one_cam, hollow, normal[2]]
    db = engine.to_device(engine.pack(frames))
    scores, persons, n_persons = engine.match(db)
    poses, valid = engine.mlp3d(db, persons, n_persons)
    tri, jv = engine.triangulate(db, persons, n_persons)
    torch.cuda.synchronize()
    n_persons = n_persons.cpu().numpy()
    assert n_persons[1] == 0 and n_persons[2] == 0
    scores = scores.cpu().numpy()
    sm = list(calib.params.used_cameras_skeleton_matching)
    for f in (0, 3, 4):
        h0, H, e0, M = db.host.frame_counts(f)
        res = onp.run_frame(frames[f], calib, sd, prm, mlp_weights, mode='mlp')
        np.testing.assert_allclose(scores[e0:e0 + M], res['scores'], rtol=0, atol=2e-5)
        head_cam = [sm.index(c) for c in res['graph']['nodes_camera'][:H]]
        own = onp.cluster(scores[e0:e0 + M], res['graph']['pairs'], H, head_cam, len(sm))
        assert n_persons[f] == len(own)
        assert np.array_equal(persons[f, :len(own)].cpu().numpy(), np.array(own, np.int32).reshape(-1, len(sm)))
    assert not valid[1].any() and not valid[2].any()


def test_capacity_is_enforced(calib):
    syn = pkg('synthetic')
    onp = oracle()
    eng = pkg('pipeline').Engine(calib.params, calib, max_frames=2, max_persons_per_camera=2)
    try:
        frame = onp.processed_input(syn.make_frame(calib, 1)[0])       # 4 skeletons per camera
        with pytest.raises(ValueError):
            eng.to_device(eng.pack([frame]))
        with pytest.raises(ValueError):
            eng.to_device(eng.pack([{}, {}, {}]))
    finally:
        eng.close()


def test_full_size_batch_properties(calib, gat_weights, mlp_weights):
    """BASELINE configs[1] at full size (1000 frames, 5 views x 4 persons).  Size-independent
    properties: (1) frames are independent units, so any split / reordering of the batch gives
    bit-identical per-frame results (scores, clusters, poses); (2) structural invariants of the
    clustering output; (3) a 25-frame sample agrees with the CPU oracle."""
    onp = oracle()
    syn = pkg('synthetic')
    sd, prm = gat_weights
    eng = pkg('pipeline').Engine(calib.params, calib, max_frames=1000, max_persons_per_camera=4)
    try:
        eng.load_gat(sd, prm)
        eng.load_mlp(mlp_weights)
        specs = [syn.FrameSpec(persons=4), syn.FrameSpec(persons=4, noise_px=2.0, joint_drop=0.1),
                 syn.FrameSpec(persons=3), syn.FrameSpec(persons=4, cameras=['trackere', 'trackerb', 'trackera', 'trackerd'])]
        uniq = [onp.processed_input(syn.make_frame(calib, 2000 + i, specs[i % 4])[0]) for i in range(100)]
        frames = [uniq[(7 * i) % 100] for i in range(1000)]

        def run(fr):
            db = eng.to_device(eng.pack(fr))
            sc, p, n = eng.match(db)
            poses, valid = eng.mlp3d(db, p, n)
            tri, jv = eng.triangulate(db, p, n)
            torch.cuda.synchronize()
            return db, sc.cpu().numpy(), p.cpu().numpy(), n.cpu().numpy(), poses.cpu().numpy(), tri.cpu().numpy()
        db, sc, p, n, poses, tri = run(frames)
        assert db.n_frames == 1000
        # (1a) ten chunks of 100
        e_off = db.host.frame_en_off
        for c in range(0, 1000, 100):
            _, sc2, p2, n2, poses2, tri2 = run(frames[c:c + 100])
            assert np.array_equal(sc2, sc[e_off[c]:e_off[c + 100]])
            assert np.array_equal(p2, p[c:c + 100]) and np.array_equal(n2, n[c:c + 100])
            assert np.array_equal(poses2, poses[c:c + 100]) and np.array_equal(tri2, tri[c:c + 100])
        # (1b) reversed order
        _, scr, pr, nr, posesr, trir = run(frames[::-1])
        assert np.array_equal(pr[::-1], p) and np.array_equal(nr[::-1], n)
        assert np.array_equal(posesr[::-1], poses) and np.array_equal(trir[::-1], tri)
        # (2) invariants: a head belongs to at most one person, every person spans >= 2 cameras,
        #     the head sits in the camera column it was detected by
        for f in range(1000):
            h0, H, e0, M = db.host.frame_counts(f)
            seen = set()
            for k in range(n[f]):
                members = [(c, h) for c, h in enumerate(p[f, k]) if h >= 0]
                assert len(members) >= calib.params.min_number_of_views
                for c, h in members:
                    assert h < H and db.host.head_cam[h0 + h] == c and h not in seen
                    seen.add(h)
            assert (p[f, n[f]:] == -1).all()
        # (3) oracle on a sample
        sm = list(calib.params.used_cameras_skeleton_matching)
        for f in range(0, 1000, 40):
            h0, H, e0, M = db.host.frame_counts(f)
            res = onp.run_frame(frames[f], calib, sd, prm, mlp_weights, mode='mlp')
            np.testing.assert_allclose(sc[e0:e0 + M], res['scores'], rtol=0, atol=2e-5)
            head_cam = [sm.index(c) for c in res['graph']['nodes_camera'][:H]]
            own = onp.cluster(sc[e0:e0 + M], res['graph']['pairs'], H, head_cam, len(sm))
            assert n[f] == len(own) and np.array_equal(p[f, :len(own)], np.array(own, np.int32).reshape(-1, len(sm)))
    finally:
        eng.close()


def test_bf16_mlp_mode_is_close_but_not_parity(calib, mlp_weights):
    """Reduced-precision variant of BASELINE configs[4] (bf16 MFMA for the MLP GEMMs): ~3
    significant digits per layer.  Checked against the fp32 result with a bf16-sized bound;
    the parity path never uses it."""
    arr, _ = load_case('c4_5x10')
    x = torch.from_numpy(arr['f0_mlp_in'])
    eng = pkg('pipeline').Engine(calib.params, calib, max_frames=4, max_persons_per_camera=10)
    try:
        eng.load_mlp(mlp_weights)
        y32 = eng.mlp_forward(x.cuda()).cpu().numpy()
        eng.set_precision(False, False, mlp_bf16=True)
        y16 = eng.mlp_forward(x.cuda()).cpu().numpy()
        eng.set_precision(False, True)
        y32b = eng.mlp_forward(x.cuda()).cpu().numpy()
    finally:
        eng.close()
    np.testing.assert_allclose(y32, arr['f0_mlp_out'], rtol=0, atol=1.5e-6)
    assert np.array_equal(y32, y32b)                      # switching modes back is clean
    scale = np.abs(arr['f0_mlp_out']).max()
    d = np.abs(y16 - arr['f0_mlp_out']).max()
    assert 1e-5 < d < 0.05 * scale, (d, scale)            # visibly reduced precision, but sane


@pytest.mark.parametrize('variant,name', [('panoptic', 'c2_5x4_clean'), ('panoptic', 'c4_5x10'), ('ring23', None)])
def test_reduced_gat_mode_is_close_but_not_parity(variant, name):
    """Reduced-precision GAT of BASELINE configs[4]: fc1/fc2 on the bf16 MFMA, ft2 stored as fp16
    rows for the attention stage.  5x4 runs the fused attention kernel, 5x10 and the 23-camera
    frames the unfused kernels (and, with 23 cameras, the per-camera grouped layer-0 GEMM).
    Scores stay within a few 1e-2 of the fp32 scores; switching back restores the fp32 bits."""
    engine = engine_for(variant)
    if name is None:
        arr, frames = load_case(sorted(c for v, c in ALL_CASES if v == variant)[0], variant)
    else:
        arr, frames = load_case(name, variant)
    db = engine.to_device(engine.pack([_pi(f) for f in frames]))
    s32, h32 = engine.gat_scores(db, heads=True)
    s32, h32 = s32.cpu().numpy(), h32.cpu().numpy()
    try:
        engine.set_precision(gat_reduced=True)
        s16, h16 = engine.gat_scores(db, heads=True)
        s16, h16 = s16.cpu().numpy(), h16.cpu().numpy()
    finally:
        engine.set_precision()
    s32b = engine.gat_scores(db).cpu().numpy()
    assert np.array_equal(s32, s32b)
    d = max(np.abs(s16 - s32).max(), np.abs(h16 - h32).max())
    assert 1e-6 < d < 0.08, d                                  # visibly reduced precision, but sane
    # well separated decisions are unchanged
    far = np.abs(s32 - 0.5) > 0.1
    assert np.array_equal(s16[far] > 0.5, s32[far] > 0.5)


def test_cluster_kernels_agree_on_random_frames(calib, monkeypatch):
    """All four clustering kernels (two parallel formulations, two sequential replays) must give
    the same persons on frames they were not tuned on: random camera occupancy (empty cameras,
    up to 12 skeletons), score ties on a coarse grid, scores at the threshold, NaN scores, many
    more than 64 / 256 matchings above the threshold (chunk boundaries of the parallel rounds).
    The sequential kernels are pinned to the reference by the known answers."""
    packing = pkg('packing')
    eng = pkg('pipeline').Engine(calib.params, calib, max_frames=256, max_persons_per_camera=13)
    try:
        V = eng.V
        rng = np.random.default_rng(11)
        B = 256
        slot_n = rng.integers(0, 13, size=(B, V)).astype(np.int32)
        slot_n[rng.random((B, V)) < 0.15] = 0
        slot_n[:8] = 12                                             # full frames: 60 heads, 1440 edge-nodes
        slot_n[8:12] = [13, 13, 13, 13, 12]                         # 64 heads: every lane of the wave kernel in use
        pb = packing.PackedBatch(V, eng.J)
        pb.n_frames = B
        pb.slot_cam = np.tile(np.arange(V, dtype=np.int32), (B, 1))
        for f in range(B):
            rng.shuffle(pb.slot_cam[f])
        pb.slot_n = slot_n
        tot = slot_n.sum(1)
        ens = (tot * tot - (slot_n * slot_n).sum(1)) // 2
        pb.frame_head_off = np.concatenate([[0], np.cumsum(tot)]).astype(np.int32)
        pb.frame_en_off = np.concatenate([[0], np.cumsum(ens)]).astype(np.int32)
        pb.head_cam = np.concatenate([np.repeat(pb.slot_cam[f], slot_n[f]) for f in range(B)]).astype(np.int32)
        n = int(tot.sum())
        pb.joint_mask = np.ones(n, np.uint32)
        pb.tri_mask = np.ones(n, np.uint32)
        pb.xy = np.zeros((n, eng.J, 2))
        pb.vp = np.zeros((n, eng.J, 2), np.float32)
        M = int(ens.sum())
        kind = rng.integers(0, 4, size=B)
        scores = np.empty(M, np.float32)
        for f in range(B):
            s = slice(pb.frame_en_off[f], pb.frame_en_off[f + 1])
            m = ens[f]
            if kind[f] == 0:
                scores[s] = rng.random(m)
            elif kind[f] == 1:
                scores[s] = np.round(rng.random(m) * 8) / 8            # ties, values exactly 0.5
            elif kind[f] == 2:
                scores[s] = 0.5 + rng.random(m) * 0.5                  # everything matches
            else:
                x = rng.random(m).astype(np.float32)
                x[rng.random(m) < 0.05] = np.nan
                scores[s] = x
        db = eng.to_device(pb)
        sc = torch.from_numpy(scores)
        res = {}
        for kernel in ('lds', 'wave', 'block', 'big'):
            monkeypatch.setenv('MPE_CLUSTER_KERNEL', kernel)
            persons, n_persons = eng.cluster(db, sc)
            res[kernel] = (persons.cpu().numpy(), n_persons.cpu().numpy())
        assert res['lds'][1].max() > 4 and (tot > 0).sum() > 200
        for kernel in ('wave', 'block', 'big'):
            assert np.array_equal(res[kernel][1], res['lds'][1]), kernel
            for f in range(B):
                k = res['lds'][1][f]
                assert np.array_equal(res[kernel][0][f, :k], res['lds'][0][f, :k]), (kernel, f)
    finally:
        eng.close()


@pytest.mark.parametrize('preset,n', [('PANOPTIC', 150), ('ARPLAB', 80), ('RING23', 12)])
def test_random_frame_shapes_vs_oracle(preset, n):
    """tests/checkers/shape_fuzz.py on frames of random shape (0-6 persons, camera subsets and orders, empty
    cameras, spurious skeletons, dropped joints, ID keys, noise; single-camera and empty frames
    included) on the three rigs: clusters equal to the oracle's (or the deciding score gap explained by
    the measured score deviation), scores within 2e-5 or no noisier than 2.5-3x the reference's own fp32
    scores, poses within 5e-6 of the output magnitude, DLT points within 1e-8 m, and the raw JSON through
    the native packer giving the same bits.  Runs as a child
    process (the tool is a script); it asserts by itself and writes gpurun_out/shape_fuzz.json."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'checkers', 'shape_fuzz.py'), str(n), '5', preset], capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep['clusters_equal'] + rep['explained'] + rep['graphless'] == n
    # measured (profiles/r02_shape_fuzz.json, r03): every frame with a graph equal, at most one explained by the deciding-gap
    # rule per few hundred frames; a regression to "mostly explained" must fail
    with_graph = n - rep['graphless']
    assert rep['explained'] <= max(2, with_graph // 25), rep
    assert rep['native_packer_same_bits']


def test_tile_kernels_and_wave_per_tile_kernels_give_the_same_scores(tmp_path):
    """Every GEMM of the path has a tile kernel (big batches) and wave-per-16x16-tile kernels (small batches, narrow outputs)
    that are meant to give a row the same bits.  tests/checkers/skinny_vs_tile.py writes the scores of every fixture frame
    and the activations of every layer (mpe_gat_layer), in the default arithmetic and in the f64-sum mode, once in a plain
    process (small frames: wave-per-tile kernels) and once with MPE_SKINNY_WAVES=0 (tile kernels at every batch size: the
    split-bf16 tile forms with four and eight MFMA waves, the fp32 tile kernel, the grouped layer-0 launch); every array must
    be identical.  (Round 4: in the f64-sum mode the grouped layer-0 launch had no f64 sums while the per-camera launches of
    a small batch had them -- 16 of 180 arrays differed by up to 9e-6.)"""
    import subprocess
    import sys
    tool = os.path.join(ROOT, 'tests', 'checkers', 'skinny_vs_tile.py')
    a, b = str(tmp_path / 'plain.npz'), str(tmp_path / 'tile.npz')
    for path, extra in ((a, {}), (b, {'MPE_SKINNY_WAVES': '0'})):
        r = subprocess.run([sys.executable, tool, 'dump', path], capture_output=True, text=True, timeout=600, env=dict(os.environ, **extra))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    r = subprocess.run([sys.executable, tool, 'cmp', a, b], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    assert last.startswith('0 of ') and int(last.split()[2]) >= 150, r.stdout[-3000:]


@pytest.mark.parametrize('k,n,slope', [(1260, 3072, 0.1), (3072, 2048, 0.1), (1024, 54, None), (416, 400, 0.15), (96, 16, None), (33, 70, 0.1),
                                       (20, 96, 0.1), (72, 208, 0.1)])      # (one K stage / three: the shortest loops of the tile kernel's two wave groups)
def test_split_bf16_linear_is_fp32_accurate_and_batch_invariant(engine, k, n, slope):
    """csrc/gemm_sb16.hip, the arithmetic of the MLP launches (MLP mode 3, the default): every fp32 operand as the exact sum
    of three bf16 numbers, six products on the bf16 MFMA, f64 sums every second K stage.  (i) Against the exactly evaluated
    layer (float64) its rms error is that of the fp32 MFMA with f64 sums per stage (mode 1) or below, under 0.3 ulp of the
    output scale (tools/sb16_numerics.hip measured 0.25-0.26 for both); (ii) the three kernels behind it (K
    split over eight waves, wave per 16 x 16 tile, 128-row tiles with LDS-DMA staging) give the same bits for the same row
    whatever the batch it travels in: M = 1, 16, 37 against 3000."""
    g = torch.Generator().manual_seed(7 * k + n)
    x = torch.randn(3000, k, generator=g)
    x = torch.where(x > 0, x, 0.1 * x)
    w = (torch.randn(n, k, generator=g) / np.sqrt(k)).numpy()
    b = torch.randn(n, generator=g).numpy()
    ex = x.double() @ torch.from_numpy(w).double().T + torch.from_numpy(b).double()
    if slope is not None:
        ex = torch.where(ex > 0, ex, slope * ex)
    big = engine.linear(x.cuda(), w, b, slope, split=True).cpu()
    ref = engine.linear(x.cuda(), w, b, slope, acc64=True).cpu()
    scale = ex.abs().max().item()
    ulp = 2.0 ** (np.floor(np.log2(scale)) - 23)
    e_split, e_acc64 = (big.double() - ex).abs().max().item(), (ref.double() - ex).abs().max().item()
    rms, rms64 = (((y.double() - ex) ** 2).mean().sqrt().item() / ulp for y in (big, ref))
    # the maximum over 48 000+ outputs is a tail statistic (up to three fp32 roundings of 0.5 ulp can line up in either form):
    # the rms decides, the maximum is bounded absolutely
    assert rms <= max(1.1 * rms64, 0.2), (rms, rms64)
    assert rms < 0.3, rms
    assert e_split <= 2.0 * ulp, (e_split / ulp, e_acc64 / ulp)
    for m in (1, 16, 37):
        small = engine.linear(x[:m].cuda(), w, b, slope, split=True).cpu()
        assert torch.equal(small, big[:m]), (m, (small - big[:m]).abs().max().item())
    # the same for rows that other waves of the tile kernel own: rows 128-255 of a 256-row tile belong to the second MFMA wave of
    # every SIMD, which runs half a K stage behind the first and reads a stage one barrier longer (round 4: a loader that
    # refilled that buffer too early went unnoticed by the checks above, which only ever looked at rows 0-36), and the last,
    # partial tile
    for m0, m in ((131, 37), (250, 12), (2950, 37), (2984, 16)):
        small = engine.linear(x[m0:m0 + m].cuda(), w, b, slope, split=True).cpu()
        assert torch.equal(small, big[m0:m0 + m]), (m0, m, (small - big[m0:m0 + m]).abs().max().item())
    # non-finite operands travel like in any fp32 GEMM (garbage in, garbage out -- but no fault, no hang)
    xb = x[:40].clone()
    xb[3, 5] = float('inf')
    xb[7, 1] = float('nan')
    yb = engine.linear(xb.cuda(), w, b, slope, split=True).cpu()
    assert not torch.isfinite(yb[3]).all() and not torch.isfinite(yb[7]).all() and torch.equal(yb[0], big[0])


@pytest.mark.parametrize('k,n,slope', [(1260, 3072, 0.1), (3072, 2048, 0.1), (1024, 54, None), (96, 16, None), (33, 70, 0.1), (72, 208, 0.1)])
def test_split_bf16_flush_per_stage_is_more_accurate_and_batch_invariant(engine, k, n, slope):
    """The maximum-accuracy cadence of the split form (MLP mode 4: an fp32 chain and an f64 flush per 32-deep K stage): (i) its rms
    error against the exactly evaluated layer is at most the default cadence's (measured 0.10-0.14 against 0.12-0.16 ulp of the
    output scale on these operands; tools/sb16_numerics.hip: 0.13-0.18 against 0.24-0.26 on MLP-shaped ones) and under 0.2 ulp;
    (ii) the three kernels behind it give a row the same bits in a batch of 1, 16, 37 and 3000, on rows of either MFMA wave group
    and of the partial last tile; (iii) selecting it does not change the default cadence's bits."""
    g = torch.Generator().manual_seed(11 * k + n)
    x = torch.randn(3000, k, generator=g)
    x = torch.where(x > 0, x, 0.1 * x)
    w = (torch.randn(n, k, generator=g) / np.sqrt(k)).numpy()
    b = torch.randn(n, generator=g).numpy()
    ex = x.double() @ torch.from_numpy(w).double().T + torch.from_numpy(b).double()
    if slope is not None:
        ex = torch.where(ex > 0, ex, slope * ex)
    dflt = engine.linear(x.cuda(), w, b, slope, split=True).cpu()
    big = engine.linear(x.cuda(), w, b, slope, split=True, split_flush_per_stage=True).cpu()
    assert torch.equal(engine.linear(x.cuda(), w, b, slope, split=True).cpu(), dflt)
    ulp = 2.0 ** (np.floor(np.log2(ex.abs().max().item())) - 23)
    rms, rms_d = (((y.double() - ex) ** 2).mean().sqrt().item() / ulp for y in (big, dflt))
    assert rms <= rms_d * 1.02 and rms < 0.2, (rms, rms_d)
    assert (big.double() - ex).abs().max().item() <= 1.5 * ulp
    for m0, m in ((0, 1), (0, 16), (0, 37), (131, 37), (250, 12), (2950, 37), (2984, 16)):
        small = engine.linear(x[m0:m0 + m].cuda(), w, b, slope, split=True, split_flush_per_stage=True).cpu()
        assert torch.equal(small, big[m0:m0 + m]), (m0, m, (small - big[m0:m0 + m]).abs().max().item())


@pytest.mark.parametrize('k,n,slope', [(1260, 3072, 0.1), (3072, 1024, 0.1), (1024, 54, None), (33, 70, 0.1)])
def test_f64_matrix_pipe_linear_is_the_exact_layer(engine, k, n, slope):
    """csrc/gemm_f64.hip (MLP mode 5): exact fp32 x fp32 products accumulated in f64 over the whole K.  The result is the layer
    evaluated in float64 and rounded to fp32 -- bit for bit in (nearly) every output: the f64 summation order can only matter
    where the exact value sits within 1e-16 of a rounding boundary -- and a row has the same bits in a batch of 1, 37 and 1500."""
    g = torch.Generator().manual_seed(13 * k + n)
    x = torch.randn(1500, k, generator=g)
    x = torch.where(x > 0, x, 0.1 * x)
    w = (torch.randn(n, k, generator=g) / np.sqrt(k)).numpy()
    b = torch.randn(n, generator=g).numpy()
    ex = x.double() @ torch.from_numpy(w).double().T + torch.from_numpy(b).double()
    if slope is not None:
        ex = torch.where(ex > 0, ex, slope * ex)                  # (the decimal slope as a double: what an f64 evaluation multiplies by)
    y = engine.linear(x.cuda(), w, b, slope, f64mm=True).cpu()
    same = (y == ex.float()).float().mean().item()
    assert same >= 0.9999, same
    assert (y.double() - ex).abs().max().item() <= 1.01 * float(np.spacing(np.float32(ex.abs().max().item())))
    for m0, m in ((0, 1), (100, 37), (1490, 10)):
        small = engine.linear(x[m0:m0 + m].cuda(), w, b, slope, f64mm=True).cpu()
        assert torch.equal(small, y[m0:m0 + m]), (m0, m)


def test_mlp_default_mode_is_the_split_form_and_mode_1_is_still_there(engine, mlp_weights):
    """The two parity forms of the MLP on the same rows: the default (split-bf16) and the fp32-MFMA form (mode 1)
    are both closer to the exactly evaluated network than torch-CPU on the golden rows of the panoptic fixtures and within an ulp
    of each other's maximum error, and
    selecting mode 1 and the default again restores the default's bits."""
    onp = oracle()
    xs = []
    for name in CASES:
        arr, frames = load_case(name)
        for n in range(len(frames)):
            if 'f%d_mlp_in' % n in arr:
                xs.append(arr['f%d_mlp_in' % n])
    x = torch.from_numpy(np.concatenate(xs))
    exact = onp.mlp_exact(mlp_weights, x).double()
    y_cpu = onp.mlp_forward(mlp_weights, x).double()
    y_def = engine.mlp_forward(x.cuda()).cpu()
    engine.set_precision(False, True, mlp_split=False)
    try:
        y_m1 = engine.mlp_forward(x.cuda()).cpu()
    finally:
        engine.set_precision(False, True)
    assert torch.equal(engine.mlp_forward(x.cuda()).cpu(), y_def)
    assert not torch.equal(y_m1, y_def)
    e_def, e_m1, e_cpu = ((y.double() - exact).abs().max().item() for y in (y_def, y_m1, y_cpu))
    # both forms are closer to the exact network than torch-CPU (the error-budget rule of §5); between the two the maxima over
    # a few hundred outputs are within an ulp of the output scale of each other (5.96e-7 vs 4.77e-7 = 1.25 vs 1 ulp here)
    q = float(np.spacing(np.float32(exact.abs().max().item())))
    assert e_def <= e_cpu and e_m1 <= e_cpu and abs(e_def - e_m1) <= q, (e_def, e_m1, e_cpu, q)
