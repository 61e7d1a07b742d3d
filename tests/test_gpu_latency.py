"""The small-batch ("latency") kernels of csrc/lat.hip against the batch kernels: the same arithmetic in the same order, so the same
bits -- a row does not change with the batch it travels in, nor with the route a batch size selects (MPE_LATENCY_PATH=0 sends small
batches through the batch path's own small-batch kernels)."""
import numpy as np
import pytest
import torch

from conftest import pkg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def engine(calib, gat_weights, mlp_weights):
    eng = pkg('pipeline').Engine(calib.params, calib, max_frames=64, max_persons_per_camera=10)
    sd, prm = gat_weights
    eng.load_gat(sd, prm)
    eng.load_mlp(mlp_weights)
    yield eng
    eng.close()


def test_mlp_small_batches_keep_their_bits_on_either_route(engine, mlp_weights, monkeypatch):
    """mpe_mlp_forward at 1 ... 800 rows (the K-split plane kernel k_linear_sb_ks up to 32 / 48 row tiles by layer width): same rows as in a batch of 1600 (tile kernels),
    with the small-batch route on and off, in the default and in the maximum-accuracy mode."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1600, 1260, generator=g) * 0.3            # (the engine holds 64 frames x 25 persons)
    try:
        for max_acc in (False, True):
            engine.set_precision(mlp_max_accuracy=max_acc)
            big = engine.mlp_forward(x.cuda()).cpu()
            for m in (1, 4, 10, 33, 128, 200, 320, 520, 800):  # (<= 512 rows: the K-split plane kernel on every layer; 520: on the 1024-wide ones; 800: tile kernels)
                monkeypatch.delenv('MPE_LATENCY_PATH', raising=False)
                small = engine.mlp_forward(x[:m].cuda()).cpu()
                monkeypatch.setenv('MPE_LATENCY_PATH', '0')
                old = engine.mlp_forward(x[:m].cuda()).cpu()
                monkeypatch.delenv('MPE_LATENCY_PATH', raising=False)
                assert torch.equal(small, big[:m]), (max_acc, m, (small - big[:m]).abs().max().item())
                assert torch.equal(old, big[:m]), (max_acc, m)
    finally:
        engine.set_precision()


def _frames(calib, n, persons=(4, 2, 5, 1, 3, 4, 6, 4), start=100):
    syn = pkg('synthetic')
    from conftest import oracle
    out = []
    for i in range(n):
        spec = syn.FrameSpec(persons=persons[i % len(persons)], empty_cameras=('trackerc',) if i % 5 == 3 else (), joint_drop=0.15 if i % 2 else 0.0)
        out.append(oracle().processed_input(syn.make_frame(calib, start + i, spec)[0]))
    return out


@pytest.mark.parametrize('n_frames', [1, 3, 8, 13, 16])
def test_small_batches_on_the_latency_launches_give_the_batch_paths_bits(engine, calib, n_frames, monkeypatch):
    """A batch of at most sixteen frames takes the latency launches of the matching stage (front + layer-0 fc1 in one launch, the
    plane-fed GEMMs, both halves of the attention stage in one launch): scores, persons and poses must be the bits of (a) the same
    batch with the route switched off (the batch path's small-batch kernels) and (b) the same frames travelling inside a batch of
    40 (tile kernels, fused attention)."""
    frames = _frames(calib, n_frames)
    filler = _frames(calib, 40 - n_frames, start=300)

    def run(fr):
        db = engine.to_device(engine.pack(fr))
        scores, persons, n_persons = engine.match(db)
        poses, valid = engine.mlp3d(db, persons, n_persons)
        engine.sync_status()
        e_off = [db.host.frame_counts(f)[2:] for f in range(len(fr))]            # (first edge-node, edge-nodes) per frame
        return scores.cpu().numpy(), persons.cpu().numpy(), n_persons.cpu().numpy(), poses.cpu().numpy(), e_off

    monkeypatch.delenv('MPE_LATENCY_PATH', raising=False)
    lat = run(frames)
    monkeypatch.setenv('MPE_LATENCY_PATH', '0')
    off = run(frames)
    monkeypatch.delenv('MPE_LATENCY_PATH', raising=False)
    big = run(frames + filler)
    assert lat[0].shape == off[0].shape and np.array_equal(lat[0], off[0]), np.abs(lat[0] - off[0]).max()
    for k in (1, 2, 3):
        assert np.array_equal(lat[k], off[k])
    m_tot = sum(m for _, m in lat[4])
    assert np.array_equal(lat[0][:m_tot], big[0][:m_tot]), np.abs(lat[0][:m_tot] - big[0][:m_tot]).max()
    for k in (1, 2, 3):
        assert np.array_equal(lat[k], big[k][:n_frames])


def test_row_kernel_fetches_pairs_only_for_the_batch_they_were_solved_for(engine, calib, monkeypatch):
    """mpe_match_batch of a small batch solves every cross-camera pair of that batch beside its clustering launch and tags the buffer
    with the batch's arrays; mpe_mlp3d_batch fetches the pairs when it is handed the tagged batch and solves them itself otherwise.
    Either way the poses are those of the route without any of this (MPE_LATENCY_PATH=0)."""
    fa, fb = _frames(calib, 3, start=500), _frames(calib, 3, persons=(5, 3, 4), start=600)
    monkeypatch.setenv('MPE_LATENCY_PATH', '0')
    dba, dbb = engine.to_device(engine.pack(fa)), engine.to_device(engine.pack(fb))
    want = {}
    for name, db in (('a', dba), ('b', dbb)):
        _, p, n = engine.match(db)
        want[name] = (p.clone(), n.clone(), engine.mlp3d(db, p, n)[0].clone())
    monkeypatch.delenv('MPE_LATENCY_PATH', raising=False)
    _, pb, nb = engine.match(dbb)                      # tags one buffer with batch b
    assert torch.equal(pb, want['b'][0]) and torch.equal(nb, want['b'][1])
    _, pa, na = engine.match(dba)                      # tags the other with batch a
    assert torch.equal(engine.mlp3d(dbb, pb, nb)[0], want['b'][2])          # fetched from b's buffer
    assert torch.equal(engine.mlp3d(dba, pa, na)[0], want['a'][2])          # fetched from a's
    engine.match(dba)
    engine.match(dba)                                  # both buffers have held batch a since: b's tag is gone
    assert torch.equal(engine.mlp3d(dbb, pb, nb)[0], want['b'][2])          # solved by the row kernel itself
    engine.sync_status()


def test_small_batch_routes_on_a_random_sweep_of_frames(calib, gat_weights, mlp_weights):
    """240 random frames (1 ... 6 persons, dropped joints, pixel noise, empty cameras, single-camera frames that have no graph) in groups
    of 1 ... 17 (up to 16: the latency launches; 17: the batch path with the K-split MLP kernel) against the same frames in ONE batch of 240 (tile kernels, fused attention, the wave
    clustering kernel, the row kernel's own pair solves): scores, persons and poses bit for bit, frame by frame -- the last layer's
    scores + clustering + pair solves of k_lat_tail and the plane-fed GEMMs on shapes the golden frames do not hold."""
    syn = pkg('synthetic')
    from conftest import oracle
    eng = pkg('pipeline').Engine(calib.params, calib, max_frames=240, max_persons_per_camera=6)
    sd, prm = gat_weights
    eng.load_gat(sd, prm)
    eng.load_mlp(mlp_weights)
    try:
        cams = list(calib.params.used_cameras_skeleton_matching)
        frames = []
        for i in range(240):
            empty = ()
            if i % 7 == 3:
                empty = (cams[i % len(cams)],)
            elif i % 31 == 5:
                empty = tuple(cams[1:])                      # one camera left: heads, no edge-node
            elif i % 43 == 9:
                empty = tuple(cams)                          # nothing at all
            spec = syn.FrameSpec(persons=1 + i % 6, empty_cameras=empty, joint_drop=(0.0, 0.15, 0.4)[i % 3], noise_px=(0.0, 1.0, 3.0)[(i // 3) % 3])
            frames.append(oracle().processed_input(syn.make_frame(calib, 7000 + i, spec)[0]))
        db = eng.to_device(eng.pack(frames))
        sc, pe, npers = eng.match(db)
        po, va = eng.mlp3d(db, pe, npers)
        eng.sync_status()
        sc, pe, npers, po, va = sc.cpu().numpy(), pe.cpu().numpy(), npers.cpu().numpy(), po.cpu().numpy(), va.cpu().numpy()
        off = [db.host.frame_counts(f) for f in range(len(frames))]
        sizes, i, k, groups = (1, 2, 3, 5, 8, 4, 16, 1, 7, 11, 6, 13, 9, 17), 0, 0, 0
        while i < len(frames):
            n = min(sizes[k % len(sizes)], len(frames) - i)
            k += 1
            g = eng.to_device(eng.pack(frames[i:i + n]))
            s2, p2, n2 = eng.match(g)
            q2, v2 = eng.mlp3d(g, p2, n2)
            eng.sync_status()
            s2, p2, n2, q2, v2 = s2.cpu().numpy(), p2.cpu().numpy(), n2.cpu().numpy(), q2.cpu().numpy(), v2.cpu().numpy()
            e = 0
            for j in range(n):
                _, _, e0, M = off[i + j]
                assert np.array_equal(s2[e:e + M], sc[e0:e0 + M]), (i + j, n)
                e += M
                cnt = int(npers[i + j])
                assert n2[j] == cnt and np.array_equal(p2[j, :cnt], pe[i + j, :cnt]), (i + j, n)
                assert np.array_equal(v2[j], va[i + j]) and np.array_equal(q2[j][v2[j] != 0], po[i + j][va[i + j] != 0]), (i + j, n)
            i += n
            groups += 1
        assert groups >= 30 and int(npers.sum()) > 300
    finally:
        eng.close()


def test_small_batch_routes_at_the_largest_frames_they_take(calib, gat_weights, mlp_weights, monkeypatch):
    """5 x 10-person frames (BASELINE configs[3]'s frame: 50 skeletons, 1000 cross-camera pairs, 1050 graph nodes) in groups of 1, 2, 8, 15
    and 16: up to 15 frames = 15 750 nodes the latency launches take the batch (their row-tile loop forms, the tail launch with 50-head
    frames and 45 x 45 pair solves per frame), 16 frames = 16 800 nodes exceed their row capacity and go to the batch kernels.  Scores,
    persons and poses equal the same frames inside one batch of 32, and the route switched off, bit for bit."""
    syn = pkg('synthetic')
    from conftest import oracle
    eng = pkg('pipeline').Engine(calib.params, calib, max_frames=32, max_persons_per_camera=10)
    sd, prm = gat_weights
    eng.load_gat(sd, prm)
    eng.load_mlp(mlp_weights)
    try:
        frames = [oracle().processed_input(syn.make_frame(calib, 8800 + i, syn.FrameSpec(persons=10 if i % 4 else 9, joint_drop=0.05 * (i % 3)))[0])
                  for i in range(32)]
        db = eng.to_device(eng.pack(frames))
        assert db.n_heads >= 32 * 40 and db.n_edge_nodes >= 32 * 700
        sc, pe, npers = eng.match(db)
        po, va = eng.mlp3d(db, pe, npers)
        eng.sync_status()
        sc, pe, npers, po, va = sc.cpu().numpy(), pe.cpu().numpy(), npers.cpu().numpy(), po.cpu().numpy(), va.cpu().numpy()
        off = [db.host.frame_counts(f) for f in range(len(frames))]
        assert int(npers.sum()) >= 32 * 5
        for route in ('1', '0'):
            monkeypatch.setenv('MPE_LATENCY_PATH', route)
            i = 0
            for n in (1, 2, 8, 15, 6):                   # 32 frames; then 16 from the start
                g = eng.to_device(eng.pack(frames[i:i + n]))
                s2, p2, n2 = eng.match(g)
                q2, v2 = eng.mlp3d(g, p2, n2)
                eng.sync_status()
                s2, p2, n2, q2, v2 = s2.cpu().numpy(), p2.cpu().numpy(), n2.cpu().numpy(), q2.cpu().numpy(), v2.cpu().numpy()
                e = 0
                for j in range(n):
                    _, _, e0, M = off[i + j]
                    assert np.array_equal(s2[e:e + M], sc[e0:e0 + M]), (route, i + j, n)
                    e += M
                    cnt = int(npers[i + j])
                    assert n2[j] == cnt and np.array_equal(p2[j, :cnt], pe[i + j, :cnt]), (route, i + j, n)
                    assert np.array_equal(v2[j], va[i + j]) and np.array_equal(q2[j][v2[j] != 0], po[i + j][va[i + j] != 0]), (route, i + j, n)
                i += n
            g = eng.to_device(eng.pack(frames[:16]))
            s2, p2, n2 = eng.match(g)
            q2, v2 = eng.mlp3d(g, p2, n2)
            eng.sync_status()
            M16 = sum(off[j][3] for j in range(16))
            assert np.array_equal(s2.cpu().numpy()[:M16], sc[:M16]) and np.array_equal(n2.cpu().numpy(), npers[:16]), route
            assert np.array_equal(q2.cpu().numpy()[v2.cpu().numpy() != 0], po[:16][va[:16] != 0]), route
    finally:
        eng.close()
