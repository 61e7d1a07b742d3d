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
    """mpe_mlp_forward at 1 ... 128 rows (the K-split plane kernel k_linear_sb_ks): same rows as in a batch of 1600 (tile kernels),
    with the small-batch route on and off, in the default and in the maximum-accuracy mode."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1600, 1260, generator=g) * 0.3            # (the engine holds 64 frames x 25 persons)
    try:
        for max_acc in (False, True):
            engine.set_precision(mlp_max_accuracy=max_acc)
            big = engine.mlp_forward(x.cuda()).cpu()
            for m in (1, 4, 10, 33, 128):
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


@pytest.mark.parametrize('n_frames', [1, 3, 8])
def test_small_batches_on_the_latency_launches_give_the_batch_paths_bits(engine, calib, n_frames, monkeypatch):
    """A batch of at most eight frames takes the latency launches of the matching stage (front + layer-0 fc1 in one launch, the
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
