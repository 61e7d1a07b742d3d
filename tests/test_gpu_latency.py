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


@pytest.mark.parametrize('k,n,slope', [(1260, 3072, 0.1), (3072, 2048, 0.1), (1024, 54, None), (902, 400, None), (6016, 64, 0.1), (96, 48, 0.1)])
def test_latency_linear_gives_the_tile_kernels_bits(engine, k, n, slope):
    """k_linear_lat_f64 (fp32 weights split in registers, all fragments of a wave requested at once, units dealt to eight waves,
    ordered f64 reduction) against k_linear_sb (three bf16 planes, 256-row tiles): identical rows at 1 ... 128 rows, in both flush
    cadences; K = 6016 runs two rounds per workgroup (four with a flush per stage), K = 902 has an odd stage count."""
    g = torch.Generator().manual_seed(k * 7 + n)
    x = torch.randn(3000, k, generator=g)
    w = (torch.randn(n, k, generator=g) / np.sqrt(k)).numpy()
    b = torch.randn(n, generator=g).numpy()
    for per_stage in (False, True):
        big = engine.linear(x.cuda(), w, b, slope, split=True, split_flush_per_stage=per_stage).cpu()
        for m in (1, 4, 16, 17, 40, 128):
            small = engine.linear(x[:m].cuda(), w, b, slope, lat=True, split_flush_per_stage=per_stage).cpu()
            assert torch.equal(small, big[:m]), (per_stage, m, (small - big[:m]).abs().max().item())


def test_mlp_small_batches_take_the_latency_kernels_and_keep_their_bits(engine, mlp_weights, monkeypatch):
    """mpe_mlp_forward routes batches of at most 128 rows to lat.hip: same rows as in a batch of 1600 (tile kernels) and as with the
    route switched off, in the default and in the maximum-accuracy mode."""
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
