"""BASELINE configs[3] at its real per-GPU size: 100 000 frames of 5 views x 10 persons x 18 joints sharded over 8 GPUs is 12 500
frames per rank.  One engine with THAT capacity (66 GB of activation workspace), a 2 500-frame batch of 5 x 10 frames through it,
and the size-independent properties of the path (test/metrics_from_model.py:120-300 reads nothing but its own frame: frames are
independent units): any split or reordering gives the same bits per frame, the clustering output is structurally sound on every
frame, a 20-frame sample agrees with the CPU oracle, both 3D stages run."""
import numpy as np
import pytest
import torch

from conftest import oracle, pkg

pytestmark = pytest.mark.gpu


def test_config3_shard_capacity_and_a_2500_frame_batch(calib, gat_weights, mlp_weights):
    onp = oracle()
    syn = pkg('synthetic')
    sd, prm = gat_weights
    eng = pkg('pipeline').Engine(calib.params, calib, max_frames=12500, max_persons_per_camera=10)
    try:
        eng.load_gat(sd, prm)
        eng.load_mlp(mlp_weights)
        specs = [syn.FrameSpec(persons=10), syn.FrameSpec(persons=10, noise_px=2.0, joint_drop=0.1), syn.FrameSpec(persons=9),
                 syn.FrameSpec(persons=10, cameras=['trackere', 'trackerb', 'trackera', 'trackerd', 'trackerc'])]
        uniq = [onp.processed_input(syn.make_frame(calib, 5000 + i, specs[i % 4])[0]) for i in range(50)]
        B = 2500
        frames = [uniq[(7 * i) % 50] for i in range(B)]

        def run(fr):
            db = eng.to_device(eng.pack(fr))
            sc, p, n = eng.match(db)
            poses, valid = eng.mlp3d(db, p, n)
            tri, jv = eng.triangulate(db, p, n)
            eng.sync_status()
            return db, sc.cpu().numpy(), p.cpu().numpy(), n.cpu().numpy(), poses.cpu().numpy(), tri.cpu().numpy(), valid.cpu().numpy()
        db, sc, p, n, poses, tri, valid = run(frames)
        assert db.n_frames == B and db.n_heads >= 45 * B and db.n_edge_nodes >= 800 * B
        assert n.min() >= 5 and n.mean() > 8                      # ten people walk through the room: most of them are found
        # (1a) five chunks of 500: same bits per frame
        e_off = db.host.frame_en_off
        for c in range(0, B, 500):
            _, sc2, p2, n2, poses2, tri2, _ = run(frames[c:c + 500])
            assert np.array_equal(sc2, sc[e_off[c]:e_off[c + 500]])
            assert np.array_equal(p2, p[c:c + 500]) and np.array_equal(n2, n[c:c + 500])
            assert np.array_equal(poses2, poses[c:c + 500]) and np.array_equal(tri2, tri[c:c + 500])
        # (1b) reversed order
        _, _, pr, nr, posesr, trir, _ = run(frames[::-1])
        assert np.array_equal(pr[::-1], p) and np.array_equal(nr[::-1], n)
        assert np.array_equal(posesr[::-1], poses) and np.array_equal(trir[::-1], tri)
        # (1c) a frame on its own (the small-batch launches) -- the same bits again
        for f in (0, 1234, B - 1):
            _, sc1, p1, n1, poses1, tri1, _ = run(frames[f:f + 1])
            assert np.array_equal(sc1, sc[e_off[f]:e_off[f + 1]]) and np.array_equal(p1[0], p[f]) and n1[0] == n[f]
            assert np.array_equal(poses1[0], poses[f]) and np.array_equal(tri1[0], tri[f])
        # (2) structure on every frame: a head belongs to at most one person, a person spans >= 2 cameras, heads sit in their camera's column
        for f in range(B):
            h0, H, e0, M = db.host.frame_counts(f)
            seen = set()
            for k in range(n[f]):
                members = [(c, h) for c, h in enumerate(p[f, k]) if h >= 0]
                assert len(members) >= calib.params.min_number_of_views
                for c, h in members:
                    assert h < H and db.host.head_cam[h0 + h] == c and h not in seen
                    seen.add(h)
            assert (p[f, n[f]:] == -1).all()
            assert valid[f, :n[f]].all() and np.isfinite(poses[f, :n[f]]).all() and not poses[f, n[f]:].any()
        # (3) the oracle on 20 frames: scores to fp32 noise, the clustering of the oracle on the device's scores exactly
        sm = list(calib.params.used_cameras_skeleton_matching)
        for f in range(0, B, B // 20):
            h0, H, e0, M = db.host.frame_counts(f)
            res = onp.run_frame(frames[f], calib, sd, prm, mlp_weights, mode='mlp')
            np.testing.assert_allclose(sc[e0:e0 + M], res['scores'], rtol=0, atol=2e-5)
            head_cam = [sm.index(c) for c in res['graph']['nodes_camera'][:H]]
            own = onp.cluster(sc[e0:e0 + M], res['graph']['pairs'], H, head_cam, len(sm))
            assert n[f] == len(own) and np.array_equal(p[f, :len(own)], np.array(own, np.int32).reshape(-1, len(sm)))
    finally:
        eng.close()
