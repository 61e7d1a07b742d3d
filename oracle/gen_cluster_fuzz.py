"""TEST INFRASTRUCTURE (build container only): MORE known answers of the reference's greedy clustering than tests/golden/cluster_cases.npz
holds -- the reference's own get_person_proposal_from_network_output (skeleton_matching_utils.py:12-132, the real file through
oracle/refenv.py) on fresh random and adversarial score vectors: uniform, trained-like with errors, saturated ties, everything-matches,
coarse grids, near-threshold values, and frames with up to ten skeletons per camera.  Writes tests/golden/cluster_cases_fuzz.npz
(inputs + the reference's persons; data only).

    python oracle/gen_cluster_fuzz.py [cases: 1500] [seed: 31]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import refenv  # noqa: E402
from gen_golden import _G, quiet  # noqa: E402


def main(n_cases=1500, seed=31):
    import torch
    import oracle_np as onp
    with quiet():
        ref = refenv.load()
    cams = list(ref['parameters'].parameters.used_cameras_skeleton_matching)
    fn = ref['skeleton_matching_utils'].get_person_proposal_from_network_output
    rng = np.random.default_rng(seed)
    arrays, k = {}, 0
    while k < n_cases:
        ncam = int(rng.integers(2, 6))
        order = list(rng.permutation(5)[:ncam])
        top = 11 if k % 5 == 0 else 6 if k % 3 == 0 else 4
        counts = [int(rng.integers(1, top)) for _ in order]
        slots, hid = [], 0
        for c, n in zip(order, counts):
            slots.append((cams[c], list(range(hid, hid + n))))
            hid += n
        N, src, dst, pairs = onp.topology(slots)
        H, M = hid, N - hid
        if M == 0:
            continue
        kind = k % 7
        if kind == 0:
            sc = rng.uniform(0, 1, M)
        elif kind == 1:
            owner = [int(rng.integers(0, max(counts))) for _ in range(H)]
            sc = np.clip(np.array([(0.9 if owner[a] == owner[b] else 0.1) for a, b in pairs]) + rng.normal(0, 0.25, M), 0, 1)
        elif kind == 2:
            sc = rng.choice([1.0, 1.0, 0.99999994, 0.75, 0.5, 0.4999], M)
        elif kind == 3:
            sc = 0.5 + 0.5 * rng.uniform(0, 1, M) ** 0.3
        elif kind == 4:
            sc = np.round(rng.uniform(0.3, 1.0, M), 1)
        elif kind == 5:      # values crowding the threshold from both sides, in float32 steps
            sc = np.float32(0.5) + rng.integers(-3, 4, M).astype(np.float32) * np.float32(2.0 ** -24) * rng.choice([1, 1, 64, 4096], M)
        else:                # a correct pairing with a few confidently wrong links (merges of two persons, repeated cameras in a component)
            owner = [int(rng.integers(0, max(counts))) for _ in range(H)]
            sc = np.array([(0.95 if owner[a] == owner[b] else 0.02) for a, b in pairs]) + rng.normal(0, 0.01, M)
            wrong = rng.random(M) < 0.06
            sc = np.clip(np.where(wrong, 0.97 + rng.normal(0, 0.01, M), sc), 0, 1)
        sc32 = np.asarray(sc, np.float32)
        outputs = torch.zeros(N)
        outputs[H:] = torch.from_numpy(sc32)
        nodes_camera = [s[0] for s in slots for _ in s[1]] + [''] * M
        with quiet():
            res = fn(outputs, _G(src.tolist(), dst.tolist()), torch.arange(H, N), nodes_camera, None, 0.5)
        arrays['c%d_slot_cam' % k] = np.array([cams.index(s[0]) for s in slots], np.int32)
        arrays['c%d_slot_n' % k] = np.array(counts, np.int32)
        arrays['c%d_scores' % k] = sc32
        arrays['c%d_persons' % k] = np.array([[(-1 if r[c] is None else r[c]) for c in cams] for r in res], np.int32).reshape(-1, len(cams))
        k += 1
    out = os.path.join(ROOT, 'tests', 'golden', 'cluster_cases_fuzz.npz')
    np.savez_compressed(out, n=np.int64(k), seed=np.int64(seed), **arrays)
    print('cluster fuzz cases', k, 'seed', seed, '->', out, os.path.getsize(out), 'bytes')


if __name__ == '__main__':
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 1500, int(sys.argv[2]) if len(sys.argv) > 2 else 31)
