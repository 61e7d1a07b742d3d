"""Do the wave-per-tile kernels and the tile kernels give the same bits?  Scores (heads + edge-nodes) of fixture frames in a
process with MPE_SKINNY_WAVES=0 (tile kernels at every batch size) against a plain one, for the default GAT arithmetic and the
f64-sum mode; per-layer activations through mpe_gat_layer for the first difference.
    python tests/checkers/skinny_vs_tile.py dump OUT.npz     (run twice, with and without MPE_SKINNY_WAVES=0)
    python tests/checkers/skinny_vs_tile.py cmp A.npz B.npz
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))


def dump(path):
    import torch
    from conftest import env, pkg
    from test_gpu_stages import ALL_CASES, load_case, engine_for, oracle, _dense_features
    onp = oracle()
    out = {}
    for variant, name in ALL_CASES:
        e = env(variant)
        eng = engine_for(variant)
        sd, prm = e.gat
        arr, frames = load_case(name, variant)
        for n, frame in enumerate(frames):
            db = eng.to_device(eng.pack([onp.processed_input(frame)]))
            for mode in ('def', 'f64'):
                try:
                    if mode == 'f64':
                        eng.set_precision(gat_acc64=True)
                    sc, sh = eng.gat_scores(db, heads=True)
                    out['%s/%s/%d/%s' % (variant, name, n, mode)] = np.concatenate([sh.cpu().numpy(), sc.cpu().numpy()])
                    feats = _dense_features(arr, 'f%d_' % n, e.meta['num_feats'])
                    x = torch.from_numpy(feats).cuda()
                    for l in range(prm['gnn_layers'] - 1):
                        x = eng.gat_layer(db, l, x, activation=0)
                        out['%s/%s/%d/%s/layer%d' % (variant, name, n, mode, l)] = x.cpu().numpy()
                finally:
                    eng.set_precision()
    np.savez(path, **out)


def cmp(a, b):
    A, B = np.load(a), np.load(b)
    bad = 0
    for k in A.files:
        if not np.array_equal(A[k], B[k]):
            bad += 1
            d = np.abs(A[k].astype(np.float64) - B[k])
            print('DIFF %-50s max %.3e at %s of %s' % (k, d.max(), np.unravel_index(d.argmax(), d.shape), A[k].shape))
    print('%d of %d arrays differ' % (bad, len(A.files)))


if __name__ == '__main__':
    if sys.argv[1] == 'dump':
        dump(sys.argv[2])
    else:
        cmp(sys.argv[2], sys.argv[3])
