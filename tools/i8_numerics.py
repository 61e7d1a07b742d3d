"""Numerics of an nn.Linear evaluated on the i8 matrix pipe (model first): every fp32 operand as S signed 8-bit
slices under a power-of-two scale per row (activation row / weight row), the slice products with i + j < S accumulated EXACTLY in
int32 per level i + j, one conversion at the end.  Pure numpy (exact integer arithmetic, no GPU): rms / max error of a dot product
in ulps of the fp32 result's scale against a long-double reference, for

  * the operands of the MLP's layers (utils/mlp.py:8-28): hash-initialised weights as the bench uses them and the REAL activations
    a capture-volume row produces at every layer (LeakyReLU(0.1) outputs: their amax / rms per row decides what a row scale costs),
    and gaussian operands at K = 416 ... 3072 for comparison with profiles/r04_sb16_numerics.txt (split-bf16: 0.24-0.26 ulp rms),
  * where the activation scale comes from: the row's own amax (needs a pass over the row before it can be sliced), a conservative
    bound 2^k above it (what a scale predicted without that pass would be), or one scale per 64-deep K block (what a producer's
    epilogue knows without a pass over the row; the consumer then has to rescale every block).

usage: python tools/i8_numerics.py [out.txt]
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = '3d_multi_pose_estimator_amd'


def pow2_at_least(x):
    """smallest power of two >= x * 128 / 127 (x > 0), elementwise: balanced 8-bit digits reach 127 / 128 (1 + 1 / 256 + ...) of the scale"""
    return np.exp2(np.ceil(np.log2(np.maximum(x, np.float64(1e-300)) * (128.0 / 127.0))))


def slices(x, scale, S):
    """x / scale in (-1, 1] as S balanced signed 8-bit digits: x ~ scale * sum_i d_i 2^(-7 - 8 i), d_i in [-128, 127];
    returns int64 [S, ...] and the exact value the digits stand for (float64)."""
    q = np.rint(np.asarray(x, np.float64) / scale * 2.0 ** (7 + 8 * (S - 1))).astype(np.int64)      # fixed point, 8 S - 1 bits + sign
    digits = []
    r = q.copy()
    for i in range(S - 1, -1, -1):                     # least significant first, balanced (carry into the next digit)
        d = ((r + 128) % 256) - 128
        digits.append(d)
        r = (r - d) // 256
    digits = digits[::-1]
    if np.any(np.abs(r) > 0):
        raise AssertionError('slice overflow')
    val = sum(d.astype(np.float64) * 2.0 ** (-7 - 8 * i) for i, d in enumerate(digits)) * scale
    return np.stack(digits), val


def linear_i8(A, W, S=4, a_scale='row', k_guard=0, block=0):
    """A [M, K] fp32, W [N, K] fp32 -> A W^T with S slices per operand, products i + j < S, exact integer sums per level."""
    A64, W64 = A.astype(np.float64), W.astype(np.float64)
    sw = pow2_at_least(np.abs(W64).max(axis=1, keepdims=True))                  # per weight row (offline)
    dw, _ = slices(W64, sw, S)
    M, K = A.shape
    out = np.zeros((M, W.shape[0]), np.float64)
    blocks = [(0, K)] if not block else [(k0, min(K, k0 + block)) for k0 in range(0, K, block)]
    for k0, k1 in blocks:
        a = A64[:, k0:k1]
        sa = pow2_at_least(np.abs(a).max(axis=1, keepdims=True)) * 2.0 ** k_guard
        da, _ = slices(a, sa, S)
        acc = np.zeros((M, W.shape[0]), np.float64)
        for i in range(S):
            for j in range(S - i):
                lvl = (da[i] @ dw[j][:, k0:k1].T)                                # exact in int64; int32 on the device (checked below)
                assert np.abs(lvl).max() < 2 ** 31
                acc += lvl.astype(np.float64) * 2.0 ** (-14 - 8 * (i + j))      # exact: integers below 2^53 times powers of two ... summed in f64
        out += acc * sa * sw.T
    return out


def ulp_stats(got, ref):
    """error in ulps of the fp32 grid at the RESULT'S typical magnitude (rms of the reference per launch), as tools/sb16_numerics"""
    scale = np.sqrt(np.mean(np.asarray(ref, np.float64) ** 2))
    ulp = 2.0 ** (np.floor(np.log2(scale)) - 23)
    err = (np.asarray(got, np.float64) - np.asarray(ref, np.float64)) / ulp
    return float(np.sqrt(np.mean(err ** 2))), float(np.abs(err).max())


def reference(A, W):
    return (A.astype(np.longdouble) @ W.astype(np.longdouble).T).astype(np.float64)


def fp32_chain(A, W):
    """what a plain fp32 accumulation gives (numpy's pairwise float32 matmul): the scale of 'fp32 accuracy'"""
    return (A.astype(np.float32) @ W.astype(np.float32).T).astype(np.float64)


def mlp_layer_operands(rows=24):
    """(name, activations [rows, K], weights [N', K]) for every layer of the hash-initialised MLP on capture-volume rows"""
    syn = importlib.import_module(PKG + '.synthetic')
    sd = syn.mlp_state_dict(11, 1260)
    g = np.random.default_rng(5)
    x = (g.standard_normal((rows, 1260)) * 0.3).astype(np.float32)
    x[:, ::14] = (g.random((rows, 90)) > 0.3).astype(np.float32)               # validity flags
    keys = sorted({int(k.split('.')[1]) for k in sd})
    out = []
    h = x
    for n, k in enumerate(keys):
        w, b = np.asarray(sd['layers.%d.weight' % k], np.float32), np.asarray(sd['layers.%d.bias' % k], np.float32)
        out.append(('mlp layer %d  K=%d' % (n, w.shape[1]), h, w[:256]))        # 256 output features are sample enough
        y = (h.astype(np.float64) @ w.astype(np.float64).T + b).astype(np.float32)
        h = np.where(y > 0, y, y * np.float32(0.1)).astype(np.float32) if n != len(keys) - 1 else y
    return out


def main():
    lines = []

    def emit(s=''):
        print(s)
        lines.append(s)
    emit(__doc__.split('usage:')[0].strip())
    emit()
    emit('%-28s %9s | %-17s %-17s %-17s %-17s %-17s | %-17s' % ('operands', 'amax/rms', 'S=4 row amax', 'S=4 bound 2^4', 'S=4 bound 2^8', 'S=4 per 64 block',
                                                               'S=3 row amax', 'fp32 chain (numpy)'))
    emit('%-28s %9s | %s' % ('', 'of A rows', 'rms / max error of a launch, ulps of the fp32 grid at the result\'s scale'))
    cases = []
    g = np.random.default_rng(1)
    for K in (416, 1024, 1280, 3072):
        A = g.standard_normal((24, K)).astype(np.float32)
        A = np.where(A > 0, A, A * np.float32(0.1))                              # LeakyReLU-shaped
        W = (g.standard_normal((256, K)) / np.sqrt(K)).astype(np.float32)
        cases.append(('gaussian, leaky  K=%d' % K, A, W))
    cases += mlp_layer_operands()
    for name, A, W in cases:
        ref = reference(A, W)
        ratio = float(np.mean(np.abs(A).max(axis=1) / np.sqrt(np.mean(A.astype(np.float64) ** 2, axis=1))))
        cols = []
        for kw in (dict(S=4), dict(S=4, k_guard=4), dict(S=4, k_guard=8), dict(S=4, block=64), dict(S=3)):
            r, m = ulp_stats(linear_i8(A, W, **kw), ref)
            cols.append('%6.3f / %-8.2f' % (r, m))
        r, m = ulp_stats(fp32_chain(A, W), ref)
        emit('%-28s %9.1f | %s | %6.3f / %-8.2f' % (name, ratio, ' '.join(cols), r, m))
    emit()
    emit('reading: "ulps" are of the fp32 grid at the rms of the launch\'s results, so 0.29 rms is what rounding an exact result to fp32 costs and is')
    emit('not in these numbers (the integer sums are exact and converted once: add 0.29 in quadrature for the stored fp32 result).')
    if len(sys.argv) > 1:
        with open(sys.argv[1], 'w') as fh:
            fh.write('\n'.join(lines) + '\n')


if __name__ == '__main__':
    main()
