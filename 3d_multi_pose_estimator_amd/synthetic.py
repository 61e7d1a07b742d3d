"""Deterministic synthetic inputs and weights.

No dataset or checkpoint of the reference is reachable offline, so both the parity
fixtures and the benchmark use

  * frames in the reference's JSON wire format (panoptic_conversor/
    get_joints_from_panoptic_model_multi.py:231-236,281,287): per camera
    ``[json-string of skeleton list, timestamp, 'no_image', bodies_3D]``, skeleton =
    ``{joint_id_str: [joint_id, x, y, valid, prob]}``; 2D points are produced by
    projecting random 3D skeletons with the Panoptic lens model (the published
    k1,k2,p1,p2,k3 model, same as OpenCV's; reference panutils.py:14-26) so that
    undistort + DLT recovers the 3D ground truth;
  * GAT / MLP weights from a counter-based integer hash (splitmix64), so the values
    are identical on every machine and torch version (no torch.manual_seed).

Everything here is host-side numpy; nothing is on the timed path.
"""
import json

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def hash_uniform(seed, stream, n):
    """n doubles in [0,1): u[i] = top 53 bits of splitmix64(mix(seed, stream) + i)."""
    with np.errstate(over='ignore'):
        base = _splitmix64(np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(stream))
        idx = np.arange(n, dtype=np.uint64)
        z = _splitmix64(_splitmix64(base + idx))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def hash_symmetric(seed, stream, shape, bound):
    """float32 array uniform in [-bound, bound)."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = hash_uniform(seed, stream, n)
    return ((u * 2.0 - 1.0) * bound).astype(np.float32).reshape(shape)


# --------------------------------------------------------------------------------------
# weights
# --------------------------------------------------------------------------------------

GAT_HIDDEN = [40, 40, 40, 30]        # reference train_skeleton_matching.py:40-57 (deployed arch)
GAT_HEADS = [10, 10, 8, 5]
GAT_ALPHA = 0.15
GAT_LAYERS = 5
MLP_WIDTHS = [3072, 3072, 2048, 2048, 1024, 1024, 1024, 1024]   # reference utils/mlp.py:8-28
MLP_SLOPE = 0.1


def gat_layer_dims(num_feats, hidden=None, heads=None, n_classes=1):
    """[(in_dim, num_heads, out_dim)] per layer, as GAT2.__init__ builds them
    (reference gat2.py:91-135)."""
    hidden = GAT_HIDDEN if hidden is None else hidden
    heads = GAT_HEADS if heads is None else heads
    dims = [(num_feats, heads[0], hidden[0])]
    for l in range(1, len(hidden)):
        dims.append((hidden[l - 1] * heads[l - 1], heads[l], hidden[l]))
    dims.append((hidden[-1] * heads[-1], 1, n_classes))
    return dims


def gat_state_dict(seed, num_feats, hidden=None, heads=None, logit_gain=1.0, logit_shift=0.0):
    """Deterministic GAT2 weights under the reference's state-dict names
    (``layers.{l}.{attn_l,attn_r,fc1.weight,fc1.bias,fc2.weight,fc2.bias}``).

    Scale follows xavier(gain=1.414) variance (reference gat2.py:35-38) but with a
    uniform distribution.  `logit_gain` multiplies the last layer's fc2 so that the
    sigmoid outputs spread over (0,1) instead of sitting in a narrow band (needed to
    keep sorted-score gaps far above fp32 reordering noise in the parity tests);
    `logit_shift` is added to the last layer's fc2 bias before the gain (centres the logits).
    """
    sd = {}
    dims = gat_layer_dims(num_feats, hidden, heads)
    for l, (din, nh, dout) in enumerate(dims):
        s = 1000 * (l + 1)
        std1 = 1.414 * np.sqrt(2.0 / (din + din))
        std2 = 1.414 * np.sqrt(2.0 / (din + nh * dout))
        stda = 1.414 * np.sqrt(2.0 / (dout + nh))
        g2 = logit_gain if l == len(dims) - 1 else 1.0
        sd['layers.%d.attn_l' % l] = hash_symmetric(seed, s + 1, (nh, dout, 1), np.sqrt(3.0) * stda)
        sd['layers.%d.attn_r' % l] = hash_symmetric(seed, s + 2, (nh, dout, 1), np.sqrt(3.0) * stda)
        sd['layers.%d.fc1.weight' % l] = hash_symmetric(seed, s + 3, (din, din), np.sqrt(3.0) * std1)
        sd['layers.%d.fc1.bias' % l] = hash_symmetric(seed, s + 4, (din,), 1.0 / np.sqrt(din))
        sd['layers.%d.fc2.weight' % l] = hash_symmetric(seed, s + 5, (nh * dout, din), np.sqrt(3.0) * std2 * g2)
        b2 = hash_symmetric(seed, s + 6, (nh * dout,), 1.0 / np.sqrt(din))
        if l == len(dims) - 1:
            b2 = ((b2.astype(np.float64) + logit_shift) * g2).astype(np.float32)
        sd['layers.%d.fc2.bias' % l] = b2
    return sd


def gat_params(num_feats, hidden=None, heads=None):
    """Contents of ``skeleton_matching.prms`` (reference train_skeleton_matching.py:231-246)
    minus the pickled activation modules (given here by slope / name)."""
    return {'gnn_layers': GAT_LAYERS, 'num_feats': num_feats, 'n_classes': 1,
            'num_hidden': list(GAT_HIDDEN if hidden is None else hidden),
            'heads': list(GAT_HEADS if heads is None else heads),
            'nonlinearity': 0.01, 'final_activation': 'sigmoid',
            'in_drop': 0.0, 'attn_drop': 0.0, 'alpha': GAT_ALPHA, 'residual': False}


def mlp_layer_dims(in_dim, out_dim=54):
    w = [in_dim] + MLP_WIDTHS + [out_dim]
    return [(w[i], w[i + 1]) for i in range(len(w) - 1)]


def mlp_state_dict(seed, in_dim, out_dim=54, out_scale=0.3):
    """Deterministic PoseEstimatorMLP weights, names ``layers.{1,3,..,17}.{weight,bias}``
    (reference utils/mlp.py:8-28).  He-style scale for LeakyReLU(0.1) keeps activations
    O(1) through the 9 layers so that the fp32 parity check is a real one."""
    sd = {}
    dims = mlp_layer_dims(in_dim, out_dim)
    for i, (din, dout) in enumerate(dims):
        key = 2 * i + 1
        last = i == len(dims) - 1
        std = np.sqrt(2.0 / ((1.0 + MLP_SLOPE ** 2) * din))
        if last:
            std = out_scale / np.sqrt(din)
        sd['layers.%d.weight' % key] = hash_symmetric(seed, 50000 + 10 * i, (dout, din), np.sqrt(3.0) * std)
        sd['layers.%d.bias' % key] = hash_symmetric(seed, 50001 + 10 * i, (dout,), 0.05 if not last else 0.2)
    return sd


# --------------------------------------------------------------------------------------
# frames
# --------------------------------------------------------------------------------------

def project_panoptic(X, K, T_d, dist):
    """Project 3xN world points (metres) with the k1,k2,p1,p2,k3 lens model.
    Returns (2xN pixel coords, depth)."""
    xc = T_d[:3, :3] @ X + T_d[:3, 3:4]
    z = xc[2]
    x = xc[0] / z
    y = xc[1] / z
    r = x * x + y * y
    k1, k2, p1, p2, k3 = dist
    radial = 1 + k1 * r + k2 * r * r + k3 * r * r * r
    xd = x * radial + 2 * p1 * x * y + p2 * (r + 2 * x * x)
    yd = y * radial + 2 * p2 * x * y + p1 * (r + 2 * y * y)
    u = K[0, 0] * xd + K[0, 1] * yd + K[0, 2]
    v = K[1, 0] * xd + K[1, 1] * yd + K[1, 2]
    return np.stack([u, v]), z


def ring_transform_manager(params, radius=3.6):
    """root->camera transforms for cameras on a ring around the capture volume, all looking
    at the centre, heights alternating (world: y down, as in the Panoptic calibration)."""
    from .calibration import TransformManager
    tm = TransformManager()
    n = len(params.camera_names)
    for i, name in enumerate(params.camera_names):
        ang = 2.0 * np.pi * i / n
        height = -(1.6 + 0.9 * ((i * 7) % 5) / 4.0)
        C = np.array([radius * np.cos(ang), height, radius * np.sin(ang)])
        target = np.array([0.0, -1.0, 0.0])
        z = target - C
        z /= np.linalg.norm(z)
        x = np.cross(np.array([0.0, 1.0, 0.0]), z)
        x /= np.linalg.norm(x)
        y = np.cross(z, x)
        R = np.stack([x, y, z])              # world -> camera rotation
        T = np.eye(4)
        T[:3, :3] = R
        T[:3, 3] = -R @ C
        tm.add_transform('root', name, T)
    return tm


class FrameSpec:
    """Knobs of the generator; defaults = clean Panoptic-shaped frame."""

    def __init__(self, persons=4, cameras=None, noise_px=0.0, joint_drop=0.0,
                 permute=True, add_id_key=False, spurious=0, empty_cameras=(),
                 float_conf=True, identity_prob=False):
        self.persons = persons
        self.cameras = cameras            # list of camera names present (dict order); None = all
        self.noise_px = noise_px
        self.joint_drop = joint_drop
        self.permute = permute
        self.add_id_key = add_id_key
        self.spurious = spurious
        self.empty_cameras = tuple(empty_cameras)
        self.float_conf = float_conf
        # person p's detections carry prob = 0.3 + 0.1 p in every view: an appearance cue that the
        # hand-built matcher network (matcher_gat_state_dict) reads, standing in for what a trained
        # checkpoint (not available offline) infers from geometry
        self.identity_prob = identity_prob


def _normal(u1, u2):
    return np.sqrt(-2.0 * np.log(np.maximum(u1, 1e-300))) * np.cos(2.0 * np.pi * u2)


def make_frame(calib, frame_index, spec=None, seed=1234):
    """One frame dict in wire format plus its ground truth.

    Returns (frame, gt) with gt = {'persons': [P x J x 3 world metres],
    'owner': {cam: [person id per skeleton in list order]}}.
    """
    spec = spec or FrameSpec()
    params = calib.params
    J = len(params.joint_list)
    W, H = params.image_width, params.image_height
    P = spec.persons
    st = int(frame_index) * 64
    u = hash_uniform(seed, st + 0, P * 3)
    root = np.stack([u[0::3] * 2.4 - 1.2, -(0.8 + 0.4 * u[1::3]), u[2::3] * 2.4 - 1.2], axis=1)   # P x 3
    g = _normal(hash_uniform(seed, st + 1, P * J * 3), hash_uniform(seed, st + 2, P * J * 3)).reshape(P, J, 3)
    pts = root[:, None, :] + 0.25 * g                                                             # P x J x 3
    cams = list(spec.cameras) if spec.cameras is not None else list(params.camera_names)
    frame = {}
    owner = {}
    for ci, cam in enumerate(cams):
        k = calib.index(cam)
        skeletons = []
        own = []
        if cam not in spec.empty_cameras:
            cst = st + 8 + 4 * k
            drop_u = hash_uniform(seed, cst + 0, P * J).reshape(P, J)
            conf_u = hash_uniform(seed, cst + 1, P * J * 2).reshape(P, J, 2)
            nz = _normal(hash_uniform(seed, cst + 2, P * J * 2), hash_uniform(seed, cst + 3, P * J * 2)).reshape(P, J, 2)
            for p in range(P):
                uv, z = project_panoptic(pts[p].T, calib.K32[k].astype(np.float64), calib.T_d[k], calib.dist[k])
                sk = {}
                for j in range(J):
                    x = float(uv[0, j] + spec.noise_px * nz[p, j, 0])
                    y = float(uv[1, j] + spec.noise_px * nz[p, j, 1])
                    if z[j] <= 0.1 or x < 0 or x >= W or y < 0 or y >= H:
                        continue
                    if drop_u[p, j] < spec.joint_drop:
                        continue
                    if spec.float_conf:
                        valid = float(np.float32(0.55 + 0.45 * conf_u[p, j, 0]))
                        prob = float(np.float32(0.30 + 0.70 * conf_u[p, j, 1]))
                    else:
                        valid, prob = 1, 1
                    if spec.identity_prob:
                        prob = float(np.float32(0.3 + 0.1 * p))
                    sk[str(j)] = [j, x, y, valid, prob]
                if spec.add_id_key and sk:
                    sk['ID'] = p
                skeletons.append(sk)
                own.append(p)
            for s in range(spec.spurious):
                su = hash_uniform(seed, cst + 100 + s, J * 2).reshape(J, 2)
                sk = {str(j): [j, float(su[j, 0] * W), float(su[j, 1] * H), 1, 1] for j in range(0, J, 2)}
                skeletons.append(sk)
                own.append(-1 - s)
            if spec.permute and len(skeletons) > 1:
                order = np.argsort(hash_uniform(seed, st + 40 + k, len(skeletons)), kind='stable')
                skeletons = [skeletons[i] for i in order]
                own = [own[i] for i in order]
        # bodies_3D is aligned with the skeleton list, as the converter writes it
        # (get_joints_from_panoptic_model_multi.py:281-287): one GT body per 2D skeleton
        gt_cm = [({('-1' if j == 2 else str(j)): [float(c) * 100.0 for c in pts[o, j]] for j in range(J)} if o >= 0 else {})
                 for o in own]
        frame[cam] = [json.dumps(skeletons), float(frame_index), 'no_image', gt_cm]
        owner[cam] = own
    return frame, {'persons': pts, 'owner': owner}


def make_frames(calib, n, spec=None, seed=1234, start=0):
    frames, gts = [], []
    for i in range(start, start + n):
        f, g = make_frame(calib, i, spec, seed)
        frames.append(f)
        gts.append(g)
    return frames, gts


def decoder_mlp_state_dict(n_cameras, n_joints=18, npj=14, noise_seed=None, noise_bound=0.0, mean_of=range(5, 18)):
    """Hand-built PoseEstimatorMLP weights (reference utils/mlp.py:8-28 shapes and key names) that
    make the network a DECODER of its own input row: output joint j = the triangulated point the
    row carries for joint j in camera block 0 (columns 11..13 = XYZ/10 where the flag in column 10
    is set, reference pose_estimator_dataset_from_json.py:280-285); joint 0, which is never
    triangulated (:75), gets the mean of the joints in `mean_of`.  A value x travels through the
    LeakyReLU(0.1) layers as the pair (x, -x): leaky(x) - leaky(-x) = 1.1 x.  With these weights the
    harness' MPJPE / AP figures are meaningful (millimetres) without a trained checkpoint, which
    does not exist offline.  `noise_bound` > 0 adds dense hash noise to every weight so that the
    GEMMs are not sparse copies."""
    dims = mlp_layer_dims(n_cameras * n_joints * npj, n_joints * 3)
    n_val = n_joints * 3
    sd = {}
    inv = np.float32(1.0 / 1.1)
    for i, (din, dout) in enumerate(dims):
        key = 2 * i + 1
        W = np.zeros((dout, din), np.float32)
        first, last = i == 0, i == len(dims) - 1
        for v in range(n_val):
            j, k = divmod(v, 3)
            if first:
                src = {}
                if j == 0:
                    for jj in mean_of:
                        src[jj * npj + 11 + k] = 1.0 / len(list(mean_of))
                else:
                    src[j * npj + 11 + k] = 1.0
                for col, w in src.items():
                    W[2 * v, col] = w
                    W[2 * v + 1, col] = -w
            elif last:
                W[v, 2 * v] = inv
                W[v, 2 * v + 1] = -inv
            else:
                W[2 * v, 2 * v] = inv
                W[2 * v, 2 * v + 1] = -inv
                W[2 * v + 1, 2 * v] = -inv
                W[2 * v + 1, 2 * v + 1] = inv
        b = np.zeros((dout,), np.float32)
        if noise_bound > 0 and noise_seed is not None:
            W = W + hash_symmetric(noise_seed, 70000 + 10 * i, (dout, din), noise_bound / np.sqrt(din))
            b = b + hash_symmetric(noise_seed, 70001 + 10 * i, (dout,), noise_bound)
        sd['layers.%d.weight' % key] = W.astype(np.float32)
        sd['layers.%d.bias' % key] = b.astype(np.float32)
    return sd


class _NetBuilder:
    """fc1 -> LeakyReLU(alpha) -> fc2 as a list of exact ReLU units: relu(y) is assembled from the
    pair (leaky(y), leaky(-y)) since leaky(y) = alpha*y + (1-alpha)*relu(y)."""

    def __init__(self, din, dout, alpha):
        self.W1 = np.zeros((din, din), np.float64)
        self.b1 = np.zeros(din, np.float64)
        self.W2 = np.zeros((dout, din), np.float64)
        self.b2 = np.zeros(dout, np.float64)
        self.alpha = alpha
        self.n = 0

    def relu(self, weights, bias):
        """New unit relu(sum_c weights[c]*x[c] + bias) -> handle (pair of hidden rows)."""
        u = self.n
        self.n += 2
        for c, w in weights.items():
            self.W1[u, c] += w
            self.W1[u + 1, c] -= w
        self.b1[u] += bias
        self.b1[u + 1] -= bias
        return u

    def out(self, channel, unit, w):
        """output[channel] += w * relu-unit."""
        a = self.alpha
        # relu(y) = (leaky(y) - a*y)/(1-a), y = (leaky(y) - leaky(-y))/(1+a)
        k = a / (1.0 + a)
        self.W2[channel, unit] += w * (1.0 - k) / (1.0 - a)
        self.W2[channel, unit + 1] += w * k / (1.0 - a)


def matcher_gat_state_dict(num_feats, n_cameras, n_joints=18, levels=8, key_joint=8, gain=12.0, noise_seed=None,
                           noise_bound=0.0):
    """Hand-built GAT2 weights (reference gat2.py shapes and state-dict names, deployed layer
    sizes) that make the network a correct matcher on frames generated with
    FrameSpec(identity_prob=True): an edge-node scores ~0.88 when its two skeletons carry the same
    `prob` level on `key_joint` and ~0.12 otherwise.

    How: attn_l = attn_r = 0, so every edge softmax is uniform and a layer's graph half is the mean
    over in-edges.  Layer 0 turns the key joint's prob of a head row into a one-hot over `levels`
    (piecewise-linear hats) plus a node-type flag read from column 0; the edge-node rows (one-hot at
    column 1) produce zeros.  After the mean an edge-node holds (onehot(h1) + onehot(h2))/3, so a
    component reaches 2/3 exactly when the two heads agree; layer 1 thresholds that at 1/2.  Layers
    2..4 carry the decision, re-deriving in every layer which rows are edge-nodes from the type
    channel (its value after the mean is a known constant at edge-nodes and lies on one side of it
    at heads) so that head rows contribute zeros to the next mean.  The last layer emits
    gain*(decision - 1/2) at edge-nodes and 0 at heads.
    With noise_bound > 0 dense hash noise is added to every weight (thresholds have margins of 1/6)."""
    dims = gat_layer_dims(num_feats)
    alpha = GAT_ALPHA
    sd = {}
    blk = n_joints * 10
    third = float(np.float32(1.0 / 3.0))
    two_thirds = float(np.float32(third) + np.float32(third))
    for l, (din, nh, dout) in enumerate(dims):
        nb = _NetBuilder(din, nh * dout, alpha)
        if l == 0:
            s_cols = {2 + c * blk + key_joint * 10 + 3: 1.0 for c in range(n_cameras)}    # prob of key_joint, any camera block
            w = 0.05
            for k in range(levels):
                ck = float(np.float32(0.3 + 0.1 * k))
                for t, coef in ((ck - w, 1.0 / w), (ck, -2.0 / w), (ck + w, 1.0 / w)):
                    col = dict(s_cols)
                    col[0] = -t                         # threshold through the head flag: edge-node rows give 0
                    nb.out(k, nb.relu(col, 0.0), coef)
            nb.out(levels, nb.relu({0: 1.0}, 0.0), 1.0)  # type flag F: 1 at heads, 0 at edge-nodes
        elif l == 1:
            c = 12.0
            for k in range(levels):
                u = nb.relu({k: 1.0, levels: c}, -0.5 - c * two_thirds)
                nb.out(0, u, 6.0)                        # decision v in {0, 1}
            nb.out(1, nb.relu({levels: c}, 1.0 - c * two_thirds), 1.0)     # type t: 1 at edge-nodes
        else:
            c = 24.0
            a = nb.relu({0: 3.0, 1: -c}, c * third)      # v at edge-nodes (type channel == 1/3), 0 at heads
            t = nb.relu({1: -c}, 1.0 + c * third)        # 1 at edge-nodes, 0 at heads
            if l < len(dims) - 1:
                nb.out(0, a, 1.0)
                nb.out(1, t, 1.0)
            else:
                nb.out(0, a, 3.0 * gain)
                nb.out(0, t, -1.5 * gain)
        W1, b1, W2, b2 = nb.W1, nb.b1, nb.W2, nb.b2
        if noise_bound > 0 and noise_seed is not None:
            W1 = W1 + hash_symmetric(noise_seed, 90000 + 10 * l, W1.shape, noise_bound / np.sqrt(din))
            W2 = W2 + hash_symmetric(noise_seed, 90001 + 10 * l, W2.shape, noise_bound / np.sqrt(din))
        sd['layers.%d.attn_l' % l] = np.zeros((nh, dout, 1), np.float32)
        sd['layers.%d.attn_r' % l] = np.zeros((nh, dout, 1), np.float32)
        sd['layers.%d.fc1.weight' % l] = W1.astype(np.float32)
        sd['layers.%d.fc1.bias' % l] = b1.astype(np.float32)
        sd['layers.%d.fc2.weight' % l] = W2.astype(np.float32)
        sd['layers.%d.fc2.bias' % l] = b2.astype(np.float32)
    return sd
