"""TEST INFRASTRUCTURE — CPU restatement (numpy / torch-CPU) of the reference's per-frame
inference path.  NOT product code: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product (3d_multi_pose_estimator_amd/)
never does and fails loudly when its HIP library is missing.

Pinning: every function here is checked by tests/test_oracle_golden.py against the
fixtures in tests/golden/, which oracle/gen_golden.py produced by running the
reference's own files (/root/reference) in the build container.  Third-party
arithmetic that is absent from the image (DGL edge_softmax / message passing, OpenCV
undistortPoints / triangulatePoints, pytransform3d) is restated from the published
algorithms (oracle/shims/), so parity with the real DGL / OpenCV builds is unpinned at
ULP level; everything the reference's own code does is pinned.

Each function cites the reference lines it follows (paths relative to /root/reference).
"""
import itertools
import json

import numpy as np
import torch

torch.set_grad_enabled(False)

J_FEATS = 10          # numbers per (camera, joint) in the graph node row (graph_generator.py:128-140)


# --------------------------------------------------------------------------------------
# frame parsing (graph_generator.py:573-605)
# --------------------------------------------------------------------------------------

def parse_frame(frame, params):
    """Heads of one frame in reference order.

    `frame` = {cam: [json-string of skeleton list, ...]} (insertion order matters).
    Head ids follow the dict order of the cameras, then list order, skipping skeletons
    without any joint key other than "ID" (graph_generator.py:583-601).
    Returns dict(heads=[(cam, skeleton_index, skeleton_dict)], slots=[(cam, [head ids])]).
    """
    heads = []
    slots = []
    for cam in frame:
        if cam not in params.used_cameras_skeleton_matching:
            continue
        ids = []
        for idx, sk in enumerate(json.loads(frame[cam][0])):
            if sum(1 for k in sk if k != 'ID') == 0:
                continue
            ids.append(len(heads))
            heads.append((cam, idx, sk))
        slots.append((cam, ids))
    return {'heads': heads, 'slots': slots}


def processed_input(frame):
    """The dict the callers hand to MergedMultipleHumansDataset: cameras with an empty
    skeleton list are dropped and the list is re-serialised (metrics_from_model.py:182-191)."""
    out = {}
    for cam in frame:
        data = json.loads(frame[cam][0])
        if data:
            out[cam] = [json.dumps(data), frame[cam][1]]
    return out


# --------------------------------------------------------------------------------------
# graph construction (graph_generator.py:444-508, 607-656, 813-876)
# --------------------------------------------------------------------------------------

def head_row(sk, cam, calib):
    """One 1 x F node row (HumanGraphFromView.initializeWithAlternative3, :444-508)."""
    params = calib.params
    J = len(params.joint_list)
    sm = list(params.used_cameras_skeleton_matching)
    F = 2 + len(sm) * J * J_FEATS
    c_sm = sm.index(cam)
    c = calib.index(cam)
    W, H = params.image_width, params.image_height
    row = torch.zeros([1, F])
    row[0, 0] = 1.0
    keys = [k for k in sk if k != 'ID']
    if not keys:
        return row, 0
    pl = torch.tensor([[sk[k][1], sk[k][2], 1.0] for k in keys]).type(torch.float32)
    zeros = torch.tensor([[0.0] * pl.shape[0]])
    pix = torch.matmul(torch.from_numpy(calib.Kinv32[c]), pl.transpose(dim0=1, dim1=0))
    ray = torch.matmul(torch.from_numpy(calib.T_i32[c]), torch.cat((pix, zeros))).transpose(dim0=1, dim1=0)
    centre = torch.from_numpy(calib.centre32[c])
    for i, k in enumerate(keys):
        v = sk[k]
        base = 2 + (c_sm * J + int(k)) * J_FEATS
        row[0, base + 0] = (v[1] - W / 2) / (W / 2)
        row[0, base + 1] = (H / 2 - v[2]) / (H / 2)
        row[0, base + 2] = v[3]
        row[0, base + 3] = v[4]
        row[0, base + 4] = centre[0]
        row[0, base + 5] = centre[1]
        row[0, base + 6] = centre[2]
        row[0, base + 7] = ray[i][0]
        row[0, base + 8] = ray[i][1]
        row[0, base + 9] = ray[i][2]
    return row, len(keys)


def topology(slots):
    """Edge list of alternative '3' from per-camera head ids (process_test :854-864,
    add_edge_node_to_graph :627-656): H self loops, then per edge-node X=(h1,h2):
    (h1,X),(X,h1),(h2,X),(X,h2),(X,X).  Returns (N, src, dst, pairs[M,2])."""
    H = sum(len(ids) for _, ids in slots)
    src = list(range(H))
    dst = list(range(H))
    pairs = []
    X = H
    for a in range(len(slots)):
        for b in range(a + 1, len(slots)):
            if slots[a][0] == slots[b][0]:
                continue
            for h1 in slots[a][1]:
                for h2 in slots[b][1]:
                    src += [h1, X, h2, X, X]
                    dst += [X, h1, X, h2, X]
                    pairs.append((h1, h2))
                    X += 1
    return X, np.array(src, np.int32), np.array(dst, np.int32), np.array(pairs, np.int32).reshape(-1, 2)


def build_graph(frame, calib):
    """Frame -> graph of MergedMultipleHumansDataset(mode='test', alt='3').  Returns None
    when no edge-node exists (the reference then produces no graph, :866)."""
    params = calib.params
    pf = parse_frame(frame, params)
    heads, slots = pf['heads'], pf['slots']
    N, src, dst, pairs = topology(slots)
    H = len(heads)
    if N == H:
        return None
    F = 2 + len(params.used_cameras_skeleton_matching) * len(params.joint_list) * J_FEATS
    feats = torch.zeros([N, F])
    for h, (cam, _, sk) in enumerate(heads):
        feats[h] = head_row(sk, cam, calib)[0][0]
    feats[H:, 1] = 1.0
    return {'N': N, 'H': H, 'src': src, 'dst': dst, 'pairs': pairs, 'feats': feats,
            'edge_nodes_indices': np.arange(H, N, dtype=np.int64),
            'nodes_camera': [h[0] for h in heads] + [''] * (N - H),
            'jsons_for_head': {i: h[2] for i, h in enumerate(heads)},
            'skeleton_index': {i: h[1] for i, h in enumerate(heads)},
            'slots': slots}


# --------------------------------------------------------------------------------------
# GAT forward (gat2.py:50-88, 137-149)
# --------------------------------------------------------------------------------------

def _t(x):
    return x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))


def gat_layer(sd, l, h, src, dst, alpha, num_heads):
    """GraphAttention2.forward (gat2.py:50-76): fc1, LeakyReLU(alpha), fc2, a1/a2 by bmm,
    e = LeakyReLU(alpha)(a1[src]+a2[dst]), softmax over incoming edges of dst
    (DGL edge_softmax: max, exp, sum, div), out[v] = sum_e score_e * ft2[src_e].
    Runs in the dtype of `h` (float32 = the reference; float64 = the same network without fp32
    rounding, used by tests to size the reference's own rounding noise)."""
    p = 'layers.%d.' % l
    N = h.shape[0]
    dt = h.dtype

    def w(name):
        return _t(sd[p + name]).to(dt)
    ft1 = torch.nn.functional.linear(h, w('fc1.weight'), w('fc1.bias'))
    h2 = torch.nn.functional.leaky_relu(ft1, alpha)
    ft2 = torch.nn.functional.linear(h2, w('fc2.weight'), w('fc2.bias')).reshape((N, num_heads, -1))
    head_ft = ft2.transpose(0, 1)
    a1 = torch.bmm(head_ft, w('attn_l')).transpose(0, 1)
    a2 = torch.bmm(head_ft, w('attn_r')).transpose(0, 1)
    s = torch.from_numpy(np.asarray(src)).long()
    d = torch.from_numpy(np.asarray(dst)).long()
    e = torch.nn.functional.leaky_relu(a1[s] + a2[d], alpha)                 # E x H x 1
    idx = d.view(-1, 1, 1).expand_as(e)
    mx = torch.full((N, num_heads, 1), float('-inf'), dtype=dt).scatter_reduce(0, idx, e, reduce='amax', include_self=True)
    score = torch.exp(e - mx[d])
    ssum = torch.zeros((N, num_heads, 1), dtype=dt).index_add_(0, d, score)
    a = score / ssum[d]
    out = torch.zeros_like(ft2).index_add_(0, d, ft2[s] * a)
    return out, {'ft2': ft2, 'a1': a1, 'a2': a2}


def gat_forward(sd, prm, feats, src, dst, keep=False, dtype=torch.float32):
    """GAT2.forward (gat2.py:137-149) with activation LeakyReLU(0.01) and Sigmoid
    (train_skeleton_matching.py:54, :34).  Returns N scores (caller squeezes N x 1 x 1)."""
    heads = list(prm['heads']) + [1]
    L = prm['gnn_layers']
    h = _t(feats).to(dtype)
    inter = []
    for l in range(L - 1):
        h, _ = gat_layer(sd, l, h, src, dst, prm['alpha'], heads[l])
        h = torch.nn.functional.leaky_relu(h.flatten(1), prm.get('nonlinearity', 0.01))
        if keep:
            inter.append(h.clone())
    h, _ = gat_layer(sd, L - 1, h, src, dst, prm['alpha'], 1)
    out = torch.sigmoid(h).reshape(-1)
    return (out, inter) if keep else out


# --------------------------------------------------------------------------------------
# greedy clustering (skeleton_matching_utils.py:12-132)
# --------------------------------------------------------------------------------------

class PySet:
    """CPython 3.10 `set` of small non-negative ints (hash(i) == i): open addressing with
    LINEAR_PROBES=9, PERTURB_SHIFT=5, resize to used*4 when fill*5 >= mask*3
    (Objects/setobject.c set_add_entry / set_insert_clean / set_table_resize).
    Only what the reference needs: add, membership, iteration in slot order."""

    def __init__(self, items=()):
        self.mask = 7
        self.table = [-1] * 8
        self.fill = 0
        for i in items:
            self.add(i)

    def _probe(self, key, table, mask, stop_on_equal):
        perturb = key
        i = key & mask
        while True:
            probes = 9 if i + 9 <= mask else 0
            j = i
            while True:
                if table[j] == -1:
                    return j, False
                if stop_on_equal and table[j] == key:
                    return j, True
                j += 1
                if probes == 0:
                    break
                probes -= 1
            perturb >>= 5
            i = (i * 5 + 1 + perturb) & mask

    def add(self, key):
        j, found = self._probe(key, self.table, self.mask, True)
        if found:
            return
        self.table[j] = key
        self.fill += 1
        if self.fill * 5 < self.mask * 3:
            return
        minused = self.fill * 4
        newsize = 8
        while newsize <= minused:
            newsize <<= 1
        new = [-1] * newsize
        for k in self.table:
            if k != -1:
                jj, _ = self._probe(k, new, newsize - 1, False)
                new[jj] = k
        self.table = new
        self.mask = newsize - 1

    def __contains__(self, key):
        return self._probe(key, self.table, self.mask, True)[1]

    def __iter__(self):
        return (k for k in self.table if k != -1)

    def __len__(self):
        return self.fill


def cluster(scores, pairs, H, head_cam, n_cams, min_views=2, thr=0.5):
    """get_person_proposal_from_network_output restated on the implicit topology.

    scores[M] f32 (edge-node X = H + m), pairs[M,2] = (h1,h2), head_cam[H] = camera index
    (position in used_cameras_skeleton_matching).  Returns a list of persons, each a list
    of n_cams head ids (-1 = None), in the reference's output order.

    Follows: edge scan / matching creation :33-55 (a,b = list(set) order), stable sort by
    score descending :60, per-matching constraints and human-index bookkeeping :61-108
    (including the quirk that a merge deletes the absorbed group's camera list without
    merging it, :97-102), connected components in node-insertion order with networkx
    3.4.2's _plain_bfs and the component set's iteration order :117-130.
    """
    M = len(pairs)
    order_nodes = []            # G node insertion order
    seen_node = set()
    matchings = []              # (a, b, score, creation index)
    for m in range(M):
        h1, h2 = int(pairs[m][0]), int(pairs[m][1])
        for other in (h1, h2):
            if other not in seen_node:
                seen_node.add(other)
                order_nodes.append(other)
        sc = float(scores[m])
        if sc > thr:
            a, b = list(PySet([h1, h2]))
            matchings.append((a, b, sc, len(matchings)))
    matchings.sort(key=lambda t: -t[2])      # stable: ties keep creation order
    linked = {h: [head_cam[h]] for h in order_nodes}
    human = {}
    cams_for_human = {}
    cur = 0
    adj = {h: [] for h in order_nodes}
    for a, b, _, _ in matchings:
        ca, cb = head_cam[a], head_cam[b]
        if ca in linked[b] or cb in linked[a]:
            continue
        if a in human and cb in cams_for_human[human[a]]:
            continue
        if b in human and ca in cams_for_human[human[b]]:
            continue
        if a not in human and b not in human:
            human[a] = cur
            human[b] = cur
            cams_for_human[cur] = [ca, cb]
            cur += 1
        elif a in human and b not in human:
            human[b] = human[a]
            cams_for_human[human[a]].append(cb)
        elif b in human and a not in human:
            human[a] = human[b]
            cams_for_human[human[b]].append(ca)
        else:
            if any(c in cams_for_human[human[a]] for c in cams_for_human[human[b]]):
                continue
            new_i, old_i = human[a], human[b]
            for n in list(human.keys()):
                if human[n] == old_i:
                    human[n] = new_i
            del cams_for_human[old_i]
        if b not in adj[a]:
            adj[a].append(b)
            adj[b].append(a)
        linked[a].append(cb)
        linked[b].append(ca)
    out = []
    done = set()
    n_nodes = len(order_nodes)
    for v in order_nodes:
        if v in done:
            continue
        comp = PySet([v])
        level = [v]
        full = False
        while level and not full:
            nxt = []
            for x in level:
                for w in adj[x]:
                    if w not in comp:
                        comp.add(w)
                        nxt.append(w)
                if len(comp) == n_nodes:
                    full = True
                    break
            level = nxt
        members = list(comp)
        done.update(members)
        if len(members) < min_views:
            continue
        person = [-1] * n_cams
        for h in members:
            person[head_cam[h]] = h
        out.append(person)
    return out


# --------------------------------------------------------------------------------------
# OpenCV restatements (f64)
# --------------------------------------------------------------------------------------

def undistort_points(pts, K32, dist):
    """cv2.undistortPoints(pts, K, dist) without R/P: normalise, 5 fixed-point iterations
    of the k1,k2,p1,p2,k3 model.  pts (n,2) f64 -> (n,2) f64."""
    pts = np.asarray(pts, np.float64).reshape(-1, 2)
    K = np.asarray(K32, np.float64)
    k1, k2, p1, p2, k3 = [float(v) for v in dist]
    ifx, ify = 1.0 / K[0, 0], 1.0 / K[1, 1]
    x = (pts[:, 0] - K[0, 2]) * ifx
    y = (pts[:, 1] - K[1, 2]) * ify
    x0, y0 = x.copy(), y.copy()
    alive = np.ones(x.shape[0], bool)
    for _ in range(5):
        r2 = x * x + y * y
        icdist = (1 + ((0.0 * r2 + 0.0) * r2 + 0.0) * r2) / (1 + ((k3 * r2 + k2) * r2 + k1) * r2)
        neg = alive & (icdist < 0)
        dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x) + 0.0 * r2 + 0.0 * r2 * r2
        dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y + 0.0 * r2 + 0.0 * r2 * r2
        xn = (x0 - dx) * icdist
        yn = (y0 - dy) * icdist
        upd = alive & ~neg
        x = np.where(upd, xn, np.where(neg, x0, x))
        y = np.where(upd, yn, np.where(neg, y0, y))
        alive = alive & ~neg
    return np.stack([x, y], axis=1)


def dlt_pair(P1, P2, x1, x2):
    """cv2.triangulatePoints for one point + dehomogenisation: rows x*P[2]-P[0],
    y*P[2]-P[1] per view, right singular vector of the smallest singular value."""
    A = np.stack([x1[0] * P1[2] - P1[0], x1[1] * P1[2] - P1[1],
                  x2[0] * P2[2] - P2[0], x2[1] * P2[2] - P2[1]])
    v = np.linalg.svd(A)[2][3]
    return v[0:3] / v[3]


# --------------------------------------------------------------------------------------
# 3D stage A: MLP input row (pose_estimator_dataset_from_json.py:49-101, 237-298)
# --------------------------------------------------------------------------------------

def person_skeletons(person, jsons_for_head, cam_names):
    """{cam: skeleton} in camera order for one clustering result
    (metrics_from_model.py:246-252)."""
    return {cam: jsons_for_head[person[i]] for i, cam in enumerate(cam_names) if person[i] is not None and person[i] >= 0}


def triangulate_mean(skels, calib):
    """get_3D_from_triangulation (:63-101): joints with values[0] > 0 seen by >= 2 cameras,
    mean over camera pairs (insertion order) of the pairwise DLT point.  {joint str: (3,) f64}."""
    pts = {}
    for cam, sk in skels.items():
        for j, pos in sk.items():
            if j == 'ID':
                continue
            if pos[0] > 0.:
                pts.setdefault(j, {})[cam] = np.array([pos[1], pos[2]])
    out = {}
    for ji in calib.params.joint_list:
        j = str(ji)
        if j in pts and len(pts[j]) > 1:
            cams = list(pts[j].keys())
            acc = np.zeros(3)
            n = 0
            for c1, c2 in itertools.combinations(range(len(cams)), 2):
                k1, k2 = calib.index(cams[c1]), calib.index(cams[c2])
                u1 = undistort_points(pts[j][cams[c1]], calib.K32[k1], calib.dist[k1])[0]
                u2 = undistort_points(pts[j][cams[c2]], calib.K32[k2], calib.dist[k2])[0]
                acc += dlt_pair(calib.P[k1], calib.P[k2], u1, u2)
                n += 1
            out[j] = acc / n
    return out


def mlp_input_row(skels, calib):
    """PoseEstimatorDataset dict branch (:237-298): the 1 x (V*J*14) f32 row of one person.
    Returns (row f32, kept) with kept = sum(|row|) > 1 (:287)."""
    params = calib.params
    J = len(params.joint_list)
    npj = params.numbers_per_joint
    W, H = params.image_width, params.image_height
    used = list(params.used_cameras)
    out = torch.zeros([len(used) * J * npj])
    res3d = triangulate_mean(skels, calib)
    for cam, sk in skels.items():
        if cam not in used:
            continue
        c = calib.index(cam)
        off = used.index(cam) * J * npj
        ti = torch.from_numpy(calib.T_i32[c])
        centre = torch.matmul(ti, torch.tensor([0.0, 0.0, 0.0, 1.0])) / 10.
        keys = [k for k in sk if k != 'ID']
        if not keys:
            continue
        pl = np.array([[sk[k][1], sk[k][2]] for k in keys])
        nf = np.array([[W / 2, H / 2]] * pl.shape[0])
        norm = (pl - nf) / nf
        und = torch.from_numpy(undistort_points(pl, calib.K32[c], calib.dist[c])).type(torch.float32)
        newcol = torch.tensor([[1.0, 0.0]] * pl.shape[0])
        ray = (torch.matmul(ti, torch.cat((und, newcol), dim=1).transpose(dim0=1, dim1=0)) / 10.).transpose(dim0=1, dim1=0)
        for i, k in enumerate(keys):
            jo = off + int(k) * npj
            out[jo] = sk[k][3]
            out[jo + 1] = norm[i][0]
            out[jo + 2] = norm[i][1]
            out[jo + 3] = sk[k][4]
            out[jo + 4: jo + 7] = centre[0:3]
            out[jo + 7: jo + 10] = ray[i][0:3]
    for ci in range(len(used)):
        off = ci * J * npj
        for j, p in res3d.items():
            jo = off + int(j) * npj
            out[jo + 10] = 1.
            out[jo + 11: jo + 14] = torch.tensor(p) / 10.
    return out, bool(torch.sum(torch.abs(out)) > 1)


def mlp_forward(sd, x, slope=0.1):
    """PoseEstimatorMLP.forward (utils/mlp.py:8-31): Linear+LeakyReLU(0.1) x8, Linear."""
    h = _t(x).float()
    keys = sorted({int(k.split('.')[1]) for k in sd})
    for n, k in enumerate(keys):
        h = torch.nn.functional.linear(h, _t(sd['layers.%d.weight' % k]), _t(sd['layers.%d.bias' % k]))
        if n != len(keys) - 1:
            h = torch.nn.functional.leaky_relu(h, slope)
    return h


def mlp_exact(sd, x, slope=0.1):
    """The same network evaluated in f64 with the activations rounded to fp32 between layers:
    what every fp32 implementation of utils/mlp.py approximates.  Tests use it to split
    |gpu - reference| into the two sides' own rounding noise (no reference counterpart)."""
    h = _t(x).double()
    keys = sorted({int(k.split('.')[1]) for k in sd})
    for n, k in enumerate(keys):
        h = h @ _t(sd['layers.%d.weight' % k]).double().T + _t(sd['layers.%d.bias' % k]).double()
        if n != len(keys) - 1:
            h = torch.nn.functional.leaky_relu(h, slope)
        h = h.float().double()
    return h


def decode_pose(out_row, n_joints):
    """results*10, joint k = (r[3k], r[3k+1], r[3k+2]) (metrics_from_model.py:281-294)."""
    r = (_t(out_row) * 10.).numpy()
    return r.reshape(n_joints, 3)


# --------------------------------------------------------------------------------------
# 3D stage B: triangulation with median filter (pose_estimator_utils.py:52-75,
# metrics_from_triangulation.py:234-272)
# --------------------------------------------------------------------------------------

def triangulate_person(skels, calib, positive_ids_only=False, all_joints=False):
    """All joint keys of the person's skeletons (no "ID"/joint-0 filter, cameras in
    parameters.cameras order), pairwise DLT, keep pairs within 5 cm of the upper median
    along axes_3D['Y'][0], mean.  Returns {joint_idx: (3,) f64}; joints outside
    used_joints come back as zeros (caller copy, :259-272).
    positive_ids_only / all_joints: the variant of test/reprojection_error.py:289-302, whose
    gather keeps joints with values[0] > 0 only and which uses `triangulate`'s result as it is."""
    params = calib.params
    axis = params.axes_3D['Y'][0]
    pts = {}
    for cam in params.camera_names:
        if cam in skels:
            for j, pos in skels[cam].items():
                if positive_ids_only and not (j != 'ID' and pos[0] > 0.):
                    continue
                pts.setdefault(j, {})[cam] = np.array([pos[1], pos[2]])
    res = {}
    for ji in params.joint_list:
        j = str(ji)
        if j in pts and len(pts[j]) > 1:
            cams = list(pts[j].keys())
            lst = []
            for c1, c2 in itertools.combinations(range(len(cams)), 2):
                k1, k2 = calib.index(cams[c1]), calib.index(cams[c2])
                u1 = undistort_points(pts[j][cams[c1]], calib.K32[k1], calib.dist[k1])[0]
                u2 = undistort_points(pts[j][cams[c2]], calib.K32[k2], calib.dist[k2])[0]
                lst.append(dlt_pair(calib.P[k1], calib.P[k2], u1, u2))
            lst = np.array(lst)
            d = lst[:, axis]
            med = np.sort(d)[d.shape[0] // 2]
            keep = np.abs(d - med) < 0.05
            res[ji] = np.mean(lst[keep], axis=0)
    if all_joints:
        return res
    out = {}
    for ji in params.joint_list:
        if ji in res:
            out[ji] = res[ji] if ji in params.used_joints else np.zeros(3)
    return out


# --------------------------------------------------------------------------------------
# whole frame
# --------------------------------------------------------------------------------------

def run_frame(frame, calib, gat_sd, gat_prm, mlp_sd=None, mode='mlp'):
    """One iteration of the callers' per-frame loop (metrics_from_model.py:178-294 /
    metrics_from_triangulation.py:187-272) on the already `processed_input` frame."""
    params = calib.params
    g = build_graph(frame, calib)
    if g is None:
        return None
    scores_all = gat_forward(gat_sd, gat_prm, g['feats'], g['src'], g['dst'])
    sm = list(params.used_cameras_skeleton_matching)
    head_cam = [sm.index(c) for c in g['nodes_camera'][:g['H']]]
    scores = scores_all[g['H']:].numpy()
    persons = cluster(scores, g['pairs'], g['H'], head_cam, len(sm), params.min_number_of_views)
    res = {'graph': g, 'scores': scores, 'persons': persons}
    if mode == 'mlp':
        rows = []
        for p in persons:
            row, kept = mlp_input_row(person_skeletons(p, g['jsons_for_head'], sm), calib)
            if kept:
                rows.append(row)
        res['mlp_in'] = torch.stack(rows) if rows else torch.zeros((0, 0))
        if rows:
            out = mlp_forward(mlp_sd, res['mlp_in'])
            res['poses'] = np.stack([decode_pose(out[i], len(params.joint_list)) for i in range(out.shape[0])])
        else:
            res['poses'] = np.zeros((0, len(params.joint_list), 3), np.float32)
    elif mode == 'tri':
        res['tri'] = [triangulate_person(person_skeletons(p, g['jsons_for_head'], sm), calib) for p in persons]
    return res


# --------------------------------------------------------------------------------------
# scenes composed from single-person files: mode='test_generated'
# (graph_generator.py:526-536, 672-810; consumer test/sm_metrics_without_gt.py:95-170)
# --------------------------------------------------------------------------------------

def generated_dataset_scenes(files_data, probabilities, limit, rng=None):
    """The sampling of MergedMultipleHumansDataset(list_of_files, probabilities, limit, mode='test_generated').

    Constructor (graph_generator.py:526-536): per file the frame indices 0..n-1, shuffled with `random.shuffle` (the
    module-level generator: `rng` = the `random` module, seeded by the caller).  process_training's sample_and_remove
    (:674-693): up to `limit` times draw num_people = randint(1, n_files), take the num_people files with the largest
    probabilities (np.argpartition(p, -k)[-k:], in that order) and pop the LAST shuffled index of each; the first file that
    has run dry ends the sampling for good (`return`, :688-689).  Yields lists of single-person frames."""
    import random as _random
    rng = rng or _random
    indices = []
    for data in files_data:
        idx = list(range(len(data)))
        rng.shuffle(idx)
        indices.append(idx)
    scenes = []
    for _ in range(limit):
        if all(len(d) == 0 for d in files_data):
            break
        k = rng.randint(1, len(files_data))
        chosen = np.argpartition(np.array(probabilities), -k)[-k:]
        views = []
        for f in chosen:
            if not indices[f]:
                return scenes
            views.append(files_data[f][indices[f].pop()])
        if views:
            scenes.append(views)
    return scenes


def generated_graph(views, calib):
    """One graph of process_training (graph_generator.py:697-810) from the sampled single-person frames `views`.

    Nodes: per view (person) its heads in camera-dict order and list order (load_people_view_graph, :573-605), ids offset
    by the heads of the views before it (:731).  Per view and camera the head with the most joints is the person's, the
    rest are spurious (:718-726).  Edge-nodes (each with the five edges of add_edge_node_to_graph, :627-656) for ORDERED
    head pairs of different cameras, in this order: per person -- own x own (label 1, :749-760), own x every other
    person's (0, :762-774), own x spurious (0, :776-786) -- then spurious x spurious (0, :788-797).  Returns None when no
    edge-node exists (:799)."""
    params = calib.params
    heads_all, people, spurious, total = [], [], [], 0
    for view in views:
        pf = parse_frame(view, params)
        person = []
        for cam, ids in pf['slots']:
            if not ids:
                continue
            nj = [sum(1 for k in pf['heads'][h][2] if k != 'ID') for h in ids]
            best = 0
            for i in range(1, len(nj)):          # max(enumerate(...), key=joints): the first of the largest
                if nj[i] > nj[best]:
                    best = i
            for i, h in enumerate(ids):
                if i == best:
                    person.append((h + total, cam))
                else:
                    spurious.append((h + total, cam))
        people.append(person)
        heads_all += pf['heads']
        total += len(pf['heads'])
    H = total
    src, dst = list(range(H)), list(range(H))
    pairs, labels = [], []

    def edge_nodes(a, b, label):
        for h1, c1 in a:
            for h2, c2 in b:
                if c1 == c2:
                    continue
                X = H + len(pairs)
                src.extend([h1, X, h2, X, X])
                dst.extend([X, h1, X, h2, X])
                pairs.append((h1, h2))
                labels.append(label)
    for ip, person in enumerate(people):
        edge_nodes(person, person, 1.0)
        for io, other in enumerate(people):
            if io != ip:
                edge_nodes(person, other, 0.0)
        edge_nodes(person, spurious, 0.0)
    edge_nodes(spurious, spurious, 0.0)
    if not pairs:
        return None
    N = H + len(pairs)
    F = 2 + len(params.used_cameras_skeleton_matching) * len(params.joint_list) * J_FEATS
    feats = torch.zeros([N, F])
    for h, (cam, _, sk) in enumerate(heads_all):
        feats[h] = head_row(sk, cam, calib)[0][0]
    feats[H:, 1] = 1.0
    return {'N': N, 'H': H, 'src': np.array(src, np.int32), 'dst': np.array(dst, np.int32),
            'pairs': np.array(pairs, np.int32).reshape(-1, 2), 'feats': feats, 'labels': np.array(labels, np.float64),
            'edge_nodes_indices': np.arange(H, N, dtype=np.int64),
            'nodes_camera': [h[0] for h in heads_all] + [''] * (N - H)}


def proposal_labels(persons, H):
    """One label per head: index of the first proposal that holds it, or the number of proposals
    (test/sm_metrics_without_gt.py:133-141, 149-157)."""
    out = []
    for h in range(H):
        idx = len(persons)
        for p, members in enumerate(persons):
            if h in members:
                idx = p
                break
        out.append(idx)
    return out


def sm_without_gt(graphs, score_fn, params, thr=0.5):
    """The loop of test/sm_metrics_without_gt.py:112-170 over `graphs` (generated_graph dicts): proposals from the model's
    scores (score_fn(graph) -> N scores) and from the labels used as scores (:143-147), adjusted Rand index / homogeneity /
    completeness / V-measure averaged over the graphs.  Returns the four means and the per-graph proposals."""
    from sklearn.metrics import adjusted_rand_score, homogeneity_completeness_v_measure
    sm = list(params.used_cameras_skeleton_matching)
    tot = {'rand score': 0.0, 'homogeneity': 0.0, 'completeness': 0.0, 'v_measure': 0.0}
    per_graph = []
    for g in graphs:
        H = g['H']
        head_cam = [sm.index(c) for c in g['nodes_camera'][:H]]
        scores = np.asarray(score_fn(g), np.float32).reshape(-1)[H:]
        est = cluster(scores, g['pairs'], H, head_cam, len(sm), params.min_number_of_views, thr)
        gt = cluster(g['labels'].astype(np.float32), g['pairs'], H, head_cam, len(sm), params.min_number_of_views, thr)
        l_est, l_gt = proposal_labels(est, H), proposal_labels(gt, H)
        tot['rand score'] += adjusted_rand_score(l_gt, l_est)
        hom, com, v = homogeneity_completeness_v_measure(l_gt, l_est)
        tot['homogeneity'] += hom
        tot['completeness'] += com
        tot['v_measure'] += v
        per_graph.append({'est': est, 'gt': gt})
    n = max(1, len(graphs))
    return {k: v / n for k, v in tot.items()}, per_graph
