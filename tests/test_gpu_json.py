"""GPU: the device-side second-level parse of the frame JSON (csrc/jsonparse.hip) against the
host packer (csrc/packer.cpp), which is itself pinned to the Python packer and, through it, to the reference's
load_people_view_graph order (tests/test_host_logic.py).  Bar: the nine arrays of mpe_batch and skeleton_index
bit for bit, on every document the device accepts; documents it does not accept must come back as "host parser,
please" (None), never as different arrays."""
import json
import random
import struct

import numpy as np
import pytest
import torch

from conftest import ALL_CASES, env, load_case, oracle, pkg

pytestmark = pytest.mark.gpu

FIELDS = ('frame_head_off', 'frame_en_off', 'slot_cam', 'slot_n', 'head_cam', 'skeleton_index', 'joint_mask', 'tri_mask', 'xy', 'vp')
_engines = {}


def engine_for(variant, max_frames=64, ppc=12):
    key = (variant, max_frames, ppc)
    if key not in _engines:
        e = env(variant)
        _engines[key] = pkg('pipeline').Engine(e.params, e.calib, max_frames=max_frames, max_persons_per_camera=ppc)
    return _engines[key]


@pytest.fixture(scope='module', autouse=True)
def _close():
    yield
    for eng in _engines.values():
        eng.close()
    _engines.clear()


def same_arrays(a, b):
    assert a.n_frames == b.n_frames
    for f in FIELDS:
        x, y = np.asarray(getattr(a, f)), np.asarray(getattr(b, f))
        assert x.shape == y.shape, f
        assert x.tobytes() == y.tobytes(), f          # bit for bit (-0.0, NaN payloads included)


@pytest.mark.parametrize('variant,name', ALL_CASES)
def test_device_parse_equals_host_packer_on_golden_frames(variant, name):
    eng = engine_for(variant)
    packing = pkg('packing')
    _, frames = load_case(name, variant)
    text = json.dumps(frames * 5)
    host = packing.pack_json(text, eng.params)
    dev = eng.pack_json_device(text)
    assert dev is not None
    same_arrays(dev.download(), host)
    assert dev.n_heads == host.n_heads and dev.n_edge_nodes == host.n_edge_nodes
    assert dev.max_heads_per_frame() == host.max_heads_per_frame()
    # windows: stride and limit
    dev2 = eng.pack_json_device(text, frame_start=1, frame_step=2, max_frames=3)
    same_arrays(dev2.download(), packing.pack_json(text, eng.params, frame_start=1, frame_step=2, max_frames=3))


def test_device_parse_feeds_the_path_with_the_same_bits():
    """match + MLP 3D + DLT on a device-parsed batch = on the host-packed batch of the same document."""
    e = env('panoptic')
    eng = pkg('pipeline').Engine(e.params, e.calib, max_frames=32, max_persons_per_camera=10)
    try:
        eng.load_gat(*e.gat)
        eng.load_mlp(e.mlp)
        frames = []
        for name in ('c2_5x4_clean', 'c2_5x4_messy', 'c4_5x10', 'c1_2view_1person'):
            frames += load_case(name)[1]
        text = json.dumps(frames)
        db = eng.to_device(eng.pack_json(text))
        pd = eng.pack_json_device(text)
        assert pd is not None
        s1, p1, n1 = eng.match(db)
        s2, p2, n2 = eng.match(pd)
        assert torch.equal(s1, s2) and torch.equal(p1, p2) and torch.equal(n1, n2)
        assert torch.equal(eng.mlp3d(db, p1, n1)[0], eng.mlp3d(pd, p2, n2)[0])
        t1, t2 = eng.triangulate(db, p1, n1)[0], eng.triangulate(pd, p2, n2)[0]
        assert torch.equal(torch.nan_to_num(t1), torch.nan_to_num(t2))
    finally:
        eng.close()


def test_device_parse_numbers_are_python_floats():
    """x, y exactly as Python's float(): shortest-repr and 17-digit forms, integers, tiny and huge
    magnitudes, negative zero.  Tokens the exact fast path declines (more than 19 significant digits,
    exponents outside [-27, 55]) send the WINDOW to the host packer -- never a different double."""
    eng = engine_for('panoptic', max_frames=512, ppc=12)
    packing = pkg('packing')
    random.seed(7)
    vals = []
    while len(vals) < 36000:
        k = random.random()
        if k < 0.4:
            v = random.uniform(0, 1920)
        elif k < 0.6:
            v = round(random.uniform(0, 1920), random.randint(0, 12))
        elif k < 0.75:
            v = float(random.randint(0, 10 ** random.randint(1, 15)))
        elif k < 0.9:
            v = random.choice([-1, 1]) * (1e-6 + random.random() * 1e-5) * 10 ** random.randint(-2, 8)
        else:
            v = random.choice([0.0, -0.0, 0.5, 1e22, 1e-5, 123456.789e3, 5e-10, 9007199254740993.0])
        vals.append(v)
    sks = []
    for i in range(0, len(vals), 36):
        ch = vals[i:i + 36]
        sks.append({str(j): [j, ch[2 * j], ch[2 * j + 1], 1, 0.25] for j in range(18)})
    text = json.dumps([{'trackera': [json.dumps(sks[i:i + 10]), 0]} for i in range(0, len(sks), 10)])
    dev = eng.pack_json_device(text)
    assert dev is not None
    got = dev.download()
    assert got.xy.tobytes() == np.array(vals).reshape(-1, 18, 2).tobytes()
    same_arrays(got, packing.pack_json(text, eng.params))
    # declined tokens -> the host parser takes the window
    for tok in ('0.12345678901234567890123', '1e300', '4.9e-324', '123456789012345678901234567890'):
        doc = '[{"trackera": ["[{\\"5\\": [5, %s, 2.5, 1, 1]}]", 0]}]' % tok
        assert eng.pack_json_device(doc) is None
        assert packing.pack_json(doc, eng.params).xy[0, 5, 0] == float(tok)


def test_device_parse_leaves_unusual_shapes_to_the_host():
    """Everything outside the canonical shapes json.dumps writes is handed back (None), and the host packer
    then either packs it or raises -- the device never decides differently."""
    eng = engine_for('panoptic')
    packing = pkg('packing')
    ok = '[{}, {"zzz": ["[]", 0]}, {"trackerb": ["[{\\"ID\\": 7}, {\\"5\\": [5, 1.5, 2.5, 1, 0.25]}, {}]", 0.0, "no_image", [{"-1": [1, 2, 3]}]]}]'
    dev = eng.pack_json_device(ok)
    assert dev is not None and dev.n_frames == 3 and dev.n_heads == 1
    same_arrays(dev.download(), packing.pack_json(ok, eng.params))
    handed_back = [
        '[{"trackera": [[{"5": [5, 1.5, 2.5, 1, 1]}], 0]}]',                          # skeleton list not a string (lenient host form)
        '[{"trackera": ["[{\\"5\\": [5, 1.5, 2.5, true, 1]}]", 0]}]',                  # literal
        '[{"trackera": ["[{\\"ID\\": [1, 2], \\"5\\": [5, 1.5, 2.5, 1, 1]}]", 0]}]',   # nested ID value
        '[{"trackera": ["[{\\"+5\\": [5, 1.5, 2.5, 1, 1]}]", 0]}]',                    # key strtol accepts, the device does not
    ]
    for doc in handed_back:
        assert eng.pack_json_device(doc) is None, doc
        packing.pack_json(doc, eng.params)            # the host packs these
    rejected = [
        '[{"trackera": ["[{\\"99\\": [1,2,3,4,5]}]", 0]}]',      # joint id out of range
        '[{"trackera": ["[{\\"5\\": [1,2]}]", 0]}]',             # fewer than five numbers
        '[{"trackera": ["[{\\"5\\": [1,2,3,4,5]}", 0]}]',        # list not closed inside the string
        '[{"trackera": ["[{\\"5\\": [1,2,3,4,5]}] x", 0]}]',     # trailing garbage in the string
    ]
    for doc in rejected:
        assert eng.pack_json_device(doc) is None, doc
        with pytest.raises(ValueError):
            packing.pack_json(doc, eng.params)
    for doc in ('', '[{"trackera": 5}]', '[{"trackera": ["[]", 0]'):        # the first level already fails on the host
        with pytest.raises(ValueError):
            eng.pack_json_device(doc)


@pytest.mark.parametrize('preset,n', [('PANOPTIC', 120), ('ARPLAB', 60), ('RING23', 10)])
def test_device_parse_on_random_frame_shapes(preset, n):
    """Frames of random shape (0-6 persons, camera subsets and orders, empty cameras, spurious skeletons,
    dropped joints, "ID" keys, integer confidences) as wire-format JSON: same arrays as the host packer."""
    syn, par, cal = pkg('synthetic'), pkg('parameters'), pkg('calibration')
    params = par.select(preset)
    calib = cal.Calibration(params, syn.ring_transform_manager(params) if preset == 'RING23' else None)
    eng = pkg('pipeline').Engine(params, calib, max_frames=n, max_persons_per_camera=8)
    try:
        rng = random.Random(31 + len(preset))
        names = list(params.camera_names)
        frames = []
        for i in range(n):
            cams = names[:]
            rng.shuffle(cams)
            cams = cams[:rng.randint(1, len(cams))]
            empty = tuple(c for c in cams if rng.random() < 0.15)
            spec = syn.FrameSpec(persons=rng.randint(0, 6), cameras=cams, noise_px=rng.choice([0.0, 1.5]),
                                 joint_drop=rng.choice([0.0, 0.3, 0.9]), add_id_key=rng.random() < 0.5,
                                 spurious=rng.randint(0, 2), empty_cameras=empty, float_conf=rng.random() < 0.5)
            frames.append(syn.make_frame(calib, 9000 + i, spec)[0])
        text = json.dumps(frames)
        dev = eng.pack_json_device(text)
        assert dev is not None
        same_arrays(dev.download(), pkg('packing').pack_json(text, params))
    finally:
        eng.close()


def test_stream_json_device_parser_equals_host_parser():
    """Engine.stream_json with the device-side parser: same poses, chunk by chunk, as with the host packer --
    including a chunk that the device hands back (a literal in one frame) and the short last chunk."""
    e = env('panoptic')
    syn = pkg('synthetic')
    eng = pkg('pipeline').Engine(e.params, e.calib, max_frames=16, max_persons_per_camera=6)
    try:
        eng.load_gat(*e.gat)
        eng.load_mlp(e.mlp)
        frames = [syn.make_frame(e.calib, 4000 + i, syn.FrameSpec(persons=2 + i % 4, add_id_key=i % 3 == 0))[0] for i in range(70)]
        # frame 37 gets a literal the device does not take (`valid` = true): its chunk goes to the host packer
        sk = json.loads(frames[37]['trackerb'][0])
        who = next(q for q in sk if any(k != 'ID' for k in q))
        key = next(k for k in who if k != 'ID')
        who[key][3] = True
        frames[37]['trackerb'][0] = json.dumps(sk)
        text = json.dumps(frames)
        assert 'true' in text

        def run(parser):
            out = []
            for info, poses, n in eng.stream_json(text, chunk_frames=16, parser=parser):
                out.append((info.n_frames, poses.copy(), n.copy()))
            return out
        def run2():
            return [(info.n_frames, poses.copy(), n.copy()) for info, poses, n in eng.stream_json(text, chunk_frames=16, contexts=2)]
        a, b, c2 = run('host'), run('device'), run2()          # c2: windows take turns on two contexts
        assert [x[0] for x in a] == [x[0] for x in b] == [x[0] for x in c2] == [16, 16, 16, 16, 6]
        for other in (b, c2):
            for (_, p1, n1), (_, p2, n2) in zip(a, other):
                assert np.array_equal(n1, n2)
                for f in range(len(n1)):
                    assert np.array_equal(p1[f, :n1[f]], p2[f, :n1[f]])
    finally:
        eng.close()


@pytest.mark.parametrize('contexts', [1, 2])
def test_stream_json_window_edges(contexts):
    """The look-ahead loop of Engine.stream_json (parse of window i+1 queued before the compute of window i, K + 2 window
    slots, K windows in flight) at its edges: a document that ends exactly on a window boundary, one shorter than a
    window, an empty list, host-packed windows at the very start and the very end, and a consumer that stops early.
    Reference for every case: the host parser's stream."""
    e = env('panoptic')
    syn = pkg('synthetic')
    eng = pkg('pipeline').Engine(e.params, e.calib, max_frames=8, max_persons_per_camera=6)
    try:
        eng.load_gat(*e.gat)
        eng.load_mlp(e.mlp)
        frames = [syn.make_frame(e.calib, 8800 + i, syn.FrameSpec(persons=1 + i % 4))[0] for i in range(40)]

        def literal(f):                        # a shape the device parser hands back: `valid` = true
            f = json.loads(json.dumps(f))
            cam = next(c for c in f if json.loads(f[c][0]))
            sk = json.loads(f[cam][0])
            key = next(k for k in sk[0] if k != 'ID')
            sk[0][key][3] = True
            f[cam][0] = json.dumps(sk)
            return f

        def run(doc, parser, **kw):
            return [(info.n_frames, poses.copy(), n.copy()) for info, poses, n in eng.stream_json(doc, chunk_frames=8, parser=parser, **kw)]

        def same(a, b):
            assert [x[0] for x in a] == [x[0] for x in b]
            for (_, p1, n1), (_, p2, n2) in zip(a, b):
                assert np.array_equal(n1, n2)
                for f in range(len(n1)):
                    assert np.array_equal(p1[f, :n1[f]], p2[f, :n1[f]])
        cases = {
            'exact multiple': frames[:32],
            'one short window': frames[:5],
            'one full window': frames[:8],
            'host-packed first window': [literal(frames[0])] + frames[1:20],
            'host-packed last window': frames[:19] + [literal(frames[19])],
            'host-packed windows back to back': frames[:8] + [literal(frames[8])] + frames[9:16] + [literal(frames[16])] + frames[17:30],
        }
        for name, fr in cases.items():
            doc = json.dumps(fr)
            want = run(doc, 'host')
            assert sum(x[0] for x in want) == len(fr), name
            same(want, run(doc, 'device', contexts=contexts))
        assert run('[]', 'device', contexts=contexts) == []
        # a consumer that stops after the first window: nothing is left running, the next call starts clean
        doc = json.dumps(frames)
        gen = eng.stream_json(doc, chunk_frames=8, contexts=contexts)
        first = next(gen)
        keep = (first[0].n_frames, first[1].copy(), first[2].copy())
        gen.close()
        torch.cuda.synchronize()
        again = run(doc, 'device', contexts=contexts)
        same([keep], again[:1])
        same(run(doc, 'host'), again)
    finally:
        eng.close()



@pytest.mark.parametrize('variant,name', ALL_CASES)
def test_device_parsed_golden_frames_against_the_reference_fixtures(variant, name):
    """f1 against the REFERENCE directly (not through the host packer): every golden `*.frames.json` document, in the wire
    format, parsed on the device; the batch that comes out gives the reference's head rows (5e-7), the reference's scores (2e-5)
    and the reference's clusters (bit-exact) for every frame -- the fixtures the reference's own files produced
    (oracle/gen_golden.py)."""
    e = env(variant)
    eng = pkg('pipeline').Engine(e.params, e.calib, max_frames=8, max_persons_per_camera=10 if variant == 'panoptic' else 3)
    try:
        eng.load_gat(*e.gat)
        arr, frames = load_case(name, variant)
        onp = oracle()
        wire = [onp.processed_input(f) for f in frames]              # what the callers hand over (metrics_from_model.py:182-191)
        dev = eng.pack_json_device(json.dumps(wire))
        assert dev is not None and dev.n_frames == len(frames)
        host = dev.download()
        feat = eng.head_features(dev).cpu().numpy()
        scores, persons, n_persons = eng.match(dev)
        sh = eng.gat_scores(dev, heads=True)[1].cpu().numpy()
        eng.sync_status()
        scores, persons, n_persons = scores.cpu().numpy(), persons.cpu().numpy(), n_persons.cpu().numpy()
        J10 = len(e.params.joint_list) * 10
        for n in range(len(frames)):
            p = 'f%d_' % n
            h0, H, e0, M = host.frame_counts(n)
            if M == 0:
                assert (p + 'scores') not in arr.files or len(arr[p + 'scores']) == 0
                continue
            want = arr[p + 'scores']
            assert len(want) == H + M
            np.testing.assert_allclose(scores[e0:e0 + M], want[H:], rtol=0, atol=2e-5)
            np.testing.assert_allclose(sh[h0:h0 + H], want[:H], rtol=0, atol=2e-5)
            dense = np.zeros((H, e.meta['num_feats']), np.float32)
            rc = arr[p + 'feat_rc']
            sel = rc[:, 0] < H
            dense[rc[sel, 0], rc[sel, 1]] = arr[p + 'feat_v'][sel]
            for h in range(H):
                c = host.head_cam[h0 + h]
                np.testing.assert_allclose(feat[h0 + h].reshape(-1), dense[h, 2 + c * J10: 2 + (c + 1) * J10], rtol=0, atol=5e-7)
            wp = arr[p + 'persons']
            assert int(n_persons[n]) == len(wp) and np.array_equal(persons[n, :len(wp)], wp)
            assert list(arr[p + 'nodes_camera'][:H]) == [e.params.used_cameras_skeleton_matching[c] for c in host.head_cam[h0:h0 + H]]
    finally:
        eng.close()


def test_device_declines_what_only_the_host_dialect_covers():
    """JSON the device-side parser leaves to the host packer (pretty-printed inner strings = \\n escapes between the members,
    Infinity, null): the window comes back as "host parser, please", the host packer then gives the Python packer's arrays,
    and stream_json() delivers the same poses as for the canonical text.  A skeleton rejected BEFORE the member walk must not
    leave the previous window's rows in the scratch (stale rows used to turn into a spurious capacity error)."""
    e = env('panoptic')
    eng = pkg('pipeline').Engine(e.params, e.calib, max_frames=8, max_persons_per_camera=4)
    packing = pkg('packing')
    try:
        eng.load_gat(*e.gat)
        eng.load_mlp(e.mlp)
        syn = pkg('synthetic')
        frames = [syn.make_frame(e.calib, 300 + i)[0] for i in range(6)]
        canon = json.dumps(frames)
        pretty = json.dumps([{c: [json.dumps(json.loads(f[c][0]), indent=1)] + f[c][1:] for c in f} for f in frames])
        # first a FULL window through the device parser (fills the scratch), then the declined one
        assert eng.pack_json_device(canon) is not None
        assert eng.pack_json_device(pretty) is None
        a, b = packing.pack_json(pretty, e.params), packing.pack_json(canon, e.params)
        same_arrays(a, b)
        got = [(type(i).__name__, p.copy(), n.copy()) for i, p, n in eng.stream_json(pretty.encode(), chunk_frames=8)]
        want = [(p.copy(), n.copy()) for _, p, n in eng.stream_json(canon.encode(), chunk_frames=8)]
        assert [g[0] for g in got] == ['PackedBatch'] and len(want) == 1
        assert np.array_equal(got[0][2], want[0][1]) and np.array_equal(got[0][1], want[0][0])
        # two generators on one engine would share the result buffers: refused
        g1 = eng.stream_json(canon.encode(), chunk_frames=2)
        next(g1)
        with pytest.raises(RuntimeError):
            next(eng.stream_json(canon.encode(), chunk_frames=2))
        g1.close()
        assert len(list(eng.stream_json(canon.encode(), chunk_frames=4))) == 2
        with pytest.raises(ValueError):
            next(eng.stream_json(canon.encode(), chunk_frames=2, contexts=3))
    finally:
        eng.close()
