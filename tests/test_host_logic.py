"""CPU tests of the host-side logic: packing order, topology mirror, library symbols,
frame sharding + all-gather on gloo (world_size 2)."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

from conftest import CASES, ROOT, load_case, oracle, pkg


@pytest.mark.parametrize('name', CASES)
def test_packing_matches_reference_order(name, calib):
    onp = oracle()
    packing = pkg('packing')
    arr, frames = load_case(name)
    sm = list(calib.params.used_cameras_skeleton_matching)
    pb = packing.pack_frames([onp.processed_input(f) for f in frames], calib.params, keep_json=True)
    assert pb.n_frames == len(frames)
    for n, frame in enumerate(frames):
        p = 'f%d_' % n
        h0, H, e0, M = pb.frame_counts(n)
        N = int(arr[p + 'N'])
        assert H + M == N and M == len(arr[p + 'edge_nodes_indices'])
        assert [sm[c] for c in pb.head_cam[h0:h0 + H]] == list(arr[p + 'nodes_camera'][:H])
        assert list(pb.skeleton_index[h0:h0 + H]) == list(arr[p + 'skeleton_index'])
        # edge list rebuilt from the packed counts == the reference graph's edges
        pairs = packing.pairs_of_frame(pb.slot_n[n])
        src = list(range(H))
        dst = list(range(H))
        for m, (a, b) in enumerate(pairs):
            X = H + m
            src += [a, X, b, X, X]
            dst += [X, a, X, b, X]
        assert np.array_equal(np.array(src), arr[p + 'src']) and np.array_equal(np.array(dst), arr[p + 'dst'])
        # joint masks / values agree with the raw skeleton dicts
        pf = onp.parse_frame(onp.processed_input(frame), calib.params)
        for h, (cam, idx, sk) in enumerate(pf['heads']):
            keys = sorted(int(k) for k in sk if k != 'ID')
            assert [j for j in range(18) if pb.joint_mask[h0 + h] >> j & 1] == keys
            assert [j for j in range(18) if pb.tri_mask[h0 + h] >> j & 1] == [j for j in keys if sk[str(j)][0] > 0]
            for j in keys:
                assert pb.xy[h0 + h, j, 0] == sk[str(j)][1] and pb.xy[h0 + h, j, 1] == sk[str(j)][2]
                assert pb.vp[h0 + h, j, 0] == np.float32(sk[str(j)][3])


def test_empty_and_ragged_frames(calib):
    packing = pkg('packing')
    frames = [{}, {'trackera': ['[]', 0]}, {'trackera': ['[{"ID": 3}]', 0], 'trackerb': ['[{"5": [5, 10.0, 20.0, 1, 1]}]', 0]},
              {'unknown_cam': ['[{"5": [5, 1.0, 2.0, 1, 1]}]', 0]}]
    pb = packing.pack_frames(frames, calib.params)
    assert pb.n_heads == 1 and pb.n_edge_nodes == 0
    assert list(pb.frame_head_off) == [0, 0, 0, 1, 1]
    assert pb.slot_n[2].tolist()[:2] == [0, 1]


def test_library_exports_every_declared_symbol():
    lib = pkg('lib')
    header = open(os.path.join(ROOT, 'include', 'mpe.h')).read()
    declared = set(re.findall(r'\b(mpe_[a-z0-9_]+)\s*\(', header))
    assert declared == set(lib.SYMBOLS), declared ^ set(lib.SYMBOLS)
    if not os.path.exists(lib.LIB_PATH):
        pytest.skip('library not built yet (run __graft_entry__.build())')
    handle = ctypes.CDLL(lib.LIB_PATH)       # load only: no compute without a GPU
    for name in declared:
        assert hasattr(handle, name), name


def test_header_is_plain_c_and_a_c_host_packs_like_the_binding(tmp_path, calib):
    """include/mpe.h is what a cgo / JNI / N-API binding includes: it has to be valid C on its own (C99 and C++11, -pedantic -Werror;
    round 6 found it leaning on a C++ translation unit's <cstddef> for size_t).  And a plain C program that includes it
    (tests/native/c_host_packer.c) packs a wire-format document through mpe_pack_json -- the host side of the boundary, no GPU -- into
    the same arrays as the Python binding (graph_generator.py:573-605 is the order they restate)."""
    import json
    import shutil
    import subprocess
    gcc, gxx = shutil.which('gcc'), shutil.which('g++')
    if not gcc or not gxx:
        pytest.skip('gcc / g++ not available')
    inc = os.path.join(ROOT, 'include')
    (tmp_path / 'h.c').write_text('#include "mpe.h"\nint main(void) { mpe_config c; mpe_batch b; mpe_pack_dst d; (void)c; (void)b; (void)d; return 0; }\n')
    (tmp_path / 'h.cpp').write_text('#include "mpe.h"\nint main() { mpe_config c; (void)c; return 0; }\n')
    subprocess.run([gcc, '-std=c99', '-Wall', '-Wextra', '-pedantic', '-Werror', '-I', inc, '-fsyntax-only', str(tmp_path / 'h.c')], check=True, capture_output=True)
    subprocess.run([gxx, '-std=c++11', '-Wall', '-Wextra', '-pedantic', '-Werror', '-I', inc, '-fsyntax-only', str(tmp_path / 'h.cpp')], check=True, capture_output=True)
    lib = pkg('lib')
    if not os.path.exists(lib.LIB_PATH):
        pytest.skip('library not built yet (run __graft_entry__.build())')
    libdir, exe = os.path.dirname(lib.LIB_PATH), str(tmp_path / 'c_host_packer')
    hip_lib = '/opt/rocm/lib'
    subprocess.run([gcc, '-std=c99', '-Wall', '-Wextra', '-pedantic', '-Werror', '-I', inc, os.path.join(ROOT, 'tests', 'native', 'c_host_packer.c'),
                    '-L', libdir, '-lmpe_hip', '-Wl,-rpath,' + libdir, '-Wl,-rpath,' + hip_lib, '-o', exe], check=True, capture_output=True, timeout=300)
    syn, packing = pkg('synthetic'), pkg('packing')
    frames = [syn.make_frame(calib, 10 + i, syn.FrameSpec(persons=2 + i, empty_cameras=(calib.params.used_cameras_skeleton_matching[2],) if i == 1 else ()))[0]
              for i in range(3)]
    text = json.dumps(frames)
    (tmp_path / 'doc.json').write_text(text)
    r = subprocess.run([exe, str(tmp_path / 'doc.json'), str(len(calib.params.joint_list))] + list(calib.params.used_cameras_skeleton_matching),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r'frames (\d+) heads (\d+) edge_nodes (\d+) sum (\S+) masks (\d+) last_off (\d+)', r.stdout)
    assert m, r.stdout
    pb = packing.pack_json(text.encode(), calib.params)
    assert (int(m.group(1)), int(m.group(2)), int(m.group(3)), int(m.group(6))) == (pb.n_frames, pb.n_heads, pb.n_edge_nodes, pb.n_heads)
    t = np.asarray(pb.xy, np.float64).reshape(-1) + np.asarray(pb.vp, np.float32).reshape(-1).astype(np.float64)
    assert float(m.group(4)) == float(np.cumsum(t)[-1])          # the C loop's sequential sum, to the last bit
    assert int(m.group(5)) == int(sum(int(x) % 1000003 for x in pb.joint_mask) + int(np.asarray(pb.head_cam, np.int64).sum()))


def test_every_entry_point_refuses_null_arguments_without_crashing():
    """The error behaviour of the boundary, without a GPU: every function include/mpe.h declares, called with a NULL context / NULL
    pointers / zero sizes, returns MPE_ERR_INVALID (or does nothing, for the void ones) -- no entry point dereferences before it checks.
    In a child process: a crash there is an exit code here, not the end of the test run."""
    import subprocess
    import sys
    lib = pkg('lib')
    if not os.path.exists(lib.LIB_PATH):
        pytest.skip('library not built yet (run __graft_entry__.build())')
    code = (
        "import ctypes as C, importlib, sys\n"
        "sys.path.insert(0, %r)\n"
        "L = importlib.import_module('3d_multi_pose_estimator_amd.lib')\n"
        "lib = L.load()\n"
        "for name, (res, args) in L.SYMBOLS.items():\n"
        "    vals = [0 if a in (C.c_int32, C.c_int, C.c_uint32, C.c_size_t) else 0.0 if a is C.c_float else None for a in args]\n"
        "    r = getattr(lib, name)(*vals)\n"
        "    if res is C.c_int:\n"
        "        assert r == -1, (name, r)\n"
        "assert lib.mpe_last_error(None) == b'null context'\n"
        "print('ok', len(L.SYMBOLS))\n" % ROOT)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith('ok'), r.stdout + r.stderr


def test_no_product_import_of_oracle():
    """The product must not route through the oracle or any CPU fallback."""
    pkg_dir = os.path.join(ROOT, '3d_multi_pose_estimator_amd')
    for dirpath, _, files in os.walk(pkg_dir):
        for fn in files:
            if fn.endswith('.py') or fn.endswith('.hip') or fn.endswith('.h'):
                txt = open(os.path.join(dirpath, fn)).read()
                assert 'oracle_np' not in txt and 'import oracle' not in txt, fn


def _worker(rank, world, n_frames, port, ret):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    d = pkg('distributed')
    lo, hi, per = d.shard_range(n_frames, rank, world)
    # stand-in for the per-rank HIP result: values that encode the global frame index
    poses = torch.arange(lo, hi, dtype=torch.float32).view(-1, 1, 1, 1).expand(-1, 2, 18, 3).contiguous()
    n_p = torch.arange(lo, hi, dtype=torch.int32) % 3
    gp, gn = d.all_gather_results(d.pad_to(poses, per), d.pad_to(n_p, per), n_frames)
    ok = bool(torch.equal(gp[:, 0, 0, 0], torch.arange(n_frames, dtype=torch.float32))
              and torch.equal(gn, torch.arange(n_frames, dtype=torch.int32) % 3))
    ret[rank] = ok
    dist.destroy_process_group()


@pytest.mark.parametrize('n_frames', [10, 11])
def test_shard_and_all_gather_gloo_world2(n_frames):
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    ret = ctx.Manager().dict()
    port = 29500 + (os.getpid() % 500) + n_frames
    procs = [ctx.Process(target=_worker, args=(r, 2, n_frames, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret[0] and ret[1]


def test_shard_range_covers_everything():
    d = pkg('distributed')
    for n in (0, 1, 7, 8, 1000, 100000):
        for w in (1, 2, 4, 8):
            seen = []
            for r in range(w):
                lo, hi, per = d.shard_range(n, r, w)
                assert hi - lo <= per
                seen += list(range(lo, hi))
            assert seen == list(range(n))


@pytest.mark.parametrize('name', CASES)
def test_native_json_packer_equals_python_packer(name, calib):
    """csrc/packer.cpp (mpe_pack_json) against packing.pack_frames on the fixture frames."""
    import json
    packing = pkg('packing')
    if not os.path.exists(pkg('lib').LIB_PATH):
        pytest.skip('library not built')
    arr, frames = load_case(name)
    text = json.dumps(frames)
    a = packing.pack_json(text, calib.params)
    b = packing.pack_frames(frames, calib.params)
    for f in ('frame_head_off', 'frame_en_off', 'slot_cam', 'slot_n', 'head_cam', 'skeleton_index', 'joint_mask',
              'tri_mask', 'xy', 'vp'):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f
    # stride / limit and a single frame object
    c = packing.pack_json(text, calib.params, frame_start=1, frame_step=2, max_frames=1)
    if len(frames) > 1:
        d = packing.pack_frames(frames[1:2], calib.params)
        assert np.array_equal(c.xy, d.xy) and np.array_equal(c.head_cam, d.head_cam)
    e = packing.pack_json(json.dumps(frames[0]), calib.params)
    assert e.n_frames == 1 and np.array_equal(e.xy, packing.pack_frames(frames[:1], calib.params).xy)


@pytest.mark.parametrize('name', CASES)
def test_native_view_packer_equals_python_packer(name, calib):
    """mpe_pack_views_into (the per-frame mirrors: one frame dict, the camera texts as the caller holds them) against
    packing.pack_frames, frame by frame; the arrays lie in one buffer at the offsets the device copy uses."""
    packing = pkg('packing')
    if not os.path.exists(pkg('lib').LIB_PATH):
        pytest.skip('library not built')
    arr, frames = load_case(name)
    fields = ('frame_head_off', 'frame_en_off', 'slot_cam', 'slot_n', 'head_cam', 'skeleton_index', 'joint_mask', 'tri_mask', 'xy', 'vp')
    for frame in frames:
        ref = packing.pack_frames([frame], calib.params)
        got = packing.pack_views(frame, calib.params, max_heads=max(1, ref.n_heads))
        buf, layout = got.upload_layout[:2]
        for f in fields:
            a, b = np.asarray(getattr(ref, f)), np.asarray(getattr(got, f))
            assert a.shape == b.shape and np.array_equal(a, b), f
            assert layout[f] % 256 == 0 and getattr(got, f).ctypes.data == buf.ctypes.data + layout[f] or b.size == 0
        if ref.n_heads > 1:
            with pytest.raises(ValueError, match='exceed'):
                packing.pack_views(frame, calib.params, max_heads=ref.n_heads - 1)


def test_native_view_packer_declines_what_json_loads_would_not_turn_into_a_list(calib):
    packing = pkg('packing')
    if not os.path.exists(pkg('lib').LIB_PATH):
        pytest.skip('library not built')
    P = calib.params
    good = '[{"5": [5, 1.5, 2.5, 1, 0.25]}]'
    ok = packing.pack_views({'zzz': ['[]', 0], 'trackerb': ['  ' + good + ' \n', 0.0], 'trackera': ['[]', 0.0]}, P, 8)
    assert ok.n_heads == 1 and ok.slot_cam[0].tolist()[:2] == [1, 0] and ok.slot_n[0].tolist()[:2] == [1, 0] and ok.xy[0, 5].tolist() == [1.5, 2.5]
    for bad in (good + ' x', json.dumps(good), '', '{"5": [5, 1, 2, 1, 1]}', good[:-1], '[{"99": [1, 2, 3, 4, 5]}]'):
        with pytest.raises(ValueError):
            packing.pack_views({'trackera': [bad, 0.0]}, P, 8)
    with pytest.raises(AttributeError):                       # (graph_generator._pack_one hands such frames to the Python packer)
        packing.pack_views({'trackera': [[{"5": [5, 1.5, 2.5, 1, 0.25]}], 0.0]}, P, 8)
    assert packing.pack_views({}, P, 8).n_heads == 0


def test_native_json_packer_rejects_garbage(calib):
    packing = pkg('packing')
    if not os.path.exists(pkg('lib').LIB_PATH):
        pytest.skip('library not built')
    for bad in ('', '[{"trackera": 5}]', '[{"trackera": ["[{\\"99\\": [1,2,3,4,5]}]", 0]}]', '[{"trackera": ["[{\\"5\\": [1,2]}]", 0]}]'):
        with pytest.raises(ValueError):
            packing.pack_json(bad, calib.params)
    ok = packing.pack_json('[{}, {"zzz": ["[]", 0]}, {"trackerb": ["[{\\"ID\\": 7}, {\\"5\\": [5, 1.5, 2.5, 1, 0.25]}]", 0.0, "no_image", [{"-1": [1, 2, 3]}]]}]', calib.params)
    assert ok.n_frames == 3 and ok.n_heads == 1 and ok.skeleton_index.tolist() == [1]
    assert ok.xy[0, 5].tolist() == [1.5, 2.5] and ok.vp[0, 5].tolist() == [1.0, 0.25]


def test_native_json_packer_numbers_are_python_floats(calib):
    """x, y must come out as exactly the doubles Python's float() gives (the reference feeds
    them to OpenCV in f64): shortest-repr and full 17-digit forms, tiny and huge magnitudes."""
    import json
    import random
    import struct
    packing = pkg('packing')
    if not os.path.exists(pkg('lib').LIB_PATH):
        pytest.skip('library not built')
    random.seed(5)
    vals = []
    while len(vals) < 36000:
        k = random.random()
        if k < 0.35:
            v = random.uniform(0, 1920)
        elif k < 0.5:
            v = round(random.uniform(0, 1920), random.randint(0, 12))
        elif k < 0.7:
            v = struct.unpack('d', struct.pack('Q', random.getrandbits(62)))[0]
        elif k < 0.8:
            v = float(random.randint(0, 10 ** random.randint(1, 18)))
        else:
            v = random.uniform(-1e-5, 1e-5) * 10 ** random.randint(-10, 10)
        if v != v or abs(v) == float('inf'):
            continue
        vals.append(v)
    sks = []
    for i in range(0, len(vals), 36):
        ch = vals[i:i + 36]
        sks.append({str(j): [j, ch[2 * j], ch[2 * j + 1], 1, 1] for j in range(18)})
    text = json.dumps([{'trackera': [json.dumps(sks[i:i + 10]), 0]} for i in range(0, len(sks), 10)])
    pb = packing.pack_json(text, calib.params)
    assert np.array_equal(pb.xy, np.array(vals).reshape(-1, 18, 2))


def test_native_json_packer_never_reads_past_len(tmp_path, calib):
    """The C entry point takes (json, len): no NUL terminator may be assumed.  CPU build of
    packer.cpp under AddressSanitizer, fed every prefix of a list document and of a single-frame
    document from exactly sized heap buffers (tests/native/packer_prefix_driver.cpp)."""
    import json
    import shutil
    import subprocess
    if not shutil.which('g++'):
        pytest.skip('no g++')
    exe = str(tmp_path / 'packer_asan')
    subprocess.check_call(['g++', '-std=c++17', '-O1', '-g', '-fsanitize=address', '-fno-omit-frame-pointer',
                           os.path.join(ROOT, '3d_multi_pose_estimator_amd', 'csrc', 'packer.cpp'),
                           os.path.join(ROOT, 'tests', 'native', 'packer_prefix_driver.cpp'), '-lpthread', '-o', exe])
    _, frames = load_case('c1_2view_1person')
    cams = list(calib.params.used_cameras_skeleton_matching)
    docs = {'list': json.dumps(frames), 'single': json.dumps(frames[0]),
            'numbers': json.dumps([{cams[0]: [json.dumps([{'3': [3, 1e-320, 1234567890.12345678901, 1, 0.1234567890123456789]}]), 0]}])}
    for name, text in docs.items():
        path = tmp_path / (name + '.json')
        path.write_text(text)
        res = subprocess.run([exe, str(path), '18'] + cams, capture_output=True, text=True, timeout=300)
        assert res.returncode == 0 and 'AddressSanitizer' not in res.stderr, res.stderr[-2000:]
        full, accepted = (int(x) for x in res.stdout.split())
        assert full == (len(frames) if name == 'list' else 1) and accepted >= 1


def test_batch_arena_layout_roundtrip(calib):
    """One contiguous buffer per batch (single H2D copy): every array lands 256-byte aligned and
    reads back bit-identically."""
    packing = pkg('packing')
    _, frames = load_case('c2_5x4_messy')
    pb = packing.pack_frames(frames, calib.params)
    arena = packing.BatchArena(pb, 'host').fill(pb)
    raw = arena.buf.numpy()
    for name, (off, n, dt) in arena.offsets.items():
        assert off % 256 == 0
        got = raw[off: off + n * np.dtype(dt).itemsize].view(dt)
        assert np.array_equal(got, np.asarray(getattr(pb, name)).reshape(-1).view(dt)), name
    assert arena.nbytes < 1.1 * sum(np.asarray(getattr(pb, n)).nbytes for n, _ in packing.ARRAYS) + 9 * 256


@pytest.mark.parametrize('extra,total,scaling', [(['--frames', '7'], 14, 'weak'), (['--total-frames', '11'], 11, 'strong')])
def test_bench_gpus_flag_spawns_ranks(extra, total, scaling):
    """`python bench.py --gpus 2` (no torchrun, no WORLD_SIZE) must itself start two ranks: the
    parent spawns them before any GPU call, rank 0 reports n_gpus = 2 and the world size the
    process group saw.  --dry-run keeps the launch, rendezvous (gloo), sharding and all-gather and
    drops the GPU work, so this runs on the CPU box."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run', '--steps', '2'] + extra,
                         capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [l for l in res.stdout.splitlines() if l.startswith('{')]
    assert len(line) == 1, res.stdout
    out = json.loads(line[0])
    assert out['n_gpus'] == 2 and out['world_size_seen'] == 2 and out['launcher'] == 'bench.py spawn'
    assert out['gather_ok'] and out['scaling'] == scaling and out['frames_per_step_total'] == total


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_under_test', os.path.join(ROOT, 'bench.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_launch_ranks_parent_makes_no_gpu_call(monkeypatch):
    """The NON --dry-run parent of `bench.py --gpus N`: the device count comes from the environment
    or sysfs (never torch.cuda / HIP), too few devices -> exit code 2 before anything is started,
    enough devices -> N children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set.  torch must not
    even be imported by the parent."""
    import subprocess
    import sys
    bench = _bench_module()
    started = []

    class FakeProc:
        def __init__(self, argv, env=None, stdout=None):
            started.append((argv, env, stdout))
            self.returncode = None

        def poll(self):
            self.returncode = 0
            return 0

        def wait(self, timeout=None):
            return 0

        def kill(self):
            pass

    args = bench.parse_args(['--gpus', '4'])
    torch_before = 'torch' in sys.modules
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4'])
    assert bench.launch_ranks(args, count_fn=lambda: 2, popen=FakeProc) == 2 and not started
    assert bench.launch_ranks(args, count_fn=lambda: 8, popen=FakeProc) == 0
    assert len(started) == 4
    for r, (argv, env, stdout) in enumerate(started):
        assert env['RANK'] == env['LOCAL_RANK'] == str(r) and env['WORLD_SIZE'] == '4'
        assert env['MASTER_ADDR'] == '127.0.0.1' and env['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
        assert (stdout is None) == (r == 0)                  # only rank 0 owns stdout
        assert argv[-2:] == ['--gpus', '4']
    assert ('torch' in sys.modules) == torch_before          # the parent did not import torch
    # unknown count (no env list, no sysfs): start the ranks and let them find out
    started.clear()
    assert bench.launch_ranks(args, count_fn=lambda: None, popen=FakeProc) == 0 and len(started) == 4
    # the count itself: an explicit list wins, no HIP involved
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '0,3,5')
    assert bench.visible_gpu_count() == 3
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '')
    assert bench.visible_gpu_count() == 0
    monkeypatch.delenv('HIP_VISIBLE_DEVICES')
    monkeypatch.delenv('ROCR_VISIBLE_DEVICES', raising=False)
    monkeypatch.delenv('CUDA_VISIBLE_DEVICES', raising=False)
    assert bench.visible_gpu_count() in (None, 0) or bench.visible_gpu_count() > 0   # sysfs or nothing; never raises
    src = open(os.path.join(ROOT, 'bench.py')).read()
    body = src[src.index('def launch_ranks'):src.index('# one rank')]
    assert '\n        import torch' not in body and 'device_count(' not in body and 'torch.' not in body


def test_launch_ranks_first_failing_rank_stops_the_others(monkeypatch):
    """A rank that dies early must end the job: its siblings (blocked in rendezvous / all-gather)
    are killed and the parent returns that rank's code instead of waiting for the deadline."""
    import subprocess
    import sys
    import time as _t
    bench = _bench_module()
    script = 'import os,sys,time\nsys.exit(7) if os.environ["RANK"] == "1" else time.sleep(120)\n'

    def popen(argv, env=None, stdout=None):
        return subprocess.Popen([sys.executable, '-c', script], env=env, stdout=subprocess.DEVNULL)
    args = bench.parse_args(['--gpus', '3'])
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '3'])
    t0 = _t.time()
    assert bench.launch_ranks(args, count_fn=lambda: 3, popen=popen) == 7
    assert _t.time() - t0 < 30


def test_bench_under_external_launcher_uses_its_world():
    """Under torchrun-style env (RANK/WORLD_SIZE set) bench.py must NOT spawn again."""
    import json
    import subprocess
    import sys
    port = 29700 + os.getpid() % 200
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.pop('MPE_BENCH_SPAWNED', None)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run', '--steps', '1',
                                       '--frames', '5'], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    line = [l for l in outs[0][0].splitlines() if l.startswith('{')]
    out = json.loads(line[0])
    assert out['n_gpus'] == 2 and out['launcher'] == 'external' and out['gather_ok']
    assert not [l for l in outs[1][0].splitlines() if l.startswith('{')]      # only rank 0 prints


def test_param_watch_sees_replaced_parameters_and_modules():
    """runtime.ParamWatch (the mirrors' cheap "did my weights change" check): same tuple while nothing changes; a new tuple
    after an in-place update, after `module.weight = nn.Parameter(...)`, after a swapped submodule, after register_parameter, after
    a pruning-style re-registration."""
    from torch import nn
    rt = pkg('runtime')
    net = nn.Sequential(nn.Linear(4, 3), nn.LeakyReLU(0.1), nn.Sequential(nn.Linear(3, 2)))
    w = rt.ParamWatch(net)
    v0 = w.version()
    assert w.version() == v0 and len(v0) == 4
    with torch.no_grad():
        net[0].weight.add_(1.0)                                   # in place: version counter
    v1 = w.version()
    assert v1 != v0
    net[0].weight = nn.Parameter(torch.zeros(3, 4))                # a NEW Parameter object under the same name
    v2 = w.version()
    assert v2 != v1 and any(ptr == net[0].weight.data_ptr() for ptr, _ in v2)
    net[2][0] = nn.Linear(3, 2)                                    # a swapped submodule
    v3 = w.version()
    assert v3 != v2 and any(ptr == net[2][0].weight.data_ptr() for ptr, _ in v3)
    net[2].register_parameter('extra', nn.Parameter(torch.ones(1)))
    v4 = w.version()
    assert len(v4) == 5
    orig = net[0].weight                                           # pruning-style: the parameter moves to another name
    del net[0]._parameters['weight']
    net[0].register_parameter('weight_orig', orig)
    assert w.version() != v4
    assert w.version() == w.version()


def test_bench_refuses_a_world_size_that_is_not_gpus_before_any_gpu_work():
    """One contract for the launch: the ranks that exist are the job and --gpus must name their number; a mismatch exits 3
    at once -- no process group, no device -- with a message that says what to pass."""
    import subprocess
    import sys
    env = dict(os.environ, RANK='0', LOCAL_RANK='0', WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT='29999')
    env.pop('MPE_BENCH_FORCE_DIST', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--dry-run', '--steps', '1', '--frames', '5'],
                       capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 3, (r.returncode, r.stderr[-500:])
    assert '--gpus 1 but WORLD_SIZE=2' in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith('{')]


def test_bench_reads_the_newest_pmc_traffic_record():
    """roofline.traffic_source names the newest profiles/rNN_pmc_traffic.json (the line cited a round-4 file while a
    round-5 one existed)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    names = bench.traffic_files()
    have = sorted(f for f in os.listdir(os.path.join(ROOT, 'profiles')) if f.endswith('_pmc_traffic.json'))
    assert names and sorted(names) == have and names[0] == have[-1]


def test_projection_helpers_match_reference_formulas():
    """pose_estimator_utils.apply_distortion / from_homogeneous(2) / get_distortion_coefficients:
    host-side tensor helpers of the reprojection check (reference pose_estimator_utils.py:32-50).
    Checked against the closed form, and apply_distortion against the harness' own projection."""
    peu = pkg('pose_estimator_utils')
    par = pkg('parameters').parameters
    v = torch.tensor([[0.10, -0.20, 0.30], [0.05, 0.40, -0.25], [1.0, 1.0, 1.0]])
    kd = peu.get_distortion_coefficients(2).cpu()
    assert kd.tolist() == pytest.approx([par.kd0[2], par.kd1[2], par.kd2[2]])
    out = peu.apply_distortion(kd, v)
    r2 = v[0] ** 2 + v[1] ** 2
    f = 1 + kd[0] * r2 + kd[1] * r2 ** 2 + kd[2] * r2 ** 3
    assert torch.allclose(out[0], v[0] * f, rtol=1e-6, atol=0) and torch.allclose(out[1], v[1] * f, rtol=1e-6, atol=0)
    assert torch.equal(out[2], v[2]) and torch.equal(v[2], torch.ones(3))          # input untouched, w kept
    h = torch.tensor([[2.0, 4.0], [6.0, 8.0], [2.0, 4.0]])
    assert torch.equal(peu.from_homogeneous(h), torch.tensor([[1.0, 1.0], [3.0, 2.0]]))
    assert torch.equal(peu.from_homogeneous2(h), torch.tensor([[1.0, 1.0], [3.0, 2.0], [1.0, 1.0]]))


def test_json_index_windows_equal_one_shot(calib):
    """mpe_json_index: a document consumed in windows (scanned once, resumably) packs to the same
    arrays as the one-shot packer, including the last short window and a window past the end."""
    import json
    packing = pkg('packing')
    if not os.path.exists(pkg('lib').LIB_PATH):
        pytest.skip('library not built')
    frames = []
    for name in CASES:
        frames += load_case(name)[1]
    text = json.dumps(frames)
    whole = packing.pack_json(text, calib.params)
    V, J = whole.V, whole.J
    arena = packing.CapacityArena(V, J, 4, 400, 'host')
    ix = packing.JsonIndex(text)
    h = 0
    for start in range(0, len(frames) + 4, 4):
        pb = packing.pack_json_into(ix, calib.params, arena, frame_start=start, max_frames=4)
        assert pb.n_frames == max(0, min(4, len(frames) - start))
        n = pb.n_heads
        assert np.array_equal(pb.xy, whole.xy[h:h + n]) and np.array_equal(pb.vp, whole.vp[h:h + n])
        assert np.array_equal(pb.head_cam, whole.head_cam[h:h + n]) and np.array_equal(pb.joint_mask, whole.joint_mask[h:h + n])
        assert np.array_equal(pb.slot_n, whole.slot_n[start:start + 4])
        h += n
    assert h == whole.n_heads
    ix.close()
    # a cut-off document: the background scan fails at the damage; windows in front of it still pack, the
    # first window that needs a frame behind it raises, and closing the index (joins the scan thread) works
    cut = text.encode()[:int(len(text) * 0.7)]
    ix = packing.JsonIndex(cut)
    pb = packing.pack_json_into(ix, calib.params, arena, frame_start=0, max_frames=2)
    assert pb.n_frames == 2 and np.array_equal(pb.slot_n, whole.slot_n[:2])
    with pytest.raises(ValueError, match='unterminated'):
        for start in range(2, len(frames) + 4, 2):
            packing.pack_json_into(ix, calib.params, arena, frame_start=start, max_frames=2)
    ix.close()
    # an index that is dropped before anything was packed (scan thread still running or already done)
    packing.JsonIndex(text).close()


def test_parallel_frame_scan_equals_serial_scan(calib, monkeypatch):
    """mpe_json_index cuts a large document at guessed frame boundaries and scans the parts concurrently; every
    guess is verified by the scanner that reaches it.  Right guesses, WRONG guesses (the boundary pattern inside
    nested ground-truth lists, i.e. at another depth) and a document damaged behind a verified boundary must all
    give exactly what the serial scan gives."""
    import json
    packing = pkg('packing')
    if not os.path.exists(pkg('lib').LIB_PATH):
        pytest.skip('library not built')
    _, frames = load_case('c2_5x4_messy')
    base = []
    for i in range(260):
        f = json.loads(json.dumps(frames[i % len(frames)]))
        for cam in f:
            # ground truth with the frame-boundary byte pattern "]]}, {" at depth 3
            f[cam] = [f[cam][0], float(i), 'no_image', [{'0': [[i, 1.5]]}, {'1': [[2.5, i]]}, {'-1': [[0, 0]]}] * 6]
        base.append(f)
    text = json.dumps(base * 8).encode()
    assert len(text) > (5 << 20) and b']]}, {"' in text
    V, J = len(calib.params.used_cameras_skeleton_matching), len(calib.params.joint_list)

    def windows(threads, doc):
        monkeypatch.setenv('MPE_SCAN_THREADS', str(threads))
        ix = packing.JsonIndex(doc)
        try:
            out = []
            arena = packing.CapacityArena(V, J, 500, 500 * 40, 'host')
            start = 0
            while True:
                pb = packing.pack_json_into(ix, calib.params, arena, frame_start=start, max_frames=500)
                if pb.n_frames == 0:
                    break
                out.append((pb.n_frames, pb.frame_head_off.copy(), pb.xy.copy(), pb.slot_cam.copy()))
                start += pb.n_frames
                if pb.n_frames < 500:
                    break
            return out
        finally:
            ix.close()
    ref = windows(1, text)
    assert sum(w[0] for w in ref) == len(base) * 8
    for threads, chunk_kb in ((2, None), (4, None), (7, None), (3, 256), (4, 64)):      # few large parts; many small ones taken in order
        if chunk_kb:
            monkeypatch.setenv('MPE_SCAN_CHUNK_KB', str(chunk_kb))
        got = windows(threads, text)
        monkeypatch.delenv('MPE_SCAN_CHUNK_KB', raising=False)
        assert len(got) == len(ref)
        for a, b in zip(ref, got):
            assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
    # a document cut off in its last eighth: the windows in front of the damage are served, the one that reaches it
    # fails -- with the same number of windows served as under the serial scan
    cut = text[:len(text) - len(text) // 8]
    served = []
    for threads in (1, 4):
        monkeypatch.setenv('MPE_SCAN_THREADS', str(threads))
        ix = packing.JsonIndex(cut)
        try:
            arena = packing.CapacityArena(V, J, 500, 500 * 40, 'host')
            n_ok, start = 0, 0
            with pytest.raises(ValueError):
                while True:
                    pb = packing.pack_json_into(ix, calib.params, arena, frame_start=start, max_frames=500)
                    assert pb.n_frames == 500
                    n_ok += 1
                    start += 500
            served.append(n_ok)
        finally:
            ix.close()
    assert served[0] == served[1] >= 2


def test_frame_scanner_fuzz_simd_and_scalar():
    """The frame scanner's byte searches (AVX2 where the CPU has it, scalar otherwise) on random
    documents with escaped quotes, backslash runs and brackets inside strings: all frames found,
    each parses.  The scalar searches are chosen at load time (MPE_PACK_NO_SIMD), hence subprocesses."""
    import subprocess
    import sys
    if not os.path.exists(pkg('lib').LIB_PATH):
        pytest.skip('library not built')
    script = os.path.join(ROOT, 'tests', 'native', 'scan_fuzz.py')
    for seed in (1, 2, 3):
        for no_simd in (False, True):
            env = dict(os.environ)
            env.pop('MPE_PACK_NO_SIMD', None)
            if no_simd:
                env['MPE_PACK_NO_SIMD'] = '1'
            r = subprocess.run([sys.executable, script, str(seed)], env=env, capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stdout + r.stderr
            assert 'frames 300 simd %s' % (not no_simd) in r.stdout


def test_tools_do_not_use_the_oracle():
    """The oracle is test infrastructure: only tests/ (incl. tests/checkers/), smoke() and the bench's
    cpu_baseline leg may touch it -- the diagnostics under tools/ must not."""
    for name in sorted(os.listdir(os.path.join(ROOT, 'tools'))):
        if name.endswith(('.py', '.sh')):
            text = open(os.path.join(ROOT, 'tools', name)).read()
            assert 'oracle' not in text.replace('vs the oracle', '').replace("the oracle's", ''), name


def test_fused_attention_window_isa(tmp_path):
    """Static pin of the hand-scheduled window of k_gat_fused (csrc/gat.hip): five asynchronous table loads,
    eight LDS-DMA pieces and a COUNTED wait (`s_waitcnt vmcnt(8)` = "my five values are here, the image is
    still landing").  The hardware gives no protection to the five destination registers until that wait;
    the source keeps the whole window in one asm statement, and this test checks what the compiler emitted
    for every VEC = 4 instantiation: between the first table load and the wait there are exactly the five
    global_load_dword, eight global_load_lds_dwordx4 and their M0 bookkeeping -- no other vector, memory, LDS
    or branch instruction, nothing that reads or writes the five destination VGPRs, no scratch access.
    (Reference semantics it protects: gat2.py:57-66,78-88.)  Compiles gat.hip to gfx950 assembly (no GPU)."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('hipcc not available')
    src = os.path.join(ROOT, '3d_multi_pose_estimator_amd', 'csrc', 'gat.hip')
    out = str(tmp_path / 'gat.s')
    subprocess.run([hipcc, '-O3', '-std=c++17', '--offload-arch=gfx950', '-S', '--cuda-device-only', '-o', out, src],
                   check=True, capture_output=True, timeout=600)
    text = open(out).read()
    kernels = re.findall(r'^(_ZN3mpe11k_gat_fusedILi4ELi\d+EE\S*):', text, re.M)
    assert kernels, 'no k_gat_fused<4, G> instantiation found'
    allowed = re.compile(r'^(s_mov_b32 m0, s\d+|s_mov_b32 s\d+, m0|s_add_u32 m0, m0, 0x1000|s_nop \d+|global_load_lds_dwordx4 v\[\d+:\d+\], off)$')
    for k in kernels:
        body = text[text.index(k + ':'):]
        body = body[:body.index('s_endpgm')]
        lines = [l.strip() for l in body.splitlines()]
        lines = [l.split(';')[0].strip() for l in lines if l and not l.startswith(('.', ';'))]
        waits = [i for i, l in enumerate(lines) if l == 's_waitcnt vmcnt(8)']
        assert len(waits) == 1, (k, waits)
        w = waits[0]
        loads = [i for i in range(w) if lines[i].startswith('global_load_dword v')]
        first = loads[-5] if len(loads) >= 5 else None
        assert first is not None and loads[-5:] == list(range(first, first + 5)), (k, 'the five table loads are not consecutive')
        dst = []
        for i in range(first, first + 5):
            m = re.match(r'global_load_dword v(\d+), v\[\d+:\d+\], off', lines[i])
            assert m, (k, lines[i])
            dst.append(int(m.group(1)))
        assert len(set(dst)) == 5
        window = lines[first + 5:w]
        assert sum(l.startswith('global_load_lds_dwordx4') for l in window) == 8, (k, window)
        for l in window:
            assert allowed.match(l), (k, 'unexpected instruction inside the window: ' + l)
            for m in re.finditer(r'v\[(\d+):(\d+)\]', l):          # address pairs of the pieces must not be a destination
                assert not any(int(m.group(1)) <= d <= int(m.group(2)) for d in dst), (k, l, dst)
        assert not re.search(r'scratch_|buffer_(load|store)', body), (k, 'scratch access in the kernel')


def test_split_bf16_tile_kernel_occupancy_pins(tmp_path):
    """Static pins of what the split-bf16 tile kernel's design rests on (csrc/gemm_sb16.hip), read from the gfx950
    assembly of every k_linear_sb instantiation (no GPU):
      * no register spills and no scratch;
      * ONE twelve-wave workgroup per CU: 768 threads and at most 168 registers (three waves per SIMD).  (Round 4: a six-wave form
        with 147 registers was meant to run two workgroups per CU and never did; round 5 removed the four-MFMA-wave forms.)
      * the loader waves' LDS-DMA pieces take the `saddr + voffset` form -- `global_load_lds_dwordx4 vN, s[a:b]` -- so that a
        piece costs them no vector instruction (the compiler's own form carried a v_lshl_add_u64 per piece);
      * the matrix instruction is v_mfma_f32_16x16x32_bf16 and nothing else (round 5: the 32 x 32 x 16 form saves cycles and loses
        them again as clock -- the launches are power-bound, profiles/r05_sb_clock.txt).
    Replaces nn.Linear of gat2.py:53-55 / utils/mlp.py:8-28 on the production path."""
    import shutil
    import subprocess
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('hipcc not available')
    src = os.path.join(ROOT, '3d_multi_pose_estimator_amd', 'csrc', 'gemm_sb16.hip')
    out = str(tmp_path / 'gemm_sb16.s')
    subprocess.run([hipcc, '-O3', '-std=c++17', '--offload-arch=gfx950', '-S', '--cuda-device-only', '-o', out, src],
                   check=True, capture_output=True, timeout=900)
    text = open(out).read()
    meta = {}
    for m in re.finditer(r'\.max_flat_workgroup_size: (\d+)\n\s+\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count: (\d+)', text):
        meta[m.group(2)] = (int(m.group(1)), int(m.group(3)), int(m.group(4)))
    tile = {k: v for k, v in meta.items() if k.startswith('_ZN3mpe2sb11k_linear_sbIL')}
    # f64-sum launches (LEAKY x flush cadence 1 | 2), plain launches (LEAKY), the coefficient launch
    assert len(tile) == 7, sorted(meta)
    for name, (threads, vgpr, spills) in tile.items():
        assert spills == 0, (name, spills)
        assert threads == 768 and vgpr <= 168, (name, threads, vgpr)
        body = text[text.index(name + ':'):]
        body = body[:body.index('s_endpgm')]
        assert not re.search(r'scratch_', body), name
        pieces = re.findall(r'global_load_lds_dwordx4 (\S+), (\S+)', body)
        assert pieces and all(re.fullmatch(r'v\d+,?', a.rstrip(',') + ',') or re.fullmatch(r'v\d+', a.rstrip(',')) for a, _ in pieces), (name, pieces[:3])
        assert all(re.fullmatch(r's\[\d+:\d+\]', b) for _, b in pieces), (name, pieces[:3])
        mfma = set(re.findall(r'(v_mfma_\S+)', body))
        assert mfma == {'v_mfma_f32_16x16x32_bf16'}, (name, mfma)
    assert 'MPE_SBEXP' not in open(src).read()
    # the K-split kernel of the small and mid-size batches (one and two row tiles per workgroup x flush cadence x LeakyReLU): no spills,
    # and few enough registers for FOUR eight-wave workgroups per CU -- its one unit in flight per wave is hidden by the other waves
    # (profiles/r06_ks_ahead_experiment.txt: the run-ahead form with 138 registers was slower)
    ks = {k: v for k, v in meta.items() if k.startswith('_ZN3mpe2sb14k_linear_sb_ksIL')}
    assert len(ks) == 8, sorted(ks)
    for name, (threads, vgpr, spills) in ks.items():
        assert spills == 0 and threads == 512 and vgpr <= 128, (name, threads, vgpr, spills)
        body = text[text.index(name + ':'):]
        body = body[:body.index('s_endpgm')]
        assert not re.search(r'scratch_', body), name
        assert set(re.findall(r'(v_mfma_\S+)', body)) == {'v_mfma_f32_16x16x32_bf16'}, name


def test_latency_gemm_kernels_fit_their_workgroup(tmp_path):
    """Static pins of the small-batch GEMM launches (csrc/lat.hip: k_lat_gemm, fc1 / fc2 of gat2.py:53-55 for <= 16 frames), read from
    the gfx950 assembly (no GPU): every instantiation the host code can launch exists, none spills or touches scratch (the loop form
    WITH the coefficient epilogue spilled 29 registers and was taken out: lat.hip, lat_row_groups), a workgroup's waves fit one CU's
    register file (512 per SIMD lane), and the matrix instruction is the split-bf16 form's v_mfma_f32_16x16x32_bf16."""
    import shutil
    import subprocess
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('hipcc not available')
    src = os.path.join(ROOT, '3d_multi_pose_estimator_amd', 'csrc', 'lat.hip')
    out = str(tmp_path / 'lat.s')
    subprocess.run([hipcc, '-O3', '-std=c++17', '--offload-arch=gfx950', '-S', '--cuda-device-only', '-o', out, src],
                   check=True, capture_output=True, timeout=900)
    text = open(out).read()
    meta = {}
    for m in re.finditer(r'\.max_flat_workgroup_size: (\d+)\n\s+\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count: (\d+)', text):
        meta[m.group(2)] = (int(m.group(1)), int(m.group(3)), int(m.group(4)))
    lat = {k: v for k, v in meta.items() if k.startswith('_ZN3mpe3lat10k_lat_gemmIL')}
    # <NK, NT, LEAKY, OPL, COEF, FUSE2, LOOP>: fc1 400 / 320 wide (single tile and loop forms), fc1 150 wide, fc1 + fc2 of the last layer
    # in one launch, fc2 400 -> H x 40 with coefficients, fc2 320 -> 5 x 30, fc2 150 -> 1 with coefficients
    want = ['Li13ELi5ELb1ELb1ELb0ELb0ELb0E', 'Li13ELi5ELb1ELb1ELb0ELb0ELb1E', 'Li10ELi5ELb1ELb1ELb0ELb0ELb0E', 'Li10ELi5ELb1ELb1ELb0ELb0ELb1E',
            'Li5ELi10ELb1ELb1ELb0ELb0ELb0E', 'Li5ELi10ELb1ELb1ELb0ELb1ELb0E', 'Li13ELi5ELb0ELb0ELb1ELb0ELb0E', 'Li10ELi5ELb0ELb0ELb0ELb0ELb0E',
            'Li5ELi1ELb0ELb0ELb1ELb0ELb0E']
    assert len(lat) == len(want), sorted(lat)
    for w in want:
        assert any(('k_lat_gemmI' + w) in k for k in lat), w
    for name, (threads, vgpr, spills) in lat.items():
        assert spills == 0, (name, spills)
        waves_per_simd = (threads // 64 + 3) // 4
        assert vgpr * waves_per_simd <= 512, (name, threads, vgpr)
        body = text[text.index(name + ':'):]
        body = body[:body.index('s_endpgm')]
        assert not re.search(r'scratch_', body), name
        assert set(re.findall(r'(v_mfma_\S+)', body)) == {'v_mfma_f32_16x16x32_bf16'}, name


def test_no_product_kernel_loads_into_registers_from_inline_asm_without_its_wait():
    """Round 4 lost a GPU box to an ablation build whose register-destination loads were issued by one `asm` statement and waited
    for by another (the compiler, which does not count asm loads, had reused an address register the landing load overwrote).
    Pinned for the library's sources: an `asm` statement that holds a global / buffer / flat / scratch load with a REGISTER
    destination (i) waits for it inside the same statement (`s_waitcnt vmcnt`) and (ii) declares every output early-clobber
    (`=&`), so no destination register is free for reuse while its load is in flight -- the one such statement is k_gat_fused's
    table loads in front of its LDS-DMA pieces; LDS-DMA pieces (`global_load_lds_*`) have no register destination and their wait
    is the caller's explicit `s_waitcnt vmcnt(0)` in front of the stage barrier.  The compile-time ablation switches are gone from
    the product kernels."""
    csrc = os.path.join(ROOT, '3d_multi_pose_estimator_amd', 'csrc')
    bad, seen = [], 0
    for fn in sorted(os.listdir(csrc)):
        if not fn.endswith(('.hip', '.h', '.cpp')):
            continue
        text = open(os.path.join(csrc, fn)).read()
        assert 'MPE_SBEXP' not in text and 'MPE_EXP' not in text, fn
        for m in re.finditer(r'\basm\s*(?:volatile)?\s*\(', text):
            depth, i = 1, m.end()
            while depth and i < len(text):                 # the statement up to its closing parenthesis
                depth += {'(': 1, ')': -1}.get(text[i], 0)
                i += 1
            body = text[m.end():i]
            reg_loads = [ins for ins in re.findall(r'(global_load_\w+|buffer_load_\w+|flat_load_\w+|scratch_load_\w+)', body) if '_lds_' not in ins]
            if not reg_loads:
                continue
            seen += 1
            outputs = re.findall(r'"(=[^"]*)"\s*\(', body)
            if 's_waitcnt vmcnt' not in body or not outputs or not all(o.startswith('=&') for o in outputs):
                bad.append((fn, reg_loads, outputs))
    assert not bad, bad
    assert seen == 1, seen          # k_gat_fused; a new one should be looked at before this number changes


def test_eisel_lemire_against_strtod(tmp_path):
    """csrc/el_double.h (the exact decimal -> binary64 conversion the device-side JSON parser uses) built for the host
    and run against glibc strtod on two million tokens: pixel-coordinate doubles in 17- and 15-digit form, random
    full-range doubles, 19-digit significands with exponents around the table range, near-halfway cases, fixed
    edge cases.  Every token the fast path ACCEPTS must give strtod's bits; declined tokens go to the host."""
    import shutil
    import subprocess
    gxx = shutil.which('g++')
    if not gxx:
        pytest.skip('g++ not available')
    exe = str(tmp_path / 'el_double_test')
    subprocess.run([gxx, '-O2', '-I', os.path.join(ROOT, '3d_multi_pose_estimator_amd', 'csrc'),
                    os.path.join(ROOT, 'tests', 'native', 'el_double_test.cpp'), '-o', exe], check=True, capture_output=True, timeout=300)
    r = subprocess.run([exe, '350000'], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r'tested (\d+) declined (\d+) bad (\d+)', r.stdout)
    assert m and int(m.group(3)) == 0 and int(m.group(1)) > 2000000 and int(m.group(2)) < int(m.group(1)) // 2


def test_lib_leaves_the_hardware_queue_setting_to_the_host_unless_asked():
    """Importing lib.py does not touch the host's environment (round 5: opt-in).  With MPE_SET_HW_QUEUES=1 it sets
    GPU_MAX_HW_QUEUES=8 before HIP initialises (two busy streams of a pipeline on one of HIP's default four hardware queues
    serialise: DESIGN.md 7.2) and never overrides an explicit setting; bench.py sets the variable for its own process."""
    import subprocess
    import sys
    code = ("import os, sys; sys.path.insert(0, %r); import importlib; importlib.import_module('3d_multi_pose_estimator_amd.lib'); "
            "print(os.environ.get('GPU_MAX_HW_QUEUES'))" % ROOT)
    env = {k: v for k, v in os.environ.items() if k not in ('GPU_MAX_HW_QUEUES', 'MPE_SET_HW_QUEUES')}
    assert subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, check=True).stdout.strip() == 'None'
    env['MPE_SET_HW_QUEUES'] = '1'
    assert subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, check=True).stdout.strip() == '8'
    env['GPU_MAX_HW_QUEUES'] = '2'
    assert subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, check=True).stdout.strip() == '2'
    assert "os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')" in open(os.path.join(ROOT, 'bench.py')).read()


def test_build_then_load_in_one_process_maps_one_hip_runtime():
    """Round 3 lost a GPU run to this: build() dlopen-ed libmpe_hip.so for its symbol check BEFORE torch was imported, which
    bound the process to the system's libamdhip64; smoke() in the same process then ran on two HIP runtimes and mpe_create
    failed with -4.  build() now checks the symbols in a child and lib.load() imports torch first and refuses to go on
    with two runtimes mapped.  CPU-runnable: the maps of the process tell."""
    import subprocess
    import sys
    code = (
        "import sys, importlib; sys.path.insert(0, %r)\n"
        "import __graft_entry__ as g\n"
        "g.build()\n"
        "maps = open('/proc/self/maps').read()\n"
        "assert 'libmpe_hip' not in maps and 'libamdhip64' not in maps, 'build() mapped a HIP library into its own process'\n"
        "lib = importlib.import_module('3d_multi_pose_estimator_amd.lib')\n"
        "lib.load()\n"
        "rts = lib.hip_runtimes_mapped()\n"
        "assert len(rts) == 1, rts\n"
        "import torch\n"
        "assert len(lib.hip_runtimes_mapped()) == 1, lib.hip_runtimes_mapped()\n"
        "print('ok', rts[0])\n" % ROOT)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and 'ok ' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    # the failure mode itself: the system runtime first, then the package -> a clear ImportError, not -4 later
    sys_hip = '/opt/rocm/lib/libamdhip64.so'
    if os.path.exists(sys_hip):
        code2 = ("import ctypes, sys, importlib; sys.path.insert(0, %r); ctypes.CDLL(%r)\n"
                 "lib = importlib.import_module('3d_multi_pose_estimator_amd.lib')\n"
                 "try:\n    lib.load()\nexcept ImportError as e:\n    print('refused:', e); sys.exit(0)\n"
                 "print('runtimes', lib.hip_runtimes_mapped()); sys.exit(0 if len(lib.hip_runtimes_mapped()) == 1 else 3)" % (ROOT, sys_hip))
        r2 = subprocess.run([sys.executable, '-c', code2], capture_output=True, text=True, timeout=300)
        assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr[-2000:]


def test_build_stamp_forces_a_rebuild_when_sources_change(tmp_path, monkeypatch):
    """build() must not trust objects that merely travel with the tree: a source hash that differs from the stamp (or no
    stamp) turns the call into `make -B`.  The make itself is stubbed out here; the decision is what is tested."""
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    g = importlib.import_module('__graft_entry__')
    calls = []
    monkeypatch.setattr(g.subprocess, 'check_call', lambda cmd, **kw: calls.append(list(cmd)))
    csrc = os.path.join(ROOT, '3d_multi_pose_estimator_amd', 'csrc')
    stamp, mode = os.path.join(csrc, '.build_stamp'), os.path.join(csrc, '.build_mode')
    keep = {p: (open(p).read() if os.path.exists(p) else None) for p in (stamp, mode)}
    try:
        with open(stamp, 'w') as fh:
            fh.write('0000000000000000\n')
        g.build()
        assert '-B' in calls[0] and open(mode).read().startswith('clean ')
        assert open(stamp).read().strip() == g.source_hash()
        calls.clear()
        if os.path.exists(os.path.join(ROOT, '3d_multi_pose_estimator_amd', 'libmpe_hip.so')):
            g.build()
            assert '-B' not in calls[0] and open(mode).read().startswith('reused ')
        calls.clear()
        monkeypatch.setattr(g, 'source_hash', lambda: 'feedfacefeedface')
        g.build()
        assert '-B' in calls[0]
    finally:
        for p, v in keep.items():
            if v is None:
                if os.path.exists(p):
                    os.remove(p)
            else:
                with open(p, 'w') as fh:
                    fh.write(v)


def _generated_dataset(tmp_path, monkeypatch, **kw):
    import random

    from conftest import generated_fixture
    exp, arr, files, probs = generated_fixture()
    gg = pkg('graph_generator')
    monkeypatch.chdir(tmp_path)                     # the dataset cache goes to ./cache/ (reference :520, 884-917)
    random.seed(exp['seed'])
    ds = gg.MergedMultipleHumansDataset(files, probs, limit=1000, mode='test_generated', alt='3', raw_dir='.', **kw)
    return gg, ds, exp, arr


def test_generated_dataset_mirror_builds_the_reference_graphs(tmp_path, monkeypatch):
    """MergedMultipleHumansDataset(files, probabilities, limit, mode='test_generated') of the package, seeded like
    the reference run, composes the reference's scenes: per graph the same edge list (one edge-node per ORDERED head pair,
    heads grouped by person), labels, edge_nodes_indices and nodes_camera as /root/reference built (tests/golden/generated/).
    Host logic only -- no GPU call."""
    gg, ds, exp, arr = _generated_dataset(tmp_path, monkeypatch)
    assert len(ds) == exp['n_graphs']
    for i in range(len(ds)):
        g, labels, idx, nodes_camera = ds[i]
        src, dst = g.edges()
        assert np.array_equal(src.numpy(), arr['src_%d' % i]) and np.array_equal(dst.numpy(), arr['dst_%d' % i])
        assert src.dtype == torch.int32 and labels.dtype == torch.float64 and idx.dtype == torch.int64
        assert labels.shape == (g.M, 1) and idx.shape == (g.M, 1)
        assert np.array_equal(labels.numpy().reshape(-1), arr['labels_%d' % i])
        assert np.array_equal(idx.numpy().reshape(-1), arr['indices_%d' % i])
        assert nodes_camera == exp['graphs'][i]['nodes_camera']
        assert g.number_of_nodes() == exp['graphs'][i]['N'] and len(g.nodes()) == exp['graphs'][i]['N']
        assert np.array_equal(g.packed.head_cam, arr['head_cam_%d' % i])
    # batch() = dgl.batch: nodes and edges relabelled graph by graph
    b = gg.batch([ds[i][0] for i in range(3)])
    src, dst = b.edges()
    off = 0
    want_s, want_d = [], []
    for i in range(3):
        want_s.append(arr['src_%d' % i] + off)
        want_d.append(arr['dst_%d' % i] + off)
        off += exp['graphs'][i]['N']
    assert np.array_equal(src.numpy(), np.concatenate(want_s)) and np.array_equal(dst.numpy(), np.concatenate(want_d))
    assert b.number_of_nodes() == off and b.batch_size == 3 and b.batch_num_nodes().tolist() == [exp['graphs'][i]['N'] for i in range(3)]
    # the cache of the reference (graph_generator.py:884-917): a second construction loads it instead of sampling again
    assert os.path.exists(tmp_path / 'cache' / 'MergedMultipleHumansDataset_test_generated_alt_3_s_1000.bin')
    ds2 = gg.MergedMultipleHumansDataset(_files_of(exp), [1.0] * 4, limit=1000, mode='test_generated', alt='3')
    assert len(ds2) == len(ds)
    for i in range(len(ds)):
        assert np.array_equal(ds2[i][0].edges()[0].numpy(), arr['src_%d' % i]) and ds2[i][3] == ds[i][3]
    # debug=True neither reads nor writes it; the training modes are out of scope and say so
    with pytest.raises(NotImplementedError):
        gg.MergedMultipleHumansDataset(_files_of(exp), [1.0] * 4, limit=10, mode='train', alt='3')
    with pytest.raises(NotImplementedError):
        gg.MergedMultipleHumansDataset(_files_of(exp), [1.0] * 4, limit=10, mode='test_generated', alt='2')


def _files_of(exp):
    return [os.path.join(ROOT, 'tests', 'golden', 'generated', f) for f in exp['files']]


def test_generated_scene_order_rules():
    """packing.generated_scene on a hand-made scene: person blocks before the spurious block, ordered pairs, same-camera pairs
    skipped, first-of-the-largest as the person's skeleton (graph_generator.py:718-797)."""
    packing = pkg('packing')
    par = pkg('parameters').parameters

    def sk(n):
        return {str(j): [j, 10.0 * j, 5.0, 1, 1] for j in range(n)}
    a, b, c = par.used_cameras_skeleton_matching[:3]
    views = [{a: [json.dumps([sk(3), sk(5), sk(5)])], b: [json.dumps([sk(4)])], 'not_a_camera': [json.dumps([sk(2)])]},
             {b: [json.dumps([{'ID': 1}, sk(2)])], c: [json.dumps([])]}]
    sc = packing.generated_scene(views, par)
    assert [(h[0], h[1]) for h in sc['heads']] == [(a, 0), (a, 1), (a, 2), (b, 0), (b, 1)]
    # person 0 = heads {1 (first of the two 5-joint skeletons), 3}; person 1 = {4}; spurious = {0, 2}
    want = [(1, 3), (3, 1),                                  # own x own
            (1, 4),                                          # person 0 x person 1 (3-4 share camera b)
            (3, 0), (3, 2),                                  # person 0 x spurious (1-0, 1-2 share camera a)
            (4, 1),                                          # person 1 x person 0
            (4, 0), (4, 2)]                                  # person 1 x spurious; spurious x spurious: both on camera a
    assert sc['pairs'].tolist() == [list(p) for p in want]
    assert sc['labels'].tolist() == [1.0, 1.0, 0, 0, 0, 0, 0, 0]
    pb = packing.pack_scenes([sc], par)
    assert pb.n_heads == 5 and pb.n_edge_nodes == 8 and pb.en_pair.shape == (8, 2) and pb.head_cam.tolist() == [0, 0, 0, 1, 1]


def test_native_packer_accepts_the_json_dialect_of_python(calib):
    """The host packer (csrc/packer.cpp) and json.loads + the Python packer on JSON that json.dumps never writes but json.loads
    accepts: a camera key spelled with escapes ("tracker\\u0061" IS trackera -- it used to be dropped silently as an unknown
    camera: different arrays, no error), a pretty-printed inner skeleton list (its line breaks arrive as \\n escapes),
    Infinity / -Infinity / NaN, null where a number may stand (numpy stores None as NaN).  Same arrays or both refuse; and the
    first-level walk of the device path (mpe_json_stage_window) finds the escaped camera as well."""
    packing = pkg('packing')
    P = calib.params
    sk = [{"0": [0, 100.5, 200.25, 1, 0.9], "5": [5, 10, 20, 0.5, 0.25]}]

    def frame(inner, ts='0.0', extra='"no_image", []'):
        return '{"trackera": [%s, %s, %s], "trackerb": [%s, 1.0]}' % (json.dumps(inner), ts, extra, json.dumps(inner))
    canon = json.dumps(sk)
    docs = {
        'unicode key': frame(canon).replace('"trackera"', '"tracker\\u0061"'),
        'unicode key upper hex': frame(canon).replace('"trackerb"', '"\\u0074racker\\u0062"'),
        'escaped slash in an unknown key': frame(canon).replace('"trackera"', '"tracker\\/a"'),
        'surrogate pair key': frame(canon).replace('"trackera"', '"\\ud83d\\ude00"'),
        'pretty inner': frame(json.dumps(sk, indent=1)),
        'pretty inner, tabs': frame(json.dumps(sk, indent='\t')),
        'Infinity': frame(canon.replace('0.9', 'Infinity')),
        '-Infinity': frame(canon.replace('100.5', '-Infinity')),
        'NaN': frame(canon.replace('0.9', 'NaN')),
        'null number': frame(canon.replace('0.9', 'null')),
        'null timestamp and bodies': frame(canon, ts='null', extra='"no_image", null'),
        'null joint id': frame(canon.replace('[0, 100.5', '[null, 100.5')),
        'bad escape in a key': frame(canon).replace('"trackera"', '"tracker\\qa"'),
    }
    n_same = n_refused = 0
    for name, body in docs.items():
        doc = '[' + body + ']'
        try:
            py = packing.pack_frames(json.loads(doc), P)
        except Exception:
            py = None
        if py is None:
            with pytest.raises(ValueError):
                packing.pack_json(doc, P)
            n_refused += 1
            continue
        nat = packing.pack_json(doc, P)
        for f in ('frame_head_off', 'frame_en_off', 'slot_cam', 'slot_n', 'head_cam', 'skeleton_index', 'joint_mask', 'tri_mask', 'xy', 'vp'):
            a, b = np.asarray(getattr(py, f)), np.asarray(getattr(nat, f))
            assert a.shape == b.shape and np.array_equal(a, b, equal_nan=a.dtype.kind == 'f'), (name, f)
        n_same += 1
    assert n_same == 11 and n_refused == 2
    # first level of the device path: both cameras found although one key is spelled with escapes
    doc = ('[' + docs['unicode key'] + ']').encode()
    ix = packing.JsonIndex(doc)
    try:
        st = packing.JsonStage(len(P.used_cameras_skeleton_matching), 4, 1 << 16, 'host')
        nf, ne, used = packing.stage_json_window(ix, P, st, max_frames=4)
        assert (nf, ne) == (1, 2)
    finally:
        ix.close()
