"""The C ABI of include/mpe.h called by a plain C++ program (tests/native/abi_roundtrip.cpp: no Python, no torch, its own HIP
stream and hipMalloc'ed buffers) against the Python binding on the same input: the drop-in boundary is the LIBRARY.  The program
replaces the body of the reference's per-frame loop (test/metrics_from_model.py:120-300, test/metrics_from_triangulation.py:187-272)
with mpe_pack_json -> mpe_match_batch -> mpe_mlp3d_batch / mpe_triangulate_batch, exactly what INTEGRATION.md tells a C / C++ host
to do."""
import json
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT, pkg

pytestmark = pytest.mark.gpu

_DT = {np.dtype(np.uint8): 0, np.dtype(np.int32): 1, np.dtype(np.float32): 2, np.dtype(np.float64): 3}
_NP = {0: np.uint8, 1: np.int32, 2: np.float32, 3: np.float64}


def _put(fh, name, arr):
    arr = np.ascontiguousarray(arr)
    fh.write(struct.pack('<I', len(name)) + name.encode() + struct.pack('<BI', _DT[arr.dtype], arr.ndim))
    fh.write(struct.pack('<%dQ' % arr.ndim, *arr.shape))
    fh.write(arr.tobytes())


def _read(path):
    out, raw, i = {}, open(path, 'rb').read(), 0
    while i < len(raw):
        nl, = struct.unpack_from('<I', raw, i)
        name = raw[i + 4:i + 4 + nl].decode()
        i += 4 + nl
        dt, nd = struct.unpack_from('<BI', raw, i)
        i += 5
        dims = struct.unpack_from('<%dQ' % nd, raw, i)
        i += 8 * nd
        n = int(np.prod(dims)) if nd else 1
        out[name] = np.frombuffer(raw, _NP[dt], n, i).reshape(dims).copy()
        i += n * np.dtype(_NP[dt]).itemsize
    return out


def _build(tmp_path):
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('hipcc not available')
    exe = str(tmp_path / 'abi_roundtrip')
    libdir = os.path.join(ROOT, '3d_multi_pose_estimator_amd')
    subprocess.run([hipcc, '-O2', '-std=c++17', '-I', os.path.join(ROOT, 'include'), os.path.join(ROOT, 'tests', 'native', 'abi_roundtrip.cpp'),
                    '-L', libdir, '-lmpe_hip', '-Wl,-rpath,' + libdir, '-o', exe], check=True, capture_output=True, timeout=600)
    return exe


def _write_case(path, eng, gat_weights, mlp_weights, text):
    sd, prm = gat_weights
    p = eng.params
    with open(path, 'wb') as fh:
        # the scalar fields of mpe_config in declaration order, as the Engine gave them to mpe_create (pipeline.py: Engine.__init__)
        _put(fh, 'cfg', np.array([eng.V, eng.J, p.image_width, p.image_height, p.numbers_per_joint, p.min_number_of_views, p.axes_3D['Y'][0],
                                  sum(1 << j for j in p.used_joints), eng._made_with['threshold'], np.float32(0.05), eng.max_frames,
                                  eng.max_frames * eng.hpf, eng.max_frames * eng.m_frame, eng.hpf, eng.pcap], np.float64))
        for k in ('Kinv', 'K', 'T_i', 'P', 'dist'):
            _put(fh, k, eng._keep[k])
        slope = prm.get('nonlinearity', 0.01)
        slope = getattr(slope, 'negative_slope', slope)
        n_layers = int(prm['gnn_layers'])
        heads = list(prm['heads']) + [1]
        _put(fh, 'gat_params', np.array([n_layers, np.float32(prm['alpha']), np.float32(slope)], np.float64))
        for l in range(n_layers):
            w1 = np.asarray(sd['layers.%d.fc1.weight' % l], np.float32)
            w2 = np.asarray(sd['layers.%d.fc2.weight' % l], np.float32)
            nh = heads[l]
            _put(fh, 'gat%d_dims' % l, np.array([w1.shape[1], nh, w2.shape[0] // nh], np.int32))
            _put(fh, 'gat%d_fc1_w' % l, w1)
            _put(fh, 'gat%d_fc1_b' % l, np.asarray(sd['layers.%d.fc1.bias' % l], np.float32))
            _put(fh, 'gat%d_fc2_w' % l, w2)
            _put(fh, 'gat%d_fc2_b' % l, np.asarray(sd['layers.%d.fc2.bias' % l], np.float32))
            _put(fh, 'gat%d_attn_l' % l, np.asarray(sd['layers.%d.attn_l' % l], np.float32).reshape(nh, -1))
            _put(fh, 'gat%d_attn_r' % l, np.asarray(sd['layers.%d.attn_r' % l], np.float32).reshape(nh, -1))
        keys = sorted({int(k.split('.')[1]) for k in mlp_weights})
        _put(fh, 'mlp_params', np.array([len(keys), np.float32(0.1)], np.float64))
        for n, k in enumerate(keys):
            _put(fh, 'mlp%d_w' % n, np.asarray(mlp_weights['layers.%d.weight' % k], np.float32))
            _put(fh, 'mlp%d_b' % n, np.asarray(mlp_weights['layers.%d.bias' % k], np.float32))
        _put(fh, 'cameras', np.frombuffer('\n'.join(p.used_cameras_skeleton_matching).encode(), np.uint8))
        _put(fh, 'json', np.frombuffer(text, np.uint8))


@pytest.mark.parametrize('n_frames,parse', [(1, 'host'), (3, 'host'), (40, 'host'), (3, 'device'), (40, 'device')])
def test_cpp_host_gets_the_python_bindings_bits(tmp_path, calib, gat_weights, mlp_weights, n_frames, parse):
    """1 and 3 frames (the small-batch launches), 40 (the batch kernels with the K-split MLP): scores, persons, MLP poses and
    triangulated poses of the C++ host equal the Python binding's, bit for bit; frames with 1 ... 5 persons, one with an empty camera.
    parse = 'device': the program stages the document's first level on the host and has the skeleton strings parsed on the GPU
    (mpe_json_stage_window -> mpe_json_parse_device) instead of packing it with mpe_pack_json."""
    syn = pkg('synthetic')
    eng = pkg('pipeline').Engine(calib.params, calib, max_frames=48, max_persons_per_camera=6)
    try:
        sd, prm = gat_weights
        eng.load_gat(sd, prm)
        eng.load_mlp(mlp_weights)
        cams = list(calib.params.used_cameras_skeleton_matching)
        frames = [syn.make_frame(calib, 4200 + i, syn.FrameSpec(persons=1 + i % 5, empty_cameras=(cams[1],) if i == 2 else (),
                                                                joint_drop=0.1 * (i % 3)))[0] for i in range(n_frames)]
        text = json.dumps(frames).encode()
        db = eng.to_device(eng.pack_json(text))
        sc, pe, npers = eng.match(db)
        po, va = eng.mlp3d(db, pe, npers)
        tp, tv = eng.triangulate(db, pe, npers)
        eng.sync_status()
        want = dict(counts=np.array([db.n_frames, db.n_heads, db.n_edge_nodes], np.int32), scores=sc.cpu().numpy(), persons=pe.cpu().numpy(),
                    n_persons=npers.cpu().numpy(), poses=po.cpu().numpy(), valid=va.cpu().numpy(), tri_poses=tp.cpu().numpy(),
                    tri_valid=tv.cpu().numpy())
        case, res = str(tmp_path / 'case.bin'), str(tmp_path / 'result.bin')
        _write_case(case, eng, gat_weights, mlp_weights, text)
    finally:
        eng.close()
    exe = _build(tmp_path)
    env = {k: v for k, v in os.environ.items() if not k.startswith('MPE_')}
    # (one and three frames: 300 further timed calls -- the reference's call pattern from a native host; the line goes to
    # gpurun_out/native_host_latency.txt, nothing is asserted about it)
    timed = n_frames <= 3 and parse == 'host'
    r = subprocess.run([exe, case, res, '300' if timed else '0'] + (['device'] if parse == 'device' else []), capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode == 0 and ('(%s parse)' % parse) in r.stdout, r.stdout + r.stderr
    if timed:
        os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
        with open(os.path.join(ROOT, 'gpurun_out', 'native_host_latency.txt'), 'a') as fh:
            fh.write(''.join(l + '\n' for l in r.stdout.splitlines() if 'us per call' in l))
    got = _read(res)
    assert np.array_equal(got['counts'], want['counts'])
    assert int(want['n_persons'].sum()) >= n_frames
    assert np.array_equal(got['scores'], want['scores'])
    assert np.array_equal(got['n_persons'], want['n_persons'])
    for f in range(n_frames):
        k = int(want['n_persons'][f])
        assert np.array_equal(got['persons'][f, :k], want['persons'][f, :k]), f
        assert np.array_equal(got['valid'][f, :k], want['valid'][f, :k]), f
        keep = want['valid'][f, :k] != 0
        assert np.array_equal(got['poses'][f, :k][keep], want['poses'][f, :k][keep]), f
        assert np.array_equal(got['tri_valid'][f, :k], want['tri_valid'][f, :k]), f
        jv = want['tri_valid'][f, :k] != 0
        assert np.array_equal(got['tri_poses'][f, :k][jv], want['tri_poses'][f, :k][jv]), f
