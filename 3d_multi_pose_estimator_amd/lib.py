"""ctypes binding of libmpe_hip.so (include/mpe.h).

The HIP library is the product: there is no CPU fallback.  Importing this module works
anywhere (so that host-side logic can be tested without a GPU); *using* it without the
built library raises immediately.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libmpe_hip.so')
if os.environ.get('MPE_LIB_VARIANT'):      # diagnostics: an experiment build beside the product library (csrc/Makefile `exp`)
    LIB_PATH = os.path.join(_HERE, 'libmpe_hip_%s.so' % os.environ['MPE_LIB_VARIANT'])

# HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The engine's pipelines keep 3-4 streams busy
# (parse / copy / matching / 3D) beside whatever the application has; two of them on one queue serialise (pipeline.py:
# _make_json_streams has the measurement), so a process that streams JSON or pipelines its copies should run with
# GPU_MAX_HW_QUEUES=8 -- set before HIP initialises.  A library does not rewrite its host's environment on import: since round 5
# this is OPT-IN (MPE_SET_HW_QUEUES=1 makes the import set it when it is unset); bench.py sets the variable itself.
if os.environ.get('MPE_SET_HW_QUEUES', '0') == '1':
    os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

MPE_MAX_CAMERAS = 32
MPE_MAX_JOINTS = 32
MPE_ERR_UNSUPPORTED = -6

c_f32p = C.POINTER(C.c_float)
c_f64p = C.POINTER(C.c_double)
c_i32p = C.POINTER(C.c_int32)
c_u32p = C.POINTER(C.c_uint32)
c_u8p = C.POINTER(C.c_uint8)


class MpeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__('libmpe_hip error %d: %s' % (code, msg))
        self.code = code


class mpe_config(C.Structure):
    _fields_ = [
        ('n_cameras', C.c_int32), ('n_joints', C.c_int32), ('image_width', C.c_int32),
        ('image_height', C.c_int32), ('numbers_per_joint', C.c_int32), ('min_views', C.c_int32),
        ('median_axis', C.c_int32), ('used_joint_mask', C.c_uint32), ('threshold', C.c_float),
        ('median_window', C.c_float), ('max_frames', C.c_int32), ('max_heads', C.c_int32),
        ('max_edge_nodes', C.c_int32), ('max_heads_per_frame', C.c_int32),
        ('max_persons_per_frame', C.c_int32),
        ('Kinv', c_f32p), ('K', c_f32p), ('T_i', c_f32p), ('P', c_f64p), ('dist', c_f64p),
    ]


class mpe_batch(C.Structure):
    _fields_ = [
        ('n_frames', C.c_int32), ('n_heads', C.c_int32), ('n_edge_nodes', C.c_int32),
        ('d_frame_head_off', C.c_void_p), ('d_frame_en_off', C.c_void_p), ('d_slot_cam', C.c_void_p),
        ('d_slot_n', C.c_void_p), ('d_head_cam', C.c_void_p), ('d_joint_mask', C.c_void_p),
        ('d_tri_mask', C.c_void_p), ('d_xy', C.c_void_p), ('d_vp', C.c_void_p),
        ('d_en_pair', C.c_void_p),      # optional explicit edge-node list (process_training topology); NULL = implicit
    ]


class mpe_packed_arrays(C.Structure):
    _fields_ = [('n_frames', C.c_int32), ('n_heads', C.c_int32), ('n_edge_nodes', C.c_int32),
                ('n_cameras', C.c_int32), ('n_joints', C.c_int32),
                ('frame_head_off', c_i32p), ('frame_en_off', c_i32p), ('slot_cam', c_i32p), ('slot_n', c_i32p),
                ('head_cam', c_i32p), ('skeleton_index', c_i32p), ('joint_mask', c_u32p), ('tri_mask', c_u32p),
                ('xy', c_f64p), ('vp', c_f32p)]


class mpe_pack_dst(C.Structure):
    _fields_ = [('max_frames', C.c_int32), ('max_heads', C.c_int32),
                ('frame_head_off', C.c_void_p), ('frame_en_off', C.c_void_p), ('slot_cam', C.c_void_p), ('slot_n', C.c_void_p),
                ('head_cam', C.c_void_p), ('skeleton_index', C.c_void_p), ('joint_mask', C.c_void_p), ('tri_mask', C.c_void_p),
                ('xy', C.c_void_p), ('vp', C.c_void_p)]


# name -> (restype, argtypes); must list every symbol include/mpe.h declares
SYMBOLS = {
    'mpe_create': (C.c_int, [C.POINTER(mpe_config), C.POINTER(C.c_void_p)]),
    'mpe_destroy': (None, [C.c_void_p]),
    'mpe_last_error': (C.c_char_p, [C.c_void_p]),
    'mpe_version': (C.c_char_p, []),
    'mpe_set_gat_params': (C.c_int, [C.c_void_p, C.c_int32, C.c_float, C.c_float]),
    'mpe_set_gat_layer': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                    c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p]),
    'mpe_set_mlp_params': (C.c_int, [C.c_void_p, C.c_int32, C.c_float]),
    'mpe_set_mlp_layer': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, c_f32p, c_f32p]),
    'mpe_set_precision': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    'mpe_match_batch': (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(mpe_batch), C.c_void_p, C.c_void_p, C.c_void_p]),
    'mpe_mlp3d_batch': (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(mpe_batch), C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p]),
    'mpe_triangulate_batch': (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(mpe_batch), C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_uint32]),
    'mpe_upload_linear': (C.c_int, [C.c_void_p, c_f32p, c_f32p, C.c_int32, C.c_int32,
                                    C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int32)]),
    'mpe_free_device': (C.c_int, [C.c_void_p, C.c_void_p]),
    'mpe_linear': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                             C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                             C.c_float]),
    'mpe_head_features': (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(mpe_batch), C.c_void_p]),
    'mpe_dense_rows': (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(mpe_batch), C.c_void_p, C.c_int32]),
    'mpe_gat_forward': (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(mpe_batch), C.c_void_p, C.c_int32, C.c_void_p,
                                  C.c_void_p]),
    'mpe_set_gat_output': (C.c_int, [C.c_void_p, C.c_int32]),
    'mpe_gat_layer': (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(mpe_batch), C.c_int32, C.c_void_p, C.c_int32,
                                C.c_void_p, C.c_int32, C.c_int32]),
    'mpe_edge_softmax_aggregate': (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(mpe_batch), C.c_int32, C.c_void_p,
                                             C.c_int32, C.c_void_p, C.c_int32]),
    'mpe_sync_status': (C.c_int, [C.c_void_p, C.c_void_p]),
    'mpe_status_queue': (C.c_int, [C.c_void_p, C.c_void_p]),
    'mpe_status_wait': (C.c_int, [C.c_void_p, C.c_void_p]),
    'mpe_set_threshold': (C.c_int, [C.c_void_p, C.c_float]),
    'mpe_cluster_batch': (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(mpe_batch), C.c_void_p, C.c_void_p,
                                    C.c_void_p]),
    'mpe_mlp_input_rows': (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(mpe_batch), C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_int32, C.c_void_p]),
    'mpe_mlp_forward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32]),
    'mpe_dlt_pairs': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    'mpe_pack_json': (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(C.c_char_p), C.c_int32, C.c_int32, C.c_int32,
                                C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    'mpe_packed_view': (C.c_int, [C.c_void_p, C.POINTER(mpe_packed_arrays)]),
    'mpe_pack_json_into': (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(C.c_char_p), C.c_int32, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_int32, C.c_int32, C.POINTER(mpe_pack_dst), C.POINTER(C.c_int32),
                                     C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    'mpe_pack_views_into': (C.c_int, [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_int32,
                                      C.POINTER(mpe_pack_dst), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_void_p]),
    'mpe_json_index_create': (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    'mpe_json_index_free': (None, [C.c_void_p]),
    'mpe_pack_indexed_into': (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                        C.c_int32, C.c_int32, C.POINTER(mpe_pack_dst), C.POINTER(C.c_int32),
                                        C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    'mpe_json_stage_window': (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                        C.c_void_p, C.c_size_t, C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(C.c_int32),
                                        C.POINTER(C.c_int32), C.POINTER(C.c_size_t)]),
    'mpe_json_scratch_bytes': (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    'mpe_json_parse_device': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                        C.c_int32, C.c_int32, C.c_void_p, C.c_size_t, C.POINTER(mpe_batch), C.c_void_p, C.c_void_p]),
    'mpe_packed_free': (None, [C.c_void_p]),
    'mpe_pack_last_error': (C.c_char_p, []),
    'mpe_profile_enable': (C.c_int, [C.c_void_p, C.c_int32]),
    'mpe_profile_read': (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                   C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    'mpe_profile_read_split': (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    'mpe_profile_read_bf16': (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
}

_lib = None


def hip_runtimes_mapped():
    """Real paths of every libamdhip64 mapped into this process (Linux: /proc/self/maps)."""
    paths = set()
    try:
        with open('/proc/self/maps') as fh:
            for line in fh:
                if 'libamdhip64' in line:
                    paths.add(os.path.realpath(line.split()[-1]))
    except OSError:
        pass
    return sorted(paths)


def load():
    """Return the loaded library; raise if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError('%s is missing: build it with `python __graft_entry__.py` or '
                          '`make -C 3d_multi_pose_estimator_amd/csrc` (hipcc, gfx950). '
                          'There is no CPU fallback.' % LIB_PATH)
    # torch first, if it is there: libmpe_hip.so needs libamdhip64, and a process that already has the system's copy loaded
    # when torch later brings its own talks to two HIP runtimes at once (device tensors of one, kernels of the other)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    rts = hip_runtimes_mapped()
    if len(rts) > 1:
        # device memory of one runtime, kernels of the other: every later call would fail with an opaque MPE_ERR_HIP (-4)
        raise ImportError('two HIP runtimes are mapped into this process (%s): something loaded a libamdhip64 before torch '
                          'brought its own.  Import torch (or this package) BEFORE anything that dlopens the system HIP '
                          'runtime, e.g. before ctypes.CDLL(%r).' % (', '.join(rts), LIB_PATH))
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(ctx, rc):
    if rc != 0:
        msg = load().mpe_last_error(ctx)
        raise MpeError(rc, msg.decode() if msg else '?')
