"""Frame JSON -> packed structure-of-arrays batch (the `mpe_batch` of include/mpe.h).

Host-side counterpart of MergedMultipleHumansDataset.load_people_view_graph (reference
skeleton_matching/graph_generator.py:573-605): it only *orders* the data; every number the
network sees is computed on the device from these arrays.

Ordering rules kept from the reference:
  * cameras in the frame dict's key order (not `parameters` order), skipping cameras that
    are not in ``used_cameras_skeleton_matching`` (:583-584);
  * skeletons in list order, skipping those without any joint key other than "ID"
    (:590-591); ``skeleton_index`` remembers the position in the original list (:597-599);
  * joint key j is the COCO joint id string; values = [id, x, y, valid, prob].
"""
import json

import numpy as np


class PackedBatch:
    """Host (numpy) form; `.to(device)` gives torch tensors + the ctypes struct."""

    def __init__(self, V, J):
        self.V, self.J = V, J
        self.n_frames = 0
        self.frame_head_off = None
        self.frame_en_off = None
        self.slot_cam = None
        self.slot_n = None
        self.head_cam = None
        self.joint_mask = None
        self.tri_mask = None
        self.xy = None
        self.vp = None
        self.skeleton_index = None     # [n_heads] position in the camera's original list
        self.jsons_for_head = None     # optional: list (per frame) of {head id: skeleton dict}
        self.en_pair = None            # optional [n_edge_nodes][2] int32: EXPLICIT edge-node list (mpe_batch::d_en_pair)

    @property
    def n_heads(self):
        return int(self.frame_head_off[-1])

    @property
    def n_edge_nodes(self):
        return int(self.frame_en_off[-1])

    def frame_counts(self, f):
        h0, h1 = int(self.frame_head_off[f]), int(self.frame_head_off[f + 1])
        e0, e1 = int(self.frame_en_off[f]), int(self.frame_en_off[f + 1])
        return h0, h1 - h0, e0, e1 - e0

    def max_heads_per_frame(self):
        return int(np.max(np.diff(self.frame_head_off))) if self.n_frames else 0

    def max_edge_nodes_per_frame(self):
        return int(np.max(np.diff(self.frame_en_off))) if self.n_frames else 0

    def pairs(self, f):
        """(h1, h2) of every edge-node of frame f, in edge-node order."""
        if self.en_pair is not None:
            return self.en_pair[int(self.frame_en_off[f]):int(self.frame_en_off[f + 1])]
        return pairs_of_frame(self.slot_n[f])

    def to(self, device):
        return DeviceBatch(self, device)


ARRAYS = (('frame_head_off', np.int32), ('frame_en_off', np.int32), ('slot_cam', np.int32), ('slot_n', np.int32),
          ('head_cam', np.int32), ('joint_mask', np.uint32), ('tri_mask', np.uint32), ('xy', np.float64),
          ('vp', np.float32))


class BatchArena:
    """The nine arrays of one packed batch as views of ONE contiguous byte buffer (256-byte
    aligned pieces), so that a batch travels host -> device as a single copy from pinned memory
    (8.6 KB per 5x4 frame) instead of nine.  `where` = 'pinned' (host, page-locked), 'host' (pageable) or a device."""

    def __init__(self, pb, where):
        import torch
        self.offsets = {}
        off = 0
        for name, dt in ARRAYS:
            n = int(np.asarray(getattr(pb, name)).size)
            self.offsets[name] = (off, n, dt)
            off += (n * np.dtype(dt).itemsize + 255) // 256 * 256
        self.nbytes = max(off, 256)
        if where == 'pinned':
            self.buf = torch.empty(self.nbytes, dtype=torch.uint8).pin_memory()
        elif where == 'host':                      # pageable (layout tests without a GPU)
            self.buf = torch.empty(self.nbytes, dtype=torch.uint8)
        else:
            self.buf = torch.empty(self.nbytes, dtype=torch.uint8, device=where)

    def fill(self, pb):
        """Host arenas only: copy the batch's arrays into the pinned buffer."""
        host = self.buf.numpy()
        for name, (off, n, dt) in self.offsets.items():
            src = np.ascontiguousarray(getattr(pb, name)).reshape(-1).view(dt) if n else np.zeros(0, dt)
            host[off: off + n * np.dtype(dt).itemsize].view(dt)[:] = src
        return self

    def ptr(self, name):
        return self.buf.data_ptr() + self.offsets[name][0]


class CapacityArena:
    """One contiguous buffer laid out for CAPACITIES (max_frames, max_heads) instead of one batch's
    exact sizes: the native packer (`mpe_pack_json_into`) parses straight into a page-locked one,
    its device twin receives every batch in a single copy, and the `mpe_batch` pointers never
    change.  `where` = 'pinned', 'host' or a device."""

    FIELDS = (('frame_head_off', np.int32, 'F1'), ('frame_en_off', np.int32, 'F1'), ('slot_cam', np.int32, 'FV'),
              ('slot_n', np.int32, 'FV'), ('head_cam', np.int32, 'H'), ('skeleton_index', np.int32, 'H'),
              ('joint_mask', np.uint32, 'H'), ('tri_mask', np.uint32, 'H'), ('xy', np.float64, 'HJ2'), ('vp', np.float32, 'HJ2'))

    def __init__(self, V, J, max_frames, max_heads, where):
        import torch
        self.V, self.J, self.max_frames, self.max_heads = V, J, int(max_frames), int(max_heads)
        count = {'F1': max_frames + 1, 'FV': max_frames * V, 'H': max_heads, 'HJ2': max_heads * J * 2}
        self.offsets = {}
        off = 0
        for name, dt, kind in self.FIELDS:
            n = int(count[kind])
            self.offsets[name] = (off, n, dt)
            off += (n * np.dtype(dt).itemsize + 255) // 256 * 256
        self.nbytes = off
        if where == 'pinned':
            self.buf = torch.empty(self.nbytes, dtype=torch.uint8).pin_memory()
        elif where == 'host':
            self.buf = torch.empty(self.nbytes, dtype=torch.uint8)
        else:
            self.buf = torch.empty(self.nbytes, dtype=torch.uint8, device=where)

    def ptr(self, name):
        return self.buf.data_ptr() + self.offsets[name][0]

    def view(self, name, n):
        """Host arenas: numpy view of the first n elements of an array (no copy)."""
        off, _, dt = self.offsets[name]
        return self.buf.numpy()[off: off + n * np.dtype(dt).itemsize].view(dt)


class JsonIndex:
    """Resumable frame index of one JSON document (mpe_json_index): windows of the document are
    packed without scanning it again.  Keeps the bytes alive."""

    def __init__(self, text):
        import ctypes as C

        from . import lib as L
        self.lib = L.load()
        self.text = text.encode() if isinstance(text, str) else text
        self.handle = C.c_void_p()
        if self.lib.mpe_json_index_create(self.text, len(self.text), C.byref(self.handle)) != 0:
            raise ValueError('mpe_json_index_create: %s' % self.lib.mpe_pack_last_error().decode())

    def close(self):
        if self.handle:
            self.lib.mpe_json_index_free(self.handle)
            self.handle = None

    def __del__(self):
        self.close()


class JsonStage:
    """Staging buffer of the device-side parse (mpe_json_stage_window -> mpe_json_parse_device): the frame
    -> entry table, the entry table and the skeleton strings of one window in ONE buffer, small fixed
    parts first, so that a window travels host -> device as one copy of the used prefix.  `where` =
    'pinned', 'host' or a device."""

    ENTRY_BYTES = 16

    def __init__(self, V, max_frames, text_cap, where):
        import torch
        self.V, self.max_frames, self.text_cap = V, int(max_frames), int(text_cap)
        self.entry_cap = self.max_frames * V
        self.off_feo = 0
        self.off_entries = ((self.max_frames + 1) * 4 + 255) // 256 * 256
        self.off_text = self.off_entries + (self.entry_cap * self.ENTRY_BYTES + 255) // 256 * 256
        self.nbytes = self.off_text + self.text_cap
        if where == 'pinned':
            self.buf = torch.empty(self.nbytes, dtype=torch.uint8).pin_memory()
        elif where == 'host':
            self.buf = torch.empty(self.nbytes, dtype=torch.uint8)
        else:
            self.buf = torch.empty(self.nbytes, dtype=torch.uint8, device=where)

    def ptr(self, what):
        return self.buf.data_ptr() + {'frame_entry_off': self.off_feo, 'entries': self.off_entries, 'text': self.off_text}[what]


class NeedsHostParser(ValueError):
    """The window holds a shape the device-side parser leaves to the host packer."""


def stage_json_window(index, params, stage, frame_start=0, frame_step=1, max_frames=0, n_threads=0):
    """First level of the wire format on the host (mpe_json_stage_window): -> (n_frames, n_entries,
    bytes of the staging buffer in use).  Raises NeedsHostParser for shapes only the host packer handles."""
    import ctypes as C

    from . import lib as L
    lib = L.load()
    sm = list(params.used_cameras_skeleton_matching)
    V = len(sm)
    assert stage.V == V
    names = (C.c_char_p * V)(*[c.encode() for c in sm])
    nf, ne, tb = C.c_int32(), C.c_int32(), C.c_size_t()
    mf = stage.max_frames if max_frames <= 0 or max_frames > stage.max_frames else max_frames
    rc = lib.mpe_json_stage_window(index.handle, names, V, frame_start, frame_step, mf, n_threads,
                                   C.c_void_p(stage.ptr('text')), stage.text_cap, C.c_void_p(stage.ptr('entries')), stage.entry_cap,
                                   C.c_void_p(stage.ptr('frame_entry_off')), C.byref(nf), C.byref(ne), C.byref(tb))
    if rc == L.MPE_ERR_UNSUPPORTED:
        raise NeedsHostParser(lib.mpe_pack_last_error().decode())
    if rc != 0:
        raise ValueError('mpe_json_stage_window: %s' % lib.mpe_pack_last_error().decode())
    return nf.value, ne.value, stage.off_text + tb.value


class ParsedOnDevice:
    """A batch whose arrays exist on the device only (device-side parse): what Engine.match / mlp3d /
    triangulate need of a DeviceBatch.  `download()` brings the arrays back as a PackedBatch (tests)."""

    def __init__(self, arena, V, J, n_frames, n_heads, n_edge_nodes, max_heads_in_a_frame):
        import ctypes as C

        from . import lib as L
        self.arena, self.V, self.J = arena, V, J
        self.n_frames, self.n_heads, self.n_edge_nodes = int(n_frames), int(n_heads), int(n_edge_nodes)
        self._max_heads = int(max_heads_in_a_frame)
        self.host = None
        s = L.mpe_batch()
        s.n_frames, s.n_heads, s.n_edge_nodes = self.n_frames, self.n_heads, self.n_edge_nodes
        for name, _ in ARRAYS:
            setattr(s, 'd_' + name, C.c_void_p(arena.ptr(name)))
        self.struct = s

    def max_heads_per_frame(self):
        return self._max_heads

    def download(self):
        raw = self.arena.buf.cpu().numpy()
        B, H, V, J = self.n_frames, self.n_heads, self.V, self.J

        def arr(name, n):
            off, _, dt = self.arena.offsets[name]
            return raw[off: off + n * np.dtype(dt).itemsize].view(dt).copy()
        pb = PackedBatch(V, J)
        pb.n_frames = B
        pb.frame_head_off, pb.frame_en_off = arr('frame_head_off', B + 1), arr('frame_en_off', B + 1)
        pb.slot_cam, pb.slot_n = arr('slot_cam', B * V).reshape(B, V), arr('slot_n', B * V).reshape(B, V)
        pb.head_cam, pb.skeleton_index = arr('head_cam', H), arr('skeleton_index', H)
        pb.joint_mask, pb.tri_mask = arr('joint_mask', H), arr('tri_mask', H)
        pb.xy, pb.vp = arr('xy', H * J * 2).reshape(H, J, 2), arr('vp', H * J * 2).reshape(H, J, 2)
        return pb


def pack_json_into(text, params, arena, frame_start=0, frame_step=1, max_frames=0, n_threads=0):
    """Native packer straight into a host CapacityArena (page-locked for the production path):
    returns a PackedBatch whose arrays are VIEWS of the arena (valid until the arena is packed into
    again).  `text` = bytes / str, or a JsonIndex of a document that is consumed in windows."""
    import ctypes as C

    from . import lib as L
    lib = L.load()
    index = text if isinstance(text, JsonIndex) else None
    if isinstance(text, str):
        text = text.encode()
    sm = list(params.used_cameras_skeleton_matching)
    V, J = len(sm), len(params.joint_list)
    assert arena.V == V and arena.J == J
    names = (C.c_char_p * V)(*[c.encode() for c in sm])
    dst = L.mpe_pack_dst()
    dst.max_frames, dst.max_heads = arena.max_frames, arena.max_heads
    for name, _, _ in CapacityArena.FIELDS:
        setattr(dst, name, C.c_void_p(arena.ptr(name)))
    nf, nh, ne = C.c_int32(), C.c_int32(), C.c_int32()
    if index is not None:
        rc = lib.mpe_pack_indexed_into(index.handle, names, V, J, frame_start, frame_step, max_frames, n_threads, C.byref(dst),
                                       C.byref(nf), C.byref(nh), C.byref(ne))
    else:
        rc = lib.mpe_pack_json_into(text, len(text), names, V, J, frame_start, frame_step, max_frames, n_threads, C.byref(dst),
                                    C.byref(nf), C.byref(nh), C.byref(ne))
    if rc != 0:
        raise ValueError('mpe_pack_json_into: %s' % lib.mpe_pack_last_error().decode())
    B, H = nf.value, nh.value
    pb = PackedBatch(V, J)
    pb.n_frames = B
    pb.frame_head_off = arena.view('frame_head_off', B + 1)
    pb.frame_en_off = arena.view('frame_en_off', B + 1)
    pb.slot_cam = arena.view('slot_cam', B * V).reshape(B, V)
    pb.slot_n = arena.view('slot_n', B * V).reshape(B, V)
    pb.head_cam = arena.view('head_cam', H)
    pb.skeleton_index = arena.view('skeleton_index', H)
    pb.joint_mask = arena.view('joint_mask', H)
    pb.tri_mask = arena.view('tri_mask', H)
    pb.xy = arena.view('xy', H * J * 2).reshape(H, J, 2)
    pb.vp = arena.view('vp', H * J * 2).reshape(H, J, 2)
    return pb


class _ViewPacker:
    """Per-thread ctypes state of `pack_views` (argument arrays, destination struct, array offsets of a one-frame layout): built once
    per (cameras, joints, capacity)."""

    def __init__(self, sm, J, max_heads):
        import ctypes as C

        from . import lib as L
        self.lib = L.load()
        self.sm, self.V, self.J, self.max_heads = list(sm), len(sm), J, int(max_heads)
        self.cam_index = {c: i for i, c in enumerate(self.sm)}
        V = self.V
        self.texts, self.lens, self.cams = (C.c_char_p * V)(), (C.c_size_t * V)(), (C.c_int32 * V)()
        self.dst = L.mpe_pack_dst()
        self.dst.max_frames, self.dst.max_heads = 1, self.max_heads
        self.nh, self.ne = C.c_int32(), C.c_int32()
        self.p_dst, self.p_nh, self.p_ne = C.byref(self.dst), C.byref(self.nh), C.byref(self.ne)
        count = {'F1': 2, 'FV': V, 'H': self.max_heads, 'HJ2': self.max_heads * J * 2}
        self.offsets, off = [], 0
        for name, dt, kind in CapacityArena.FIELDS:
            self.offsets.append((name, off, np.dtype(dt), kind))
            off += (count[kind] * np.dtype(dt).itemsize + 255) // 256 * 256
        self.nbytes = off
        self.void_p = C.c_void_p
        self.pin = None


_view_packers = None


def pack_views(frame, params, max_heads):
    """ONE frame dict {camera: [text of the skeleton list, ...]} -> PackedBatch through the native packer (`mpe_pack_views_into`),
    without serialising the frame into a document first.  The arrays are views of one byte buffer laid out like the device copy
    (`pb.upload_layout`), so the batch travels to the device as it lies.  ValueError where the packer declines the text."""
    global _view_packers
    import threading
    if _view_packers is None:
        _view_packers = threading.local()
    sm = params.used_cameras_skeleton_matching
    J = len(params.joint_list)
    key = (tuple(sm), J, int(max_heads))
    st = getattr(_view_packers, 'state', None)
    if st is None or st[0] != key:
        st = _view_packers.state = (key, _ViewPacker(sm, J, max_heads))
    vp = st[1]
    n = 0
    keep = []
    ascii_only = True
    for cam, entry in frame.items():
        c = vp.cam_index.get(cam)
        if c is None:
            continue
        if n >= vp.V:
            raise ValueError('frame lists more cameras than configured')
        t = entry[0].encode()          # (AttributeError for a non-string entry: the caller falls back to the Python packer)
        keep.append(t)
        ascii_only = ascii_only and len(t) == len(entry[0])
        vp.texts[n], vp.lens[n], vp.cams[n] = t, len(t), c
        n += 1
    # page-locked where a GPU is present (torch's caching host allocator: a few microseconds once warm, and it keeps a block away from
    # reuse while an asynchronous copy out of it is in flight): the upload is then one asynchronous copy, not a staged blocking one
    pinned = None
    if vp.pin is None:
        import torch
        vp.pin = bool(torch.cuda.is_available())
    if vp.pin:
        import torch
        pinned = torch.empty(vp.nbytes, dtype=torch.uint8, pin_memory=True)
        buf = pinned.numpy()
    else:
        buf = np.empty(vp.nbytes, np.uint8)
    base = buf.ctypes.data
    dst = vp.dst
    for name, off, _, _ in vp.offsets:
        setattr(dst, name, vp.void_p(base + off))
    extents = np.empty((vp.max_heads, 2), np.int32)
    rc = vp.lib.mpe_pack_views_into(vp.texts, vp.lens, vp.cams, n, vp.V, J, vp.p_dst, vp.p_nh, vp.p_ne, extents.ctypes.data)
    if rc != 0:
        raise ValueError('mpe_pack_views_into: %s' % vp.lib.mpe_pack_last_error().decode())
    H = vp.nh.value
    V = vp.V
    size = {'F1': 2, 'FV': V, 'H': H, 'HJ2': H * J * 2}
    pb = PackedBatch(V, J)
    pb.n_frames = 1
    layout = {}
    for name, off, dt, kind in vp.offsets:
        layout[name] = off
        setattr(pb, name, buf[off: off + size[kind] * dt.itemsize].view(dt))
    pb.slot_cam, pb.slot_n = pb.slot_cam.reshape(1, V), pb.slot_n.reshape(1, V)
    pb.xy, pb.vp = pb.xy.reshape(H, J, 2), pb.vp.reshape(H, J, 2)
    pb.upload_layout = (buf, layout, pinned)
    # [begin, end) of every head's skeleton object inside its camera's text (character = byte offsets: ASCII texts only)
    pb.skeleton_extent = extents[:H] if ascii_only else None
    return pb


class DeviceBatch:
    def __init__(self, pb, device, arena=None):
        """arena = None: nine independent device tensors uploaded from pageable numpy arrays
        (tests, one-off calls).  arena = a device BatchArena: the struct points into it and the
        data arrives with `upload(pinned_arena)` (one async copy, stream-ordered)."""
        import ctypes as C

        import torch

        from . import lib as L
        self.host = pb
        self.device = torch.device(device)
        self.arena = arena
        s = L.mpe_batch()
        s.n_frames, s.n_heads, s.n_edge_nodes = pb.n_frames, pb.n_heads, pb.n_edge_nodes
        lay = getattr(pb, 'upload_layout', None)
        if arena is None and lay is not None and pb.en_pair is None:
            # a frame packed by `pack_views`: its arrays already lie in one buffer, 256-byte aligned -- it travels as it is
            if lay[2] is not None and self.device.type == 'cuda':
                self.buf = torch.empty(lay[0].size, dtype=torch.uint8, device=self.device)
                self.buf.copy_(lay[2], non_blocking=True)
            else:
                self.buf = torch.from_numpy(lay[0]).to(self.device)
            base = self.buf.data_ptr()
            for name, _ in ARRAYS:
                setattr(s, 'd_' + name, C.c_void_p(base + lay[1][name]))
        elif arena is None:
            # ONE pageable buffer, one H2D copy (the per-frame mirrors upload a batch per call: nine small copies cost more
            # than the batch's kernels); 256-byte aligned pieces like the arenas
            parts, off = [], 0
            names = [(n, dt) for n, dt in ARRAYS]
            if pb.en_pair is not None:
                ep = np.ascontiguousarray(pb.en_pair, np.int32).reshape(-1, 2)
                if ep.shape[0] != pb.n_edge_nodes:
                    raise ValueError('en_pair holds %d pairs, frame_en_off says %d' % (ep.shape[0], pb.n_edge_nodes))
                names.append(('en_pair', np.int32))
            offs = {}
            for name, dt in names:
                a = np.ascontiguousarray(getattr(pb, name)).reshape(-1)
                a = a.view(dt) if a.dtype != dt and a.dtype.itemsize == np.dtype(dt).itemsize else a.astype(dt, copy=False)
                offs[name] = off
                parts.append((off, a))
                off += (a.nbytes + 255) // 256 * 256
            host = np.zeros(max(off, 256), np.uint8)
            for o, a in parts:
                host[o:o + a.nbytes] = a.view(np.uint8)
            self.buf = torch.from_numpy(host).to(self.device)
            base = self.buf.data_ptr()
            for name, _ in names:
                setattr(s, 'd_' + name, C.c_void_p(base + offs[name]))
        else:
            if pb.en_pair is not None:
                raise ValueError('arena-backed batches carry the implicit topology only')
            for name, _ in ARRAYS:
                setattr(s, 'd_' + name, C.c_void_p(arena.ptr(name)))
        self.struct = s

    def rebind(self, pb):
        """Capacity arenas: the same device buffer now holds another batch (sizes only change)."""
        self.host = pb
        self.struct.n_frames, self.struct.n_heads, self.struct.n_edge_nodes = pb.n_frames, pb.n_heads, pb.n_edge_nodes
        return self

    def upload(self, pinned):
        """One H2D copy of the whole batch from a pinned BatchArena of the same layout, on the
        current stream."""
        assert self.arena is not None and pinned.nbytes == self.arena.nbytes
        self.arena.buf.copy_(pinned.buf, non_blocking=True)

    @property
    def n_frames(self):
        return self.host.n_frames

    @property
    def n_heads(self):
        return self.host.n_heads

    @property
    def n_edge_nodes(self):
        return self.host.n_edge_nodes


def parse_skeletons(frame_cam_entry):
    """`frame[cam]` is `[json-string or list of skeleton dicts, ...]`."""
    sk = frame_cam_entry[0]
    return json.loads(sk) if isinstance(sk, (str, bytes)) else sk


def skeleton_arrays(sk, J):
    """One skeleton dict {joint id: [id, x, y, valid, prob], optional "ID"} -> (joint mask, triangulation mask,
    xy [J,2] f64, vp [J,2] f32); mask 0 = no joint key (the reference drops such skeletons, graph_generator.py:590-591)."""
    xy = np.zeros((J, 2), np.float64)
    vp = np.zeros((J, 2), np.float32)
    m = 0
    t = 0
    for key, val in sk.items():
        if key == 'ID':
            continue
        j = int(key)
        if j < 0 or j >= J:
            raise ValueError('joint id %r out of range' % key)
        m |= 1 << j
        if val[0] > 0.:
            t |= 1 << j
        xy[j, 0], xy[j, 1] = val[1], val[2]
        vp[j, 0], vp[j, 1] = val[3], val[4]
    return m, t, xy, vp


def pack_frames(frames, params, keep_json=False):
    """frames: list of {cam: [json string of skeleton list, ...]} -> PackedBatch."""
    sm = list(params.used_cameras_skeleton_matching)
    V, J = len(sm), len(params.joint_list)
    B = len(frames)
    pb = PackedBatch(V, J)
    pb.n_frames = B
    slot_cam = np.full((B, V), -1, np.int32)
    slot_n = np.zeros((B, V), np.int32)
    head_off = np.zeros(B + 1, np.int32)
    en_off = np.zeros(B + 1, np.int32)
    head_cam, jmask, tmask, skidx = [], [], [], []
    xy_rows, vp_rows = [], []
    jsons = [] if keep_json else None
    for f, frame in enumerate(frames):
        s = 0
        counts = []
        fj = {} if keep_json else None
        nh = 0
        for cam in frame:
            if cam not in sm:
                continue
            if s >= V:
                raise ValueError('frame %d lists more cameras than configured' % f)
            c = sm.index(cam)
            n_here = 0
            for idx, sk in enumerate(parse_skeletons(frame[cam])):
                m, t, xy, vp = skeleton_arrays(sk, J)
                if m == 0:
                    continue
                head_cam.append(c)
                jmask.append(m)
                tmask.append(t)
                skidx.append(idx)
                xy_rows.append(xy)
                vp_rows.append(vp)
                if keep_json:
                    fj[nh] = sk
                n_here += 1
                nh += 1
            slot_cam[f, s] = c
            slot_n[f, s] = n_here
            counts.append(n_here)
            s += 1
        tot = sum(counts)
        pairs = (tot * tot - sum(k * k for k in counts)) // 2
        head_off[f + 1] = head_off[f] + tot
        en_off[f + 1] = en_off[f] + pairs
        if keep_json:
            jsons.append(fj)
    n = len(head_cam)
    pb.frame_head_off, pb.frame_en_off = head_off, en_off
    pb.slot_cam, pb.slot_n = slot_cam, slot_n
    pb.head_cam = np.array(head_cam, np.int32).reshape(n)
    pb.joint_mask = np.array(jmask, np.uint32).reshape(n)
    pb.tri_mask = np.array(tmask, np.uint32).reshape(n)
    pb.skeleton_index = np.array(skidx, np.int32).reshape(n)
    pb.xy = np.stack(xy_rows).reshape(n, J, 2) if n else np.zeros((0, J, 2), np.float64)
    pb.vp = np.stack(vp_rows).reshape(n, J, 2) if n else np.zeros((0, J, 2), np.float32)
    pb.jsons_for_head = jsons
    return pb


def pairs_of_frame(slot_n_row):
    """(h1, h2) of every edge-node of one frame, in edge-node order (host mirror of the
    device topology kernel; used by the Python API mirrors and tests)."""
    starts = np.concatenate([[0], np.cumsum(slot_n_row)])
    out = []
    V = len(slot_n_row)
    for a in range(V):
        for b in range(a + 1, V):
            for i in range(slot_n_row[a]):
                for j in range(slot_n_row[b]):
                    out.append((starts[a] + i, starts[b] + j))
    return np.array(out, np.int32).reshape(-1, 2)


def pack_json(text, params, frame_start=0, frame_step=1, max_frames=0, n_threads=0):
    """Native packer (csrc/packer.cpp, `mpe_pack_json`): the whole JSON document (bytes or str,
    list of frames in the reference's wire format) -> PackedBatch, without Python-level JSON
    parsing.  Same ordering rules and numbers as `pack_frames`."""
    import ctypes as C

    from . import lib as L
    lib = L.load()
    if isinstance(text, str):
        text = text.encode()
    sm = list(params.used_cameras_skeleton_matching)
    V, J = len(sm), len(params.joint_list)
    names = (C.c_char_p * V)(*[c.encode() for c in sm])
    handle = C.c_void_p()
    rc = lib.mpe_pack_json(text, len(text), names, V, J, frame_start, frame_step, max_frames, n_threads,
                           C.byref(handle))
    if rc != 0:
        raise ValueError('mpe_pack_json: %s' % lib.mpe_pack_last_error().decode())
    try:
        v = L.mpe_packed_arrays()
        lib.mpe_packed_view(handle, C.byref(v))

        def arr(ptr, n, dt):
            if n == 0:
                return np.zeros(0, dt)
            return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dt, copy=True)
        pb = PackedBatch(V, J)
        B, H = v.n_frames, v.n_heads
        pb.n_frames = B
        pb.frame_head_off = arr(v.frame_head_off, B + 1, np.int32)
        pb.frame_en_off = arr(v.frame_en_off, B + 1, np.int32)
        pb.slot_cam = arr(v.slot_cam, B * V, np.int32).reshape(B, V)
        pb.slot_n = arr(v.slot_n, B * V, np.int32).reshape(B, V)
        pb.head_cam = arr(v.head_cam, H, np.int32)
        pb.skeleton_index = arr(v.skeleton_index, H, np.int32)
        pb.joint_mask = arr(v.joint_mask, H, np.uint32)
        pb.tri_mask = arr(v.tri_mask, H, np.uint32)
        pb.xy = arr(v.xy, H * J * 2, np.float64).reshape(H, J, 2)
        pb.vp = arr(v.vp, H * J * 2, np.float32).reshape(H, J, 2)
        return pb
    finally:
        lib.mpe_packed_free(handle)


def generated_scene(views, params):
    """Node and edge-node order of ONE graph of MergedMultipleHumansDataset.process_training (reference
    graph_generator.py:697-797; what mode='test_generated' of test/sm_metrics_without_gt.py:108 builds).  `views` = the
    sampled single-person frames ({cam: [json string of the skeleton list, ...]}) of the scene.

    Heads: view after view (person-major), inside a view in the dict's camera order and list order, skipping skeletons
    without a joint key (load_people_view_graph, :573-605).  Per view and camera the skeleton with the most joints is the
    person's (first one on ties, :723), the others are spurious (:724-726).  Edge-nodes, one per ORDERED pair of heads of
    different cameras: per person its own heads (label 1, :749-760), then its heads x every other person's (label 0,
    :762-774), then its heads x the spurious ones (:776-786); after all persons the spurious x spurious pairs (:788-797).
    Returns dict(heads=[(camera name, index in its list, skeleton dict)], pairs [M,2] int32, labels [M] f64)."""
    sm = list(params.used_cameras_skeleton_matching)
    heads, people, spurious = [], [], []
    for view in views:
        person = []
        for cam in view:
            if cam not in sm:
                continue
            ids, nj = [], []
            for idx, sk in enumerate(parse_skeletons(view[cam])):
                n = sum(1 for k in sk if k != 'ID')
                if n == 0:
                    continue
                ids.append(len(heads))
                nj.append(n)
                heads.append((cam, idx, sk))
            if nj:
                good = max(enumerate(nj), key=lambda x: x[1])[0]
                spurious += [(h, cam) for h in ids if h != ids[good]]
                person.append((ids[good], cam))
        people.append(person)
    pairs, labels = [], []

    def link(group1, group2, label):
        for h1, c1 in group1:
            for h2, c2 in group2:
                if c1 == c2:
                    continue
                pairs.append((h1, h2))
                labels.append(label)
    for ip, person in enumerate(people):
        link(person, person, 1.0)
        for io, other in enumerate(people):
            if io != ip:
                link(person, other, 0.0)
        link(person, spurious, 0.0)
    link(spurious, spurious, 0.0)
    return {'heads': heads, 'pairs': np.array(pairs, np.int32).reshape(-1, 2), 'labels': np.array(labels, np.float64)}


def pack_scenes(scenes, params, keep_json=False):
    """[generated_scene(...)] -> PackedBatch with an explicit edge-node list (one frame per scene)."""
    sm = list(params.used_cameras_skeleton_matching)
    V, J = len(sm), len(params.joint_list)
    B = len(scenes)
    pb = PackedBatch(V, J)
    pb.n_frames = B
    head_off = np.zeros(B + 1, np.int32)
    en_off = np.zeros(B + 1, np.int32)
    head_cam, jmask, tmask, skidx, xy_rows, vp_rows, pairs = [], [], [], [], [], [], []
    jsons = [] if keep_json else None
    for f, sc in enumerate(scenes):
        for cam, idx, sk in sc['heads']:
            m, t, xy, vp = skeleton_arrays(sk, J)
            head_cam.append(sm.index(cam))
            jmask.append(m)
            tmask.append(t)
            skidx.append(idx)
            xy_rows.append(xy)
            vp_rows.append(vp)
        if keep_json:
            jsons.append({i: h[2] for i, h in enumerate(sc['heads'])})
        head_off[f + 1] = head_off[f] + len(sc['heads'])
        en_off[f + 1] = en_off[f] + len(sc['pairs'])
        pairs.append(np.asarray(sc['pairs'], np.int32).reshape(-1, 2))
    n = len(head_cam)
    pb.frame_head_off, pb.frame_en_off = head_off, en_off
    pb.slot_cam = np.full((B, V), -1, np.int32)
    pb.slot_n = np.zeros((B, V), np.int32)
    pb.head_cam = np.array(head_cam, np.int32).reshape(n)
    pb.joint_mask = np.array(jmask, np.uint32).reshape(n)
    pb.tri_mask = np.array(tmask, np.uint32).reshape(n)
    pb.skeleton_index = np.array(skidx, np.int32).reshape(n)
    pb.xy = np.stack(xy_rows).reshape(n, J, 2) if n else np.zeros((0, J, 2), np.float64)
    pb.vp = np.stack(vp_rows).reshape(n, J, 2) if n else np.zeros((0, J, 2), np.float32)
    pb.en_pair = np.concatenate(pairs).reshape(-1, 2) if pairs else np.zeros((0, 2), np.int32)
    pb.jsons_for_head = jsons
    return pb


def concat_packed(batches):
    """Frames of several PackedBatch objects back to back (the engine's frame batch = what dgl.batch is to the
    reference, train_skeleton_matching.py:67-84).  All explicit or all implicit."""
    first = batches[0]
    explicit = first.en_pair is not None
    if any((b.en_pair is not None) != explicit or b.V != first.V or b.J != first.J for b in batches):
        raise ValueError('cannot batch graphs of different kinds')
    pb = PackedBatch(first.V, first.J)
    pb.n_frames = sum(b.n_frames for b in batches)

    def offs(name):
        out, base = [np.zeros(1, np.int32)], 0
        for b in batches:
            a = np.asarray(getattr(b, name), np.int32)
            out.append(a[1:] + base)
            base += int(a[-1])
        return np.concatenate(out).astype(np.int32)
    pb.frame_head_off, pb.frame_en_off = offs('frame_head_off'), offs('frame_en_off')
    for name in ('slot_cam', 'slot_n', 'head_cam', 'joint_mask', 'tri_mask', 'skeleton_index', 'xy', 'vp'):
        setattr(pb, name, np.concatenate([np.asarray(getattr(b, name)) for b in batches]))
    if explicit:
        pb.en_pair = np.concatenate([np.asarray(b.en_pair, np.int32).reshape(-1, 2) for b in batches])
    if all(b.jsons_for_head is not None for b in batches):
        pb.jsons_for_head = [j for b in batches for j in b.jsons_for_head]
    return pb
