"""Batched engine over libmpe_hip.so: the frame-batched form of the reference's per-frame
loop body (test/metrics_from_model.py:178-294, test/metrics_from_triangulation.py:187-272).

PyTorch is plumbing here (device memory, current stream); every computation on the path is
a HIP kernel behind the C ABI of include/mpe.h.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import lib as L
from .calibration import Calibration
from .packing import (CapacityArena, DeviceBatch, JsonIndex, JsonStage, NeedsHostParser, PackedBatch, ParsedOnDevice, pack_frames,
                      pack_json, pack_json_into, stage_json_window)


def _f32p(a):
    return a.ctypes.data_as(L.c_f32p)


class Engine:
    def __init__(self, params=None, calib=None, max_frames=1024, max_heads_per_frame=None,
                 max_persons_per_camera=4, device='cuda:0', threshold=0.5, max_edge_nodes_per_frame=None):
        from .parameters import parameters as default_params
        self.params = params or default_params
        self.calib = calib or Calibration(self.params)
        self.lib = L.load()
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError('Engine needs a GPU device (there is no CPU fallback)')
        torch.cuda.set_device(self.device)
        if self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        p = self.params
        sm = list(p.used_cameras_skeleton_matching)
        if sm != list(p.used_cameras) or sm != [c for c in p.camera_names if c in sm]:
            raise ValueError('used_cameras_skeleton_matching must equal used_cameras in camera_names order')
        self.V, self.J = len(sm), len(p.joint_list)
        cal_idx = [self.calib.index(c) for c in sm]
        hpf = max_heads_per_frame or self.V * max_persons_per_camera
        self.max_frames = int(max_frames)
        self.hpf = int(hpf)
        self.pcap = max(1, hpf // max(1, p.min_number_of_views))
        # largest edge-node count of a frame with `hpf` heads (even spread over V cameras)
        m_frame = hpf * hpf * (self.V - 1) // (2 * self.V) + 1
        # batches with an explicit edge-node list (process_training graphs: one edge-node per ORDERED head pair) hold up to
        # explicit_m_cap(hpf) edge-nodes per frame; `max_edge_nodes_per_frame` sizes the batch totals for them
        if max_edge_nodes_per_frame:
            m_frame = max(m_frame, int(max_edge_nodes_per_frame))
        self._keep = {
            'Kinv': np.ascontiguousarray(self.calib.Kinv32[cal_idx].reshape(-1), np.float32),
            'K': np.ascontiguousarray(self.calib.K32[cal_idx].reshape(-1), np.float32),
            'T_i': np.ascontiguousarray(self.calib.T_i32[cal_idx].reshape(-1), np.float32),
            'P': np.ascontiguousarray(self.calib.P[cal_idx].reshape(-1), np.float64),
            'dist': np.ascontiguousarray(self.calib.dist[cal_idx].reshape(-1), np.float64),
        }
        cfg = L.mpe_config()
        cfg.n_cameras, cfg.n_joints = self.V, self.J
        cfg.image_width, cfg.image_height = p.image_width, p.image_height
        cfg.numbers_per_joint = p.numbers_per_joint
        cfg.min_views = p.min_number_of_views
        cfg.median_axis = p.axes_3D['Y'][0]
        cfg.used_joint_mask = sum(1 << j for j in p.used_joints)
        cfg.threshold = threshold
        cfg.median_window = 0.05
        cfg.max_frames = self.max_frames
        cfg.max_heads = self.max_frames * hpf
        cfg.max_edge_nodes = self.max_frames * m_frame
        cfg.max_heads_per_frame = hpf
        cfg.max_persons_per_frame = self.pcap
        cfg.Kinv = _f32p(self._keep['Kinv'])
        cfg.K = _f32p(self._keep['K'])
        cfg.T_i = _f32p(self._keep['T_i'])
        cfg.P = self._keep['P'].ctypes.data_as(L.c_f64p)
        cfg.dist = self._keep['dist'].ctypes.data_as(L.c_f64p)
        self.ctx = C.c_void_p()
        rc = self.lib.mpe_create(C.byref(cfg), C.byref(self.ctx))
        if rc != 0:
            raise L.MpeError(rc, 'mpe_create failed')
        self.gat_dims = None
        self.mlp_out = None
        self.m_frame = m_frame
        self._made_with = dict(params=self.params, calib=self.calib, max_frames=self.max_frames, max_heads_per_frame=self.hpf,
                               max_persons_per_camera=max_persons_per_camera, device=str(self.device), threshold=threshold,
                               max_edge_nodes_per_frame=max_edge_nodes_per_frame)
        self._state = {}                 # what was loaded / set, so that sibling() can repeat it
        self._siblings = []
        self._json_streams = None
        if os.environ.get('MPE_JSON_STREAMS_EARLY', '0') == '1':           # diagnostics (see _make_json_streams)
            self._make_json_streams()

    def _make_json_streams(self):
        """The three streams of the JSON pipeline (parse; matching / 3D compute), made at the first stream_json call.
        Compute at high priority, parse at normal: the parse kernels of window i+1 take the slots the GEMMs of window i
        leave.  HIP multiplexes its streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); with the 6-9 streams of
        an application that also pipelines its copies, the parse stream and the compute stream can land on one queue
        and then serialise (measured inside bench.py: 145-159 k or 171 k frames/s depending on the order in which streams
        were first used).  With 8 hardware queues (GPU_MAX_HW_QUEUES=8 before HIP initialises: bench.py sets it for itself, lib.py on
        import when MPE_SET_HW_QUEUES=1 asks for it): 170.5-170.7 k in 5 of 5 runs."""
        if self._json_streams is None:
            self._json_streams = (torch.cuda.Stream(self.device, priority=int(os.environ.get('MPE_JSON_PARSE_PRIO', '0'))),
                                  torch.cuda.Stream(self.device, priority=int(os.environ.get('MPE_JSON_COMPUTE_PRIO', '-1'))),
                                  torch.cuda.Stream(self.device, priority=int(os.environ.get('MPE_JSON_COMPUTE_PRIO', '-1'))))
            for s_ in self._json_streams:
                with torch.cuda.stream(s_):
                    torch.zeros(8, device=self.device).add_(1)
        return self._json_streams

    def close(self):
        for e in getattr(self, '_siblings', []):
            e.close()
        self._siblings = []
        if getattr(self, 'ctx', None):
            self.lib.mpe_destroy(self.ctx)
            self.ctx = None

    def sibling(self, k=1):
        """The k-th further context with the same configuration, weights, precision mode and threshold (own workspace,
        124 MB of weights again).  A context is one dependent chain of ~45 kernels per batch; two contexts that take
        turns on the batches (run_pipelined(contexts=2), stream_json(contexts=2), bench.py) keep two such chains in
        flight and fill each other's launch tails: 195.6 k against 186-189 k frames/s for one context with its two stages
        on two streams (tools/two_engines.py, MPE_ALTERNATE=1).  Results are bit-identical: same kernels, same data."""
        while len(self._siblings) < k:
            e = Engine(**self._made_with)
            st = self._state
            if 'gat' in st:
                e.load_gat(*st['gat'])
            if 'mlp' in st:
                e.load_mlp(*st['mlp'])
            if 'precision' in st:
                e.set_precision(*st['precision'])
            if 'threshold' in st:
                e.set_threshold(st['threshold'])
            if 'gat_output' in st:
                e.set_gat_output(st['gat_output'])
            self._siblings.append(e)
        return self._siblings[k - 1]

    def contexts(self, n):
        """[self, sibling(1), ..., sibling(n - 1)]"""
        return [self] + [self.sibling(k) for k in range(1, int(n))]

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- weights ----------------------------------------------------------------------
    def _chk(self, rc):
        L.check(self.ctx, rc)

    def load_gat(self, state_dict, prm):
        """state_dict: reference names ``layers.{l}.{fc1,fc2}.{weight,bias}``, ``attn_l/r``
        (numpy or torch); prm: contents of skeleton_matching.prms."""
        sd = {k: _np32(v) for k, v in state_dict.items()}
        self._state['gat'] = (sd, prm)
        for e in self._siblings:
            e.load_gat(sd, prm)
        n_layers = int(prm['gnn_layers'])
        heads = list(prm['heads']) + [1]
        slope = prm.get('nonlinearity', 0.01)
        slope = getattr(slope, 'negative_slope', slope)
        self._chk(self.lib.mpe_set_gat_params(self.ctx, n_layers, float(prm['alpha']), float(slope)))
        dims = []
        for l in range(n_layers):
            w1 = sd['layers.%d.fc1.weight' % l]
            w2 = sd['layers.%d.fc2.weight' % l]
            in_dim, nh = w1.shape[1], heads[l]
            out_dim = w2.shape[0] // nh
            al = np.ascontiguousarray(sd['layers.%d.attn_l' % l].reshape(nh, out_dim))
            ar = np.ascontiguousarray(sd['layers.%d.attn_r' % l].reshape(nh, out_dim))
            self._chk(self.lib.mpe_set_gat_layer(
                self.ctx, l, in_dim, nh, out_dim, _f32p(w1), _f32p(sd['layers.%d.fc1.bias' % l]),
                _f32p(w2), _f32p(sd['layers.%d.fc2.bias' % l]), _f32p(al), _f32p(ar)))
            dims.append((in_dim, nh, out_dim))
        self.gat_dims = dims

    def load_mlp(self, state_dict, slope=0.1):
        sd = {k: _np32(v) for k, v in state_dict.items()}
        self._state['mlp'] = (sd, slope)
        for e in self._siblings:
            e.load_mlp(sd, slope)
        keys = sorted({int(k.split('.')[1]) for k in sd})
        self._chk(self.lib.mpe_set_mlp_params(self.ctx, len(keys), float(slope)))
        for n, k in enumerate(keys):
            w = sd['layers.%d.weight' % k]
            self._chk(self.lib.mpe_set_mlp_layer(self.ctx, n, w.shape[1], w.shape[0], _f32p(w),
                                                 _f32p(sd['layers.%d.bias' % k])))
            self.mlp_out = w.shape[0]

    # ---- batches ----------------------------------------------------------------------
    def pack(self, frames, keep_json=False):
        return pack_frames(frames, self.params, keep_json=keep_json)

    def pack_json(self, text, frame_start=0, frame_step=1, max_frames=0, n_threads=0):
        """Native (C++) packer: JSON text of a list of frames -> PackedBatch."""
        return pack_json(text, self.params, frame_start, frame_step, max_frames, n_threads)

    def stream_json(self, text, chunk_frames=None, mode='mlp', frame_step=1, n_threads=0, parser='device', contexts=1):
        """Frame JSON (bytes, the reference's wire format) -> 3D poses, chunk by chunk, with the
        host side off the critical path: the native packer parses chunk i+1 straight into a
        page-locked arena (worker thread; the C call releases the GIL) while chunk i is copied to
        the device in ONE transfer and runs mpe_match_batch + the 3D stage; results come back into
        page-locked memory.  Yields (PackedBatch view, poses [B,Pcap,J,3], n_persons [B]) per chunk.
        LIFETIME: the yielded view and arrays are valid until the next `next()` on the generator only (with contexts = 2
        there are four buffer sets instead of two and two windows in flight; the rule is the same) --
        resuming it starts the parse of a later chunk into the arena behind the view (two host arenas)
        and the result arrays of the slot are rewritten one chunk after that.  With the device parser the
        page-locked result buffers belong to the ENGINE (kept across calls: pinning them costs more than a
        window), so arrays from an earlier call are rewritten by the next call as well.  Copy what you keep.

        parser = 'device' (default): the host keeps the first level of the format only (frame extents, camera
        keys, the extent of every skeleton STRING: stage_json_window) and the strings are parsed ON THE DEVICE
        (csrc/jsonparse.hip) on a side stream while the previous chunk computes; the first element yielded is
        then a ParsedOnDevice (n_frames, n_heads, ...; `.download()` for the arrays; a window that fell back to the host
        packer yields a PackedBatch there instead -- both carry n_frames / n_heads / n_edge_nodes / frame_counts-style
        offsets, test with isinstance if you need the host arrays).  A chunk holding a shape
        the device parser leaves to the host (literals, nested values, numbers beyond the exact fast path)
        is packed by the host packer instead -- same arrays either way.  parser = 'host': the round-2 path.
        contexts = 2 (device parser only): windows take turns on two contexts (sibling()), two windows in flight."""
        from concurrent.futures import ThreadPoolExecutor
        if isinstance(text, str):
            text = text.encode()
        B = int(chunk_frames or self.max_frames)
        if B > self.max_frames:
            raise ValueError('chunk of %d frames exceeds max_frames=%d' % (B, self.max_frames))
        if int(contexts) not in (1, 2):
            raise ValueError('stream_json takes contexts = 1 or 2 (two compute streams exist; more contexts measured slower, DESIGN 7.1)')
        if parser == 'device':
            # the page-locked result buffers, the arenas and the streams of the device parser belong to the ENGINE: a second
            # generator running on it at the same time would overwrite the first one's results
            if getattr(self, '_json_busy', False):
                raise RuntimeError('another stream_json(parser="device") generator is still open on this engine; exhaust or close() it '
                                   'first, or use a second Engine / Engine.sibling()')
            self._json_busy = True
            try:
                yield from self._stream_json_device(text, B, mode, frame_step, n_threads, contexts)
            finally:
                self._json_busy = False
            return
        H = B * self.hpf
        host = [CapacityArena(self.V, self.J, B, H, 'pinned') for _ in range(2)]
        dev = [CapacityArena(self.V, self.J, B, H, self.device) for _ in range(2)]
        dbs = [None, None]
        out_dt = torch.float32 if mode == 'mlp' else torch.float64
        out = [(torch.empty((B, self.pcap, self.J, 3), dtype=out_dt).pin_memory(),
                torch.empty((B,), dtype=torch.int32).pin_memory()) for _ in range(2)]
        uploaded = [torch.cuda.Event() for _ in range(2)]
        done = [torch.cuda.Event() for _ in range(2)]
        pool = ThreadPoolExecutor(1)
        index = JsonIndex(text)                 # the document is scanned once, window by window

        def parse(i):
            return pack_json_into(index, self.params, host[i & 1], frame_start=i * B * frame_step, frame_step=frame_step,
                                  max_frames=B, n_threads=n_threads)
        try:
            fut = pool.submit(parse, 0)
            i = 0
            pending = None                                  # (slot, pb) whose results are still in flight
            while True:
                pb = fut.result()
                if pb.n_frames == 0:
                    break
                k = i & 1
                self.check_capacity(pb)
                # arrays of the views are exact-size; keep a private copy of the offsets the caller may read
                if dbs[k] is None:
                    dbs[k] = DeviceBatch(pb, self.device, arena=dev[k])
                db = dbs[k].rebind(pb)
                dev[k].buf.copy_(host[k].buf, non_blocking=True)
                uploaded[k].record()
                _, persons, n_persons = self.match(db, want_scores=False)
                poses = (self.mlp3d(db, persons, n_persons) if mode == 'mlp' else self.triangulate(db, persons, n_persons))[0]
                out[k][0][:pb.n_frames].copy_(poses, non_blocking=True)
                out[k][1][:pb.n_frames].copy_(n_persons, non_blocking=True)
                done[k].record()
                last = pb.n_frames < B
                if pending is not None:
                    pk, ppb = pending
                    done[pk].synchronize()
                    yield ppb, out[pk][0][:ppb.n_frames].numpy(), out[pk][1][:ppb.n_frames].numpy()
                pending = (k, pb)
                if last:
                    break
                # the other host arena may be parsed into again once ITS upload has completed
                # (that was chunk i-1, whose results were just handed out)
                fut = pool.submit(parse, i + 1)
                i += 1
            if pending is not None:
                pk, ppb = pending
                done[pk].synchronize()
                yield ppb, out[pk][0][:ppb.n_frames].numpy(), out[pk][1][:ppb.n_frames].numpy()
        finally:
            pool.shutdown(wait=True)
            index.close()

    def _stream_json_device(self, text, B, mode, frame_step, n_threads, contexts=1):
        from concurrent.futures import ThreadPoolExecutor
        import os
        import time
        t_call = time.perf_counter()
        H = B * self.hpf
        # page-locked staging, the device arenas, the page-locked result buffers and the streams are kept with the engine:
        # allocating and pinning ~150 MB per call cost more than parsing a few windows
        cache = self.__dict__.setdefault('_json_bufs', {})
        if B not in cache:
            cache[B] = ([self.json_device_buffers(B) for _ in range(2)], {}, {}, self._make_json_streams())
        bufs, fb, outs, (s_parse, s_m, s_d) = cache[B]      # fb: host-packer fallback buffers, made on first use
        K = 2 if int(contexts) >= 2 else 1                   # contexts taking turns on the windows (two compute streams exist)
        # Window slots (staging + arena + result buffers): S = K + 2 -- K windows computing, one parsed and about to compute,
        # one being uploaded and parsed (the loop below queues the parse one window ahead).  A slot is reused when its window
        # has been handed out.
        S = K + 2
        while len(bufs) < S:
            bufs.append(self.json_device_buffers(B))
        engs = self.contexts(K)
        if K == 2:
            # window i on context i & 1, each context on its own stream: two windows in flight fill each other's tails
            lanes = [(s_m, s_m), (s_d, s_d)]
        elif os.environ.get('MPE_JSON_STREAMS', '1') == '1':
            # ONE compute stream beside the parse stream: a second compute stream (matching of window i+1 beside the 3D stage
            # of window i, what run_pipelined does for resident batches) measured the same or worse here -- the parse kernels
            # already fill what the GEMMs leave
            lanes = [(s_m, s_m)]
        else:
            lanes = [(s_m, s_d)]
        out_dt = torch.float32 if mode == 'mlp' else torch.float64
        outs.setdefault(mode, [])
        while len(outs[mode]) < S:
            outs[mode].append((torch.empty((B, self.pcap, self.J, 3), dtype=out_dt).pin_memory(),
                               torch.empty((B,), dtype=torch.int32).pin_memory()))
        out = outs[mode]
        done = [None] * S
        cur = torch.cuda.current_stream(self.device)
        for s_ in (s_parse, s_m, s_d):
            s_.wait_stream(cur)
        pool = ThreadPoolExecutor(1)
        index = JsonIndex(text)
        t_setup = time.perf_counter() - t_call

        def stage(i):
            try:
                return stage_json_window(index, self.params, bufs[i % S]['host'], frame_start=i * B * frame_step, frame_step=frame_step,
                                         max_frames=B, n_threads=n_threads)
            except NeedsHostParser:
                return None
            except ValueError as e:
                if 'too small' in str(e):                    # an unusually long window: let the host packer size it
                    return None
                raise

        def host_pack(i):
            # rare path: everything that is in flight finishes first (the two fallback arenas and the context are then free)
            for j in range(S):
                copy_out(j)
            torch.cuda.synchronize(self.device)
            k = i & 1
            if not fb:
                fb['host'] = CapacityArena(self.V, self.J, B, H, 'pinned')
                fb['dev'] = [CapacityArena(self.V, self.J, B, H, self.device) for _ in range(2)]
                fb['db'] = [None, None]
            pb = pack_json_into(index, self.params, fb['host'], frame_start=i * B * frame_step, frame_step=frame_step, max_frames=B,
                                n_threads=n_threads)
            if pb.n_frames == 0:
                return None
            self.check_capacity(pb)
            import copy
            keep = copy.copy(pb)
            for name in ('frame_head_off', 'frame_en_off', 'slot_cam', 'slot_n', 'head_cam', 'skeleton_index', 'joint_mask', 'tri_mask', 'xy', 'vp'):
                setattr(keep, name, np.array(getattr(pb, name)))      # the one host arena is reused by the next fallback
            if fb['db'][k] is None:
                fb['db'][k] = DeviceBatch(keep, self.device, arena=fb['dev'][k])
            db = fb['db'][k].rebind(keep)
            fb['dev'][k].buf.copy_(fb['host'].buf, non_blocking=True)
            cur.synchronize()                                  # the pinned arena is free again (rare path)
            return db
        timing = [] if os.environ.get('MPE_JSON_TIMING') else None
        gpu_ev = []
        # Copy engines serve their requests in order: a D2H of results queued behind kernels that are still running holds up the
        # H2D of a later window's strings until those kernels are done (measured: 6.4 ms instead of 0.57 ms).  So the results of
        # a window are queued for copy-out only after the upload of the window that is parsed next.
        res = [None] * S                                     # (poses, n_persons, n, lane) of the slot, still on the device

        def copy_out(k):
            if res[k] is None:
                return
            poses_, n_persons_, n_, lane_ = res[k]
            with torch.cuda.stream(lane_):                 # behind the 3D stage that produced them
                out[k][0][:n_].copy_(poses_, non_blocking=True)
                out[k][1][:n_].copy_(n_persons_, non_blocking=True)
                done[k] = torch.cuda.Event()
                done[k].record()
            res[k] = None
        # The loop runs the PARSE ONE WINDOW AHEAD of the compute: in iteration i the upload and the parse kernels of window
        # i+1 are queued first, then the host waits for the totals of window i (queued an iteration ago, so they are there or
        # nearly), launches its compute and hands out the results of window i-K.  Queued in the same iteration as its own
        # compute, a parse was on the host's critical path for its whole duration beside K computes (3 ms of a 5.5 ms window).
        futs, pev = {}, {}

        def queue(i):
            """Stage result of window i -> its upload + parse on the parse stream.  ('dev', frames) | ('host', B) | ('end', 0)"""
            t0_ = time.perf_counter()
            st = futs.pop(i).result()
            t1_ = time.perf_counter()
            if st is None:
                return ('host', B, t1_ - t0_, 0.0)
            nf, ne, used = st
            if nf == 0:
                return ('end', 0, t1_ - t0_, 0.0)
            k = i % S
            assert res[k] is None, 'window slot reused before its results were handed out'
            with torch.cuda.stream(s_parse):
                if done[k] is not None:
                    s_parse.wait_event(done[k])          # the arena of this slot: window i - S has computed and its results are out
                if timing is not None:
                    e0 = torch.cuda.Event(enable_timing=True)
                    e0.record()
                    bufs[k]['ev_h2d'] = torch.cuda.Event(enable_timing=True)
                self.parse_json_device(bufs[k], nf, ne, used)
                if timing is not None:
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record()
                    pev[i] = (e0, e1, bufs[k]['ev_h2d'])
            return ('dev', nf, t1_ - t0_, time.perf_counter() - t1_)

        def may_continue(w):
            return w[0] == 'host' or (w[0] == 'dev' and w[1] == B)
        try:
            futs[0] = pool.submit(stage, 0)
            w_cur = queue(0)
            if may_continue(w_cur):
                futs[1] = pool.submit(stage, 1)
            i = 0
            pending = []
            while w_cur[0] != 'end':
                k = i % S                                   # slot of this window
                w_nxt = ('end', 0, 0.0, 0.0)
                if i + 1 in futs:
                    w_nxt = queue(i + 1)
                    if may_continue(w_nxt):
                        futs[i + 2] = pool.submit(stage, i + 2)   # into the staging buffer of window i + 2 - S, uploaded long ago
                t_c = time.perf_counter()
                db = None
                if w_cur[0] == 'dev':
                    db = self.finish_parse(bufs[k])
                    if db is not None:
                        if db.max_heads_per_frame() > self.hpf:
                            raise ValueError('a frame holds %d skeletons, capacity is %d (raise max_persons_per_camera / '
                                             'max_heads_per_frame)' % (db.max_heads_per_frame(), self.hpf))
                        lanes[i % K][0].wait_event(bufs[k]['ready'])
                t_d = time.perf_counter()
                from_device = db is not None
                if db is None:                                  # this window goes through the host packer
                    db = host_pack(i)
                    if db is None:
                        break
                n_here = db.n_frames
                eng, (l_m, l_d) = engs[i % K], lanes[i % K]
                # the results of window i-K (same context, same stream) leave the device BEFORE this window's kernels are queued
                # behind them on that stream; this window's upload is already in the copy queue, the next one's comes an
                # iteration later, when window i-K has long finished (copy engines serve in order)
                if len(pending) >= K:
                    copy_out(pending[0][0])
                if timing is not None:
                    ev_c0 = torch.cuda.Event(enable_timing=True)
                    ev_c0.record(l_m)
                if db.host is not None:
                    l_m.wait_stream(cur)                     # host-packed window: its upload went over the caller's stream
                with torch.cuda.stream(l_m):
                    _, persons, n_persons = eng.match(db, want_scores=False)
                    ev_m = torch.cuda.Event()
                    ev_m.record()
                with torch.cuda.stream(l_d):
                    if l_d is not l_m:
                        l_d.wait_event(ev_m)
                    poses = (eng.mlp3d(db, persons, n_persons) if mode == 'mlp' else eng.triangulate(db, persons, n_persons))[0]
                if l_d is not l_m:
                    for t_ in (persons, n_persons):
                        t_.record_stream(l_d)
                res[k] = (poses, n_persons, n_here, l_d)
                done[k] = None
                if timing is not None and from_device and i in pev:
                    ev_c1 = torch.cuda.Event(enable_timing=True)
                    ev_c1.record(l_d)
                    gpu_ev.append(pev.pop(i)[:2] + (ev_c0, ev_c1) + (bufs[k]['ev_h2d'],))
                t_e = time.perf_counter()
                pending.append((k, db if db.host is None else db.host, n_here))
                if len(pending) > K:                          # K windows stay in flight; the oldest one is handed out
                    pk, pinfo, pn = pending.pop(0)
                    copy_out(pk)                              # (already queued above)
                    done[pk].synchronize()
                    if timing is not None:
                        timing.append((w_nxt[2], w_nxt[3], t_d - t_c, t_e - t_d, time.perf_counter() - t_e))
                    yield pinfo, out[pk][0][:pn].numpy(), out[pk][1][:pn].numpy()
                if n_here < B:
                    break
                w_cur = w_nxt
                i += 1
            while pending:
                pk, pinfo, pn = pending.pop(0)
                copy_out(pk)
                done[pk].synchronize()
                yield pinfo, out[pk][0][:pn].numpy(), out[pk][1][:pn].numpy()
        finally:
            for f_ in list(futs.values()):
                f_.cancel()
            pool.shutdown(wait=True)
            for s_ in (s_parse, s_m, s_d):
                cur.wait_stream(s_)
            torch.cuda.synchronize(self.device)
            index.close()
            if timing and len(timing) > 2:
                import sys
                m = np.array(timing[1:]).mean(axis=0) * 1e3
                print('stream_json(device): buffers, streams and the frame index ready after %.2f ms; first window staged after %.2f ms'
                      % (1e3 * t_setup, 1e3 * timing[0][0] + 1e3 * t_setup), file=sys.stderr)
                print('stream_json(device) per window, ms: wait staging %.2f | queue parse %.2f | wait parse %.2f | launch compute %.2f | '
                      'wait previous results %.2f' % tuple(m), file=sys.stderr)
                worst = np.array(timing[1:]).max(axis=0) * 1e3
                print('  slowest window of each, ms:            wait staging %.2f | queue parse %.2f | wait parse %.2f | launch compute %.2f | '
                      'wait previous results %.2f' % tuple(worst), file=sys.stderr)
                base = gpu_ev[2][0]
                for w in range(2, min(7, len(gpu_ev))):
                    p0, p1, c0, c1, h = gpu_ev[w]
                    print('  window %d on the GPU clock, ms: H2D %.2f .. %.2f, parse kernels .. %.2f | compute %.2f .. %.2f'
                          % (w, base.elapsed_time(p0), base.elapsed_time(h), base.elapsed_time(p1), base.elapsed_time(c0), base.elapsed_time(c1)), file=sys.stderr)

    # ---- device-side parse -----------------------------------------------
    def json_device_buffers(self, max_frames, text_cap=None):
        """Buffers of one in-flight window of the device-side parse: pinned + device staging, the device
        arena the arrays are parsed into, scratch and the totals (device + pinned mirror)."""
        B = int(max_frames)
        H = B * self.hpf
        if text_cap is None:
            text_cap = B * self.V * (64 + 96 * self.J * max(1, self.hpf // self.V) * 2)      # ~2x a full camera string
        text_cap = (int(text_cap) + 255) // 256 * 256
        return {'host': JsonStage(self.V, B, text_cap, 'pinned'), 'dev': JsonStage(self.V, B, text_cap, self.device),
                'arena': CapacityArena(self.V, self.J, B, H, self.device),
                'kcap': self.hpf,
                'scratch': torch.empty(int(self.lib.mpe_json_scratch_bytes(B * self.V, self.hpf, self.J)), dtype=torch.uint8, device=self.device),
                'totals': torch.zeros(4, dtype=torch.int32, device=self.device),
                'totals_host': torch.zeros(4, dtype=torch.int32).pin_memory(), 'ready': torch.cuda.Event()}

    def parse_json_device(self, bufs, n_frames, n_entries, used_bytes):
        """Queue (current stream): staging H2D (one copy of the used prefix) -> mpe_json_parse_device -> totals
        D2H into pinned memory; records bufs['ready'].  `finish_parse` turns the totals into a batch."""
        bufs['dev'].buf[:used_bytes].copy_(bufs['host'].buf[:used_bytes], non_blocking=True)
        if bufs.get('ev_h2d') is not None:
            bufs['ev_h2d'].record()
        arena = bufs['arena']
        out = L.mpe_batch()
        for name in ('frame_head_off', 'frame_en_off', 'slot_cam', 'slot_n', 'head_cam', 'joint_mask', 'tri_mask', 'xy', 'vp'):
            setattr(out, 'd_' + name, C.c_void_p(arena.ptr(name)))
        dev = bufs['dev']
        self._chk(self.lib.mpe_json_parse_device(self.ctx, self._stream(), C.c_void_p(dev.ptr('text')), C.c_void_p(dev.ptr('entries')),
                                                 C.c_void_p(dev.ptr('frame_entry_off')), n_entries, n_frames, arena.max_heads,
                                                 bufs['kcap'], _ptr(bufs['scratch']), bufs['scratch'].numel(), C.byref(out),
                                                 C.c_void_p(arena.ptr('skeleton_index')), _ptr(bufs['totals'])))
        bufs['totals_host'].copy_(bufs['totals'], non_blocking=True)
        bufs['ready'].record()
        bufs['n_frames'] = n_frames

    def finish_parse(self, bufs):
        """Wait for the parse of `bufs` and return the batch (ParsedOnDevice), or None if the window holds
        something the device parser leaves to the host packer (status bit 0) -- the caller then packs it
        on the host.  More heads than the arena holds raises like the host packer does."""
        bufs['ready'].synchronize()
        n_heads, n_en, status, max_h = (int(x) for x in bufs['totals_host'].tolist())
        if status & 1:                       # first: a window the device declines is sized (and refused, if need be) by the host packer
            return None
        if status & 2:
            raise ValueError('device-side parse: %d skeletons exceed the arena capacity %d' % (n_heads, bufs['arena'].max_heads))
        return ParsedOnDevice(bufs['arena'], self.V, self.J, bufs['n_frames'], n_heads, n_en, max_h)

    def pack_json_device(self, text, frame_start=0, frame_step=1, max_frames=0, n_threads=0):
        """One window of a document through the device-side parser (tests, one-off calls): JSON bytes ->
        ParsedOnDevice, or None where the host packer has to take over."""
        index = text if isinstance(text, JsonIndex) else JsonIndex(text)
        try:
            B = int(max_frames) if max_frames > 0 else self.max_frames
            if B > self.max_frames:
                raise ValueError('window of %d frames exceeds max_frames=%d' % (B, self.max_frames))
            size = len(index.text)
            bufs = self.json_device_buffers(B, text_cap=size + 16 * B * self.V + 256)
            try:
                nf, ne, used = stage_json_window(index, self.params, bufs['host'], frame_start, frame_step, B, n_threads)
            except NeedsHostParser:
                return None
            self.parse_json_device(bufs, nf, ne, used)
            pd = self.finish_parse(bufs)
            if pd is not None:
                pd.keep = bufs
            return pd
        finally:
            if index is not text:
                index.close()

    def run_pipelined(self, batches, mode='mlp', contexts=1):
        """Batches (DeviceBatch or PackedBatch) -> (poses, n_persons, persons) per batch, in order.

        contexts == 1: the two stages on their own streams: the matching stage of batch i+1 (GAT workspace) runs while
        the 3D stage of batch i (MLP workspace / DLT) is still in flight -- the two workspaces are disjoint, and
        concurrent kernels fill the tails of each other's dependent launch chains (+2-5 % throughput at 1000-frame
        batches).  contexts == K > 1: K contexts (sibling()) take turns on the batches, each on its own stream, so K whole
        batches are in flight (195.6 k against 186-189 k frames/s at K = 2; K = 3: 194.7 k).  Either way the results are
        the same bits as match() + mlp3d() / triangulate() called one after the other.  Each result is yielded once its
        3D stage has finished (with K contexts: when K - 1 later batches have been queued)."""
        K = max(1, int(contexts))
        engs = self.contexts(K)
        two = K == 1
        s_match = [torch.cuda.Stream(self.device) for _ in range(K)]
        s_3d = [torch.cuda.Stream(self.device)] if two else s_match
        cur = torch.cuda.current_stream(self.device)
        pending = []
        try:
            for i, b in enumerate(batches):
                e, sm, sd = engs[i % K], s_match[i % K], s_3d[i % K]
                db = self.to_device(b)
                # whatever produced this batch (an upload the iterator queued on the current stream) is ordered
                # before its matching stage -- per batch, not once in front of the loop
                sm.wait_stream(cur)
                with torch.cuda.stream(sm):
                    _, persons, n_persons = e.match(db, want_scores=False)
                    ev = torch.cuda.Event()
                    ev.record(sm)
                with torch.cuda.stream(sd):
                    if two:
                        sd.wait_event(ev)
                    poses = (e.mlp3d(db, persons, n_persons) if mode == 'mlp' else e.triangulate(db, persons, n_persons))[0]
                    done = torch.cuda.Event()
                    done.record(sd)
                # tensors allocated on one stream and consumed on another: the allocator must not hand
                # their memory out again before the consumer is done
                if two:
                    for t_ in (persons, n_persons):
                        t_.record_stream(sd)
                pending.append((done, poses, n_persons, persons, db))   # db: keeps the batch alive while it is in flight
                if len(pending) > K:
                    prev = pending.pop(0)
                    prev[0].synchronize()
                    yield prev[1:]
            while pending:
                last = pending.pop(0)
                last[0].synchronize()
                yield last[1:]
        finally:
            # also on early exit (the consumer closed the generator): nothing of ours is still running on
            # the side streams when the caller's stream goes on, and nothing in flight is freed under them
            for s_ in set(s_match + s_3d):
                cur.wait_stream(s_)
            for p_ in pending:
                p_[0].synchronize()

    def to_device(self, pb):
        if isinstance(pb, DeviceBatch):
            return pb
        assert isinstance(pb, PackedBatch)
        self.check_capacity(pb)
        return pb.to(self.device)

    def check_capacity(self, pb):
        """The kernels size their LDS / scratch from the capacities given at construction; a
        batch beyond them must be rejected here (the C ABI cannot see per-frame counts)."""
        if pb.n_frames > self.max_frames:
            raise ValueError('batch of %d frames exceeds max_frames=%d' % (pb.n_frames, self.max_frames))
        if pb.n_frames == 1 and pb.n_heads <= self.hpf and pb.V == self.V and pb.J == self.J and getattr(pb, 'en_pair', None) is None:
            return                           # (one frame of the per-frame mirrors: nothing below can fail)
        if pb.n_frames and pb.max_heads_per_frame() > self.hpf:
            raise ValueError('a frame holds %d skeletons, capacity is %d (raise max_persons_per_camera / '
                             'max_heads_per_frame)' % (pb.max_heads_per_frame(), self.hpf))
        if pb.V != self.V or pb.J != self.J:
            raise ValueError('batch packed for %d cameras x %d joints, engine built for %d x %d'
                             % (pb.V, pb.J, self.V, self.J))
        if getattr(pb, 'en_pair', None) is not None and pb.n_frames:
            cap = explicit_m_cap(self.hpf)
            if cap == 0:
                raise ValueError('explicit edge-node lists need max_heads_per_frame <= 1024 and 16-bit node ids (capacity %d)' % self.hpf)
            if pb.max_edge_nodes_per_frame() > cap:
                raise ValueError('a graph holds %d edge-nodes, an engine with max_heads_per_frame = %d takes %d per frame'
                                 % (pb.max_edge_nodes_per_frame(), self.hpf, cap))
            if pb.n_edge_nodes > self.max_frames * self.m_frame:
                raise ValueError('batch of %d edge-nodes exceeds the capacity %d (max_edge_nodes_per_frame)'
                                 % (pb.n_edge_nodes, self.max_frames * self.m_frame))
            ep = np.asarray(pb.en_pair).reshape(-1, 2)
            H = np.repeat(np.diff(pb.frame_head_off), np.diff(pb.frame_en_off))
            if ep.size and (ep.min() < 0 or (ep >= H[:, None]).any() or (ep[:, 0] == ep[:, 1]).any()):
                raise ValueError('explicit edge-node list: a pair lies outside its frame or joins a head with itself')

    def _stream(self):
        # torch.cuda.current_stream() builds a Stream object (~80 us on the GPU box's host; ten calls per frame in the
        # one-frame-per-call mirrors); the raw handle is what the C ABI takes
        try:
            return C.c_void_p(torch._C._cuda_getCurrentRawStream(self.device.index or 0))
        except AttributeError:
            return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def match(self, db, want_scores=True):
        """-> (scores[n_edge_nodes] f32 or None, persons[B,Pcap,V] i32, n_persons[B] i32)."""
        B = db.n_frames
        scores = torch.empty(max(db.n_edge_nodes, 1), dtype=torch.float32, device=self.device) if want_scores else None
        persons = torch.empty((B, self.pcap, self.V), dtype=torch.int32, device=self.device)
        n_persons = torch.empty((B,), dtype=torch.int32, device=self.device)
        self._chk(self.lib.mpe_match_batch(self.ctx, self._stream(), C.byref(db.struct),
                                           _ptr(scores), _ptr(persons), _ptr(n_persons)))
        return (scores[:db.n_edge_nodes] if want_scores else None), persons, n_persons

    def gat_scores(self, db, heads=False, feats=None):
        """GAT2.forward; `feats` (optional) = dense [n_nodes, F] rows supplied by the caller."""
        sc = torch.empty(max(db.n_edge_nodes, 1), dtype=torch.float32, device=self.device)
        sh = torch.empty(max(db.n_heads, 1), dtype=torch.float32, device=self.device) if heads else None
        ld = 0
        if feats is not None:
            feats = feats.to(self.device, torch.float32).contiguous()
            ld = feats.shape[1]
        self._chk(self.lib.mpe_gat_forward(self.ctx, self._stream(), C.byref(db.struct), _ptr(feats), ld,
                                           _ptr(sc), _ptr(sh)))
        return (sc[:db.n_edge_nodes], sh[:db.n_heads]) if heads else sc[:db.n_edge_nodes]

    def gat_scores_joined(self, db, feats=None):
        """GAT2.forward of a one-graph batch in the node order of the reference's output (heads, then edge-nodes): both halves are
        written into ONE [n_heads + n_edge_nodes] buffer (the last layer's rows are single floats: no alignment beyond 4 bytes is
        asked of either half).  -> (all scores, the edge-node part as a view)."""
        H, M = db.n_heads, db.n_edge_nodes
        out = torch.empty(H + M, dtype=torch.float32, device=self.device)
        ld = 0
        if feats is not None:
            feats = feats.to(self.device, torch.float32).contiguous()
            ld = feats.shape[1]
        base = out.data_ptr()
        self._chk(self.lib.mpe_gat_forward(self.ctx, self._stream(), C.byref(db.struct), _ptr(feats), ld,
                                           C.c_void_p(base + 4 * H), C.c_void_p(base)))
        return out, out[H:]

    def set_gat_output(self, sigmoid=True):
        """Last-layer activation: sigmoid (deployed model) or identity (final_activation=None)."""
        if self._state.get('gat_output', True) == bool(sigmoid):
            return
        self._chk(self.lib.mpe_set_gat_output(self.ctx, 1 if sigmoid else 2))
        self._state['gat_output'] = bool(sigmoid)
        for e in self._siblings:
            e.set_gat_output(sigmoid)

    def gat_layer(self, db, layer, x, activation=0):
        """One GraphAttention2 layer + the activation GAT2.forward applies (mpe_gat_layer).
        x [n_nodes, in_dim] rows in node order -> [n_nodes, heads*out_dim]."""
        x = x.to(self.device, torch.float32).contiguous()
        _, nh, od = self.gat_dims[layer]
        out = torch.empty((x.shape[0], nh * od), dtype=torch.float32, device=self.device)
        self._chk(self.lib.mpe_gat_layer(self.ctx, self._stream(), C.byref(db.struct), layer, _ptr(x), x.shape[1],
                                         _ptr(out), out.shape[1], int(activation)))
        return out

    def edge_softmax_aggregate(self, db, layer, ft2):
        """The DGL half of a layer (gat2.py:57-66): ft2 [n_nodes, heads*out_dim] -> aggregated rows."""
        ft2 = ft2.to(self.device, torch.float32).contiguous()
        out = torch.empty_like(ft2)
        self._chk(self.lib.mpe_edge_softmax_aggregate(self.ctx, self._stream(), C.byref(db.struct), layer, _ptr(ft2),
                                                      ft2.shape[1], _ptr(out), out.shape[1]))
        return out

    def sync_status(self):
        """Synchronise and raise MpeError(MPE_ERR_CAPACITY) if a frame of a batch since the last
        call exceeded max_heads_per_frame (detected on the device)."""
        self._chk(self.lib.mpe_sync_status(self.ctx, self._stream()))

    def status_queue(self):
        """First half of sync_status: the read-back of the status word joins what is queued so far (mpe_status_queue)."""
        self._chk(self.lib.mpe_status_queue(self.ctx, self._stream()))

    def status_wait(self):
        """Second half: synchronise and raise what the read-back saw (mpe_status_wait)."""
        self._chk(self.lib.mpe_status_wait(self.ctx, self._stream()))

    def set_threshold(self, thr):
        if self._state.get('threshold', self._made_with['threshold']) == float(thr):
            return                          # (the C call re-uploads the whole device-side configuration, synchronously)
        self._chk(self.lib.mpe_set_threshold(self.ctx, float(thr)))
        self._state['threshold'] = float(thr)
        for e in self._siblings:
            e.set_threshold(thr)

    def cluster(self, db, scores):
        B = db.n_frames
        scores = scores.to(self.device, torch.float32).contiguous()
        persons = torch.empty((B, self.pcap, self.V), dtype=torch.int32, device=self.device)
        n_persons = torch.empty((B,), dtype=torch.int32, device=self.device)
        self._chk(self.lib.mpe_cluster_batch(self.ctx, self._stream(), C.byref(db.struct), _ptr(scores),
                                             _ptr(persons), _ptr(n_persons)))
        return persons, n_persons

    def dense_rows(self, db):
        """graph.ndata['h'] of a one-frame batch: the dense [N, 2 + V*J*10] rows in node order (mpe_dense_rows)."""
        F = 2 + self.V * self.J * 10
        out = torch.empty((db.n_heads + db.n_edge_nodes, F), dtype=torch.float32, device=self.device)
        self._chk(self.lib.mpe_dense_rows(self.ctx, self._stream(), C.byref(db.struct), _ptr(out), F))
        return out

    def head_features(self, db):
        out = torch.empty((max(db.n_heads, 1), self.J, 10), dtype=torch.float32, device=self.device)
        self._chk(self.lib.mpe_head_features(self.ctx, self._stream(), C.byref(db.struct), _ptr(out)))
        return out[:db.n_heads]

    def mlp_input_rows(self, db, persons, n_persons):
        B = db.n_frames
        width = self.V * self.J * self.params.numbers_per_joint
        ld = (width + 127) // 128 * 128
        rows = torch.zeros((B * self.pcap, ld), dtype=torch.float32, device=self.device)
        valid = torch.zeros((B * self.pcap,), dtype=torch.uint8, device=self.device)
        self._chk(self.lib.mpe_mlp_input_rows(self.ctx, self._stream(), C.byref(db.struct), _ptr(persons),
                                              _ptr(n_persons), _ptr(rows), ld, _ptr(valid)))
        return rows.view(B, self.pcap, ld)[:, :, :width], valid.view(B, self.pcap)

    def mlp_forward(self, x):
        """x [m, in_dim] f32 (device) -> [m, out_dim]."""
        m, k = x.shape
        ld = (k + 127) // 128 * 128
        stream = self._stream()
        st = self.__dict__.get('_mlp_stage')
        if m <= 64 and (st is None or st[0].shape[1] != ld or st[1] != stream.value):
            # small batches (the one-frame-per-call mirrors): one padded staging buffer per engine and stream, zeroed once -- the pad
            # columns stay zero, a call costs one copy instead of a fill and a copy
            st = self.__dict__['_mlp_stage'] = (torch.zeros((64, ld), dtype=torch.float32, device=self.device), stream.value)
        if m <= 64:
            xp = st[0][:m]
            xp[:, :k].copy_(x)
        else:
            xp = torch.zeros((m, ld), dtype=torch.float32, device=self.device)
            xp[:, :k] = x
        y = torch.empty((m, self.mlp_out), dtype=torch.float32, device=self.device)
        self._chk(self.lib.mpe_mlp_forward(self.ctx, stream, _ptr(xp), ld, m, _ptr(y), self.mlp_out))
        return y

    def mlp3d(self, db, persons, n_persons):
        """-> (poses[B,Pcap,J,3] f32 metres, valid[B,Pcap] u8)."""
        B = db.n_frames
        poses = torch.empty((B, self.pcap, self.J, 3), dtype=torch.float32, device=self.device)
        valid = torch.empty((B, self.pcap), dtype=torch.uint8, device=self.device)
        self._chk(self.lib.mpe_mlp3d_batch(self.ctx, self._stream(), C.byref(db.struct), _ptr(persons),
                                           _ptr(n_persons), _ptr(poses), _ptr(valid)))
        return poses, valid

    def triangulate(self, db, persons, n_persons, all_joints=False, positive_ids_only=False):
        """-> (poses[B,Pcap,J,3] f64, joint_valid[B,Pcap,J] u8).  positive_ids_only: the gather of
        test/reprojection_error.py:296-300 (joints with values[0] > 0 only)."""
        B = db.n_frames
        poses = torch.empty((B, self.pcap, self.J, 3), dtype=torch.float64, device=self.device)
        jv = torch.empty((B, self.pcap, self.J), dtype=torch.uint8, device=self.device)
        self._chk(self.lib.mpe_triangulate_batch(self.ctx, self._stream(), C.byref(db.struct), _ptr(persons),
                                                 _ptr(n_persons), _ptr(poses), _ptr(jv),
                                                 (1 if all_joints else 0) | (2 if positive_ids_only else 0)))
        return poses, jv

    def dlt_pairs(self, pts, cams):
        pts = torch.as_tensor(pts, dtype=torch.float64, device=self.device).contiguous()
        cams = torch.as_tensor(cams, dtype=torch.int32, device=self.device).contiguous()
        n = pts.shape[0]
        out = torch.empty((n, 3), dtype=torch.float64, device=self.device)
        self._chk(self.lib.mpe_dlt_pairs(self.ctx, self._stream(), _ptr(pts), _ptr(cams), n, _ptr(out)))
        return out

    def set_precision(self, gat_acc64=False, mlp_acc64=True, mlp_bf16=False, gat_reduced=False, attn_fp16=False, mlp_split=None,
                      gat_split=None, mlp_max_accuracy=False, mlp_f64=False):
        """GAT: plain fp32 MFMA chain / f64 running sums / `attn_fp16` (BASELINE configs[4] as worded: the transformed
        features ft2 travel to the attention stage as fp16 rows, the GEMMs stay fp32) / `gat_reduced` (additionally
        bf16 MFMA for fc1/fc2); MLP: fp32 / f64 running sums (default, parity) / bf16 MFMA.  The reduced modes are
        never used on the parity path."""
        # gat_split (default True): fc1 / fc2 of the layers >= 1 in the split-bf16 form (fp32-accurate, bf16 matrix pipe);
        # gat_split=False = the fp32 MFMA of rounds 1-3.  Layer 0 always runs on the fp32 MFMA (head rows only).
        if gat_split is None:
            gat_split = not gat_reduced
        gat = 2 if gat_reduced else 3 if attn_fp16 else int(gat_acc64)
        if gat_split and gat != 2:
            gat = {0: 4, 1: 5, 3: 6}[gat]
        # The parity MLP (mlp_acc64=True, not bf16) has two forms of the same accuracy class: the library default
        # (mlp_split=None / True: fp32 operands as three bf16 planes, six products on the bf16 matrix pipe, f64 sums every
        # second K stage; csrc/gemm_sb16.hip) and the fp32 MFMA with f64 sums per stage of rounds 1-3 (mlp_split=False).
        # mlp_max_accuracy: the split form with an f64 flush after EVERY K stage (MLP mode 4): rms error of a launch 0.13-0.18
        # instead of 0.24-0.26 ulp of its output scale, the MLP launches ~7 % slower.
        if mlp_split is None:
            mlp_split = bool(mlp_acc64) and not mlp_bf16
        # mlp_f64: the f64-evaluated network (MLP mode 5; LeakyReLU slope as the decimal double, include/mpe.h): exact products, f64 accumulation on the f64 matrix pipe; several times slower.
        if mlp_max_accuracy and (mlp_bf16 or not mlp_split or mlp_f64):
            raise ValueError('mlp_max_accuracy is a mode of the split-bf16 MLP (mlp_acc64=True, not mlp_bf16, mlp_split not False)')
        if mlp_f64 and mlp_bf16:
            raise ValueError('mlp_f64 and mlp_bf16 exclude each other')
        self._chk(self.lib.mpe_set_precision(self.ctx, gat, 5 if mlp_f64 else 2 if mlp_bf16 else 4 if mlp_max_accuracy else 3 if mlp_split else int(mlp_acc64)))
        self._state['precision'] = (gat_acc64, mlp_acc64, mlp_bf16, gat_reduced, attn_fp16, mlp_split, gat_split, mlp_max_accuracy, mlp_f64)
        for e in self._siblings:
            e.set_precision(gat_acc64, mlp_acc64, mlp_bf16, gat_reduced, attn_fp16, mlp_split, gat_split, mlp_max_accuracy, mlp_f64)

    def linear(self, x, w, b, slope=None, acc64=False, split=False, split_f64=True, split_flush_per_stage=False, f64mm=False):
        """act(x @ w.T + b) through the MFMA GEMM (parity tests). x device [m,k]; w,b host.  split: the split-bf16
        arithmetic of csrc/gemm_sb16.hip."""
        w = _np32(w)
        b = _np32(b)
        n, k = w.shape
        dw, dbias, ldw = C.c_void_p(), C.c_void_p(), C.c_int32()
        self._chk(self.lib.mpe_upload_linear(self.ctx, _f32p(w), _f32p(b), n, k, C.byref(dw), C.byref(dbias),
                                             C.byref(ldw)))
        try:
            m = x.shape[0]
            xp = torch.zeros((m, ldw.value), dtype=torch.float32, device=self.device)
            xp[:, :k] = x
            ldc = (n + 3) // 4 * 4
            y = torch.empty((m, ldc), dtype=torch.float32, device=self.device)
            self._chk(self.lib.mpe_linear(self.ctx, self._stream(), _ptr(xp), ldw.value, dw, ldw.value, dbias,
                                          _ptr(y), ldc, m, None, n, k, (0 if slope is None else 1) | (2 if acc64 else 0) | (4 if split else 0) | (0 if split_f64 else 8) | (16 if split_flush_per_stage else 0) | (32 if f64mm else 0),
                                          0.0 if slope is None else float(slope)))
            torch.cuda.synchronize(self.device)
            return y[:, :n].contiguous()
        finally:
            self.lib.mpe_free_device(self.ctx, dw)
            self.lib.mpe_free_device(self.ctx, dbias)

    # ---- profiling --------------------------------------------------------------------
    def profile(self, on, resume=False):
        """Per-GEMM HIP event pairs on/off.  resume=True keeps the records taken so far (sampling: events
        on every n-th batch only, since the event packets cost ~2 % of a step)."""
        self._chk(self.lib.mpe_profile_enable(self.ctx, (2 if resume else 1) if on else 0))

    def profile_read(self):
        ms, fl, n, tot = C.c_double(), C.c_double(), C.c_int64(), C.c_double()
        self._chk(self.lib.mpe_profile_read(self.ctx, C.byref(ms), C.byref(fl), C.byref(n), C.byref(tot)))
        out = {'gemm_ms': ms.value, 'gemm_flop': fl.value, 'gemm_launches': n.value}
        self._chk(self.lib.mpe_profile_read_split(self.ctx, C.byref(ms), C.byref(fl), C.byref(n)))
        out.update({'split_ms': ms.value, 'split_flop': fl.value, 'split_launches': n.value})    # MLP launches on the bf16 MFMA
        self._chk(self.lib.mpe_profile_read_bf16(self.ctx, C.byref(ms), C.byref(fl), C.byref(n)))
        out.update({'bf16_ms': ms.value, 'bf16_flop': fl.value, 'bf16_launches': n.value})       # plain bf16 launches (reduced modes)
        return out


def explicit_m_cap(max_heads_per_frame):
    """Edge-nodes a frame of an explicit edge-node list may hold on a context of that capacity (include/mpe.h: the power of
    two >= max(512, hmax^2 / 2 + 1), the clustering scratch; 0 = mode unavailable)."""
    need, n = max_heads_per_frame * max_heads_per_frame // 2 + 1, 512
    while n < need:
        n <<= 1
    return n if max_heads_per_frame <= 1024 and max_heads_per_frame + n <= 65535 else 0


def _np32(v):
    if isinstance(v, torch.Tensor):
        v = v.detach().cpu().numpy()
    return np.ascontiguousarray(v, dtype=np.float32)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())
