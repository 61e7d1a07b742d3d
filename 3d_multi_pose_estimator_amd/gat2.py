"""API mirror of the reference's skeleton_matching/gat2.py.

``GAT2`` keeps the constructor signature, parameter names (so the reference's
``skeleton_matching.tch`` state dict loads unchanged) and call convention
``model(feats, g) -> N x 1 x 1`` (reference gat2.py:91-149), but forward() is one call into
libmpe_hip.so (mpe_gat_forward): fp32 MFMA GEMMs for fc1/fc2, wavefront-level edge softmax and
aggregation on the implicit topology (or the explicit edge-node list of a generated scene, graph_generator.py:672-810).  Dropout and the residual branch are inactive in the
deployed model (train_skeleton_matching.py:49-52) and are rejected if requested.
"""
import torch
import torch.nn as nn

from . import runtime


class GraphAttention2(nn.Module):
    """Parameter container with the reference's names (gat2.py:18-48)."""

    def __init__(self, g, in_dim, out_dim, num_heads, feat_drop, attn_drop, alpha, residual=False, name=None,
                 bias=False):
        super().__init__()
        if feat_drop or attn_drop or residual:
            raise NotImplementedError('dropout / residual are not part of the deployed inference path')
        self.g = g
        self.num_heads = num_heads
        self.name = name
        self.alpha = alpha
        self.fc1 = nn.Linear(in_dim, in_dim, bias=bias)
        self.fc2 = nn.Linear(in_dim, num_heads * out_dim, bias=bias)
        self.attn_l = nn.Parameter(torch.empty(num_heads, out_dim, 1))
        self.attn_r = nn.Parameter(torch.empty(num_heads, out_dim, 1))
        for p in (self.fc1.weight, self.fc2.weight, self.attn_l, self.attn_r):
            nn.init.xavier_normal_(p.data, gain=1.414)


class GAT2(nn.Module):
    def __init__(self, g, num_layers, in_dim, num_classes, num_hidden, heads, activation, final_activation,
                 feat_drop, attn_drop, alpha, residual, bias=False):
        super().__init__()
        self.g = g
        self.num_layers = num_layers
        self.activation = activation
        self.final_activation = final_activation
        self.alpha = alpha
        self.heads = list(heads)
        self.num_hidden = list(num_hidden)
        self.layers = nn.ModuleList()
        self.layers.append(GraphAttention2(g, in_dim, num_hidden[0], heads[0], feat_drop, attn_drop, alpha, False,
                                           '0', bias))
        for l in range(1, num_layers - 1):
            self.layers.append(GraphAttention2(g, num_hidden[l - 1] * heads[l - 1], num_hidden[l], heads[l],
                                               feat_drop, attn_drop, alpha, residual, str(l), bias))
        self.layers.append(GraphAttention2(g, num_hidden[-1] * heads[-1], num_classes, 1, feat_drop, attn_drop,
                                           alpha, residual, 'X', bias))
        self._engine = None
        self._version = None

    def set_g(self, g):
        self.g = g
        for layer in self.layers:
            layer.g = g

    def _state_version(self):
        # (walking the module tree for every forward cost 0.13 ms per frame in the one-frame-per-call loop: runtime.ParamWatch keeps
        # the walk and checks per call that it still describes the tree)
        w = self.__dict__.get('_param_watch')
        if w is None:
            w = self.__dict__['_param_watch'] = runtime.ParamWatch(self)
        return w.version()

    def _ensure_engine(self, n_graphs=1):
        ver = self._state_version()
        if self._engine is not None and ver == self._version and self._engine.max_frames >= n_graphs:
            return self._engine
        slope = getattr(self.activation, 'negative_slope', None)
        if not isinstance(self.activation, nn.LeakyReLU) or slope is None:
            raise NotImplementedError('hidden activation must be nn.LeakyReLU (train_skeleton_matching.py:54)')
        if self.final_activation is not None and not isinstance(self.final_activation, nn.Sigmoid):
            raise NotImplementedError('final activation must be nn.Sigmoid or None')
        if self._engine is not None:
            self._engine.close()
        eng = runtime.new_engine(max_frames=max(1, n_graphs))      # a batch of graphs = a batch of frames of the engine
        sd = {k: v.detach().cpu() for k, v in self.state_dict().items()}
        for l, layer in enumerate(self.layers):         # bias=False models: zero biases
            for fc in ('fc1', 'fc2'):
                key = 'layers.%d.%s.bias' % (l, fc)
                if key not in sd:
                    sd[key] = torch.zeros(getattr(layer, fc).out_features)
        prm = {'gnn_layers': self.num_layers, 'heads': self.heads, 'alpha': self.alpha, 'nonlinearity': slope}
        eng.load_gat(sd, prm)
        self._engine, self._version = eng, ver
        return eng

    def forward(self, inputs, g):
        self.set_g(g)
        eng = self._ensure_engine(getattr(g, 'batch_size', 1))
        db = g.device_batch(eng)
        own = dict.get(g.ndata, 'h')            # the graph's own dense rows, if the caller ever asked for them
        feats = None
        if inputs is not None and (own is None or inputs.data_ptr() != own.data_ptr()):
            # caller-made features: honour them (dense path); the graph's own rows are
            # re-derived on the device from the packed skeletons instead
            if not (own is not None and inputs.shape == own.shape and torch.equal(inputs.to(own.device).float(), own)):
                feats = inputs
        one = getattr(g, 'batch_size', 1) == 1
        if one and feats is None and runtime.prefetch_enabled():
            fs = runtime.queued_scores(g, eng, self.final_activation is not None)
            if fs is not None:           # this engine already scored the frame when the dataset was built (runtime.start_frame)
                g._ahead = fs.ahead
                return fs.out
        eng.set_gat_output(sigmoid=self.final_activation is not None)     # None: raw logits (gat2.py:146-148)
        if one and db.n_edge_nodes > 0 and getattr(g.packed, 'en_pair', None) is None:
            out, sc = eng.gat_scores_joined(db, feats=feats)
            out = out.reshape(-1, 1, 1)
            # the reference's caller clusters these scores next and builds one MLP row per person (metrics_from_model.py:214-277):
            # both are queued behind the scores now (runtime.queue_proposals) and used only if the caller does exactly that
            g._ahead = None
            if runtime.prefetch_enabled():
                g._ahead = runtime.queue_proposals(eng, db, sc, out)
                runtime.note_matcher(eng)
            return out
        sc, sh = eng.gat_scores(db, heads=True, feats=feats)
        # A graph beyond the engine's per-frame capacity must raise instead of scoring 0.  For the implicit topology of a packed
        # frame that was decided on the host (FrameGraph.device_batch -> Engine.check_capacity raises before anything is
        # launched); what only the device can see -- an explicit edge-node list whose repeated pairs overflow a head's in-degree
        # -- needs the status word, i.e. a stream synchronisation (0.2 ms per frame in the one-frame-per-call loop).
        if getattr(g.packed, 'en_pair', None) is not None or getattr(g, 'batch_size', 1) != 1:
            eng.sync_status()
        if one:
            return torch.cat([sh, sc]).reshape(-1, 1, 1)
        # a batch of graphs (graph_generator.batch = the reference's dgl.batch): node order graph by graph, heads then edge-nodes
        parts, h0, e0 = [], 0, 0
        for H, M in zip(g.batch_num_heads, g.batch_num_edge_nodes):
            parts += [sh[h0:h0 + H], sc[e0:e0 + M]]
            h0, e0 = h0 + H, e0 + M
        return torch.cat(parts).reshape(-1, 1, 1)
