"""Host microseconds per frame of the pieces inside the drop-in mirrors (harness/dropin_loop.py), measured with thin timing wrappers
(no profiler: cProfile doubles the Python share).   python tools/dropin_steps.py [frames] [out.txt]"""
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = '3d_multi_pose_estimator_amd'
ACC = {}


def wrap(owner, name, label=None):
    fn = getattr(owner, name)
    label = label or '%s.%s' % (getattr(owner, '__name__', owner), name)

    def timed(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            e = ACC.setdefault(label, [0.0, 0])
            e[0] += time.perf_counter() - t0
            e[1] += 1
    setattr(owner, name, timed)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    m = lambda x: importlib.import_module(PKG + '.' + x)
    syn, cal, par, loop, packing, gg, rt, pl = m('synthetic'), m('calibration'), m('parameters'), m('harness.dropin_loop'), m('packing'), m('graph_generator'), m('runtime'), m('pipeline')
    params = par.select('PANOPTIC')
    calib = cal.Calibration(params, None)
    V, J = len(params.used_cameras_skeleton_matching), len(params.joint_list)
    nf = 2 + V * J * 10
    gat_sd = syn.gat_state_dict(7, nf, logit_gain=25.0, logit_shift=0.698 + 0.25)
    prm = syn.gat_params(nf)
    mlp_sd = syn.mlp_state_dict(11, len(params.cameras) * J * params.numbers_per_joint)
    wire = [syn.make_frame(calib, i, syn.FrameSpec(persons=4))[0] for i in range(50)]
    matcher, lifter = loop.build_models(gat_sd, prm, mlp_sd)
    device = torch.device('cuda', 0)
    frames = [wire[i % 50] for i in range(n + 10)]
    loop.run(frames[:30], matcher, lifter, warmup=10, device=device)
    wrap(packing, 'pack_views', 'pack_views')
    wrap(gg, '_pack_one', '_pack_one')
    wrap(gg.FrameGraph, '__init__', 'FrameGraph.__init__')
    wrap(gg.FrameGraph, 'device_batch', 'FrameGraph.device_batch')
    wrap(gg.FrameGraph, '_dense_features', 'FrameGraph._dense_features')
    wrap(rt, 'start_frame', 'runtime.start_frame')
    wrap(rt, 'queue_proposals', 'runtime.queue_proposals')
    wrap(rt, 'take_proposals', 'runtime.take_proposals')
    wrap(rt.ParamWatch, 'version', 'ParamWatch.version')
    for name in ('gat_scores', 'sync_status', 'status_wait', 'mlp_forward', 'dense_rows', 'check_capacity'):
        wrap(pl.Engine, name, 'Engine.' + name)
    wrap(packing.DeviceBatch, '__init__', 'DeviceBatch.__init__')
    wrap(torch, 'cat', 'torch.cat')
    res = loop.run(frames, matcher, lifter, warmup=10, device=device)
    res.pop('last')
    lines = [json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in res.items() if k != 'reference_readme_ms'})]
    lines.append('%-32s %10s %8s' % ('piece (nested pieces overlap)', 'us/frame', 'calls/f'))
    for k, (s, c) in sorted(ACC.items(), key=lambda kv: -kv[1][0]):
        lines.append('%-32s %10.1f %8.2f' % (k, 1e6 * s / (n + 10), c / (n + 10)))
    text = '\n'.join(lines)
    print(text)
    if len(sys.argv) > 2:
        open(sys.argv[2], 'w').write(text + '\n')


if __name__ == '__main__':
    main()
