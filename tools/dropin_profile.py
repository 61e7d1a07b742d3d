"""Where does a frame of the one-frame-per-call loop (harness/dropin_loop.py: the reference's test/metrics_from_model.py:178-294 over the
drop-in mirrors) spend its host time?  Per mirror symbol (the loop's own clock) and per Python function (cProfile over the same frames).
  python tools/dropin_profile.py [frames] [out.txt]"""
import cProfile
import importlib
import io
import json
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = '3d_multi_pose_estimator_amd'


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    syn = importlib.import_module(PKG + '.synthetic')
    cal = importlib.import_module(PKG + '.calibration')
    par = importlib.import_module(PKG + '.parameters')
    loop = importlib.import_module(PKG + '.harness.dropin_loop')
    params = par.select('PANOPTIC')
    calib = cal.Calibration(params, None)
    V, J = len(params.used_cameras_skeleton_matching), len(params.joint_list)
    nf = 2 + V * J * 10
    gat_sd = syn.gat_state_dict(7, nf, logit_gain=25.0, logit_shift=0.698 + 0.25)       # bench.py's networks and frames
    prm = syn.gat_params(nf)
    mlp_sd = syn.mlp_state_dict(11, len(params.cameras) * J * params.numbers_per_joint)
    wire = [syn.make_frame(calib, i, syn.FrameSpec(persons=4))[0] for i in range(50)]
    matcher, lifter = loop.build_models(gat_sd, prm, mlp_sd)
    device = torch.device('cuda', 0)
    frames = [wire[i % 50] for i in range(n + 10)]
    loop.run(frames[:30], matcher, lifter, warmup=10, device=device)
    out = []
    for rep in range(3):
        res = loop.run(frames, matcher, lifter, warmup=10, device=device)
        res.pop('last')
        out.append('run %d: %s' % (rep, json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in res.items() if k != 'reference_readme_ms'})))
    pr = cProfile.Profile()
    pr.enable()
    loop.run(frames, matcher, lifter, warmup=10, device=device)
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).strip_dirs().sort_stats('cumulative').print_stats(45)
    out.append('cProfile (the profiler itself roughly doubles the Python share), %d frames:' % n)
    out.append(s.getvalue())
    s = io.StringIO()
    pstats.Stats(pr, stream=s).strip_dirs().sort_stats('tottime').print_stats(30)
    out.append(s.getvalue())
    text = '\n'.join(out)
    print(text)
    if len(sys.argv) > 2:
        with open(sys.argv[2], 'w') as fh:
            fh.write(text + '\n')


if __name__ == '__main__':
    main()
