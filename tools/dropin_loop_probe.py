"""The reference's per-frame loop over the drop-in mirrors (harness/dropin_loop.py), with a cProfile of where a frame's time
goes.  python tools/dropin_loop_probe.py [frames] [persons]"""
import cProfile, importlib, io, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = '3d_multi_pose_estimator_amd'
import torch
syn = importlib.import_module(PKG + '.synthetic'); cal = importlib.import_module(PKG + '.calibration')
par = importlib.import_module(PKG + '.parameters'); loop = importlib.import_module(PKG + '.harness.dropin_loop')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
P = int(sys.argv[2]) if len(sys.argv) > 2 else 4
params = par.parameters
calib = cal.Calibration(params)
V, J = len(params.used_cameras_skeleton_matching), len(params.joint_list)
nf = 2 + V * J * 10
gat_sd = syn.gat_state_dict(7, nf, logit_gain=25.0, logit_shift=0.698 + 0.25); prm = syn.gat_params(nf)
mlp_sd = syn.mlp_state_dict(11, len(params.cameras) * J * params.numbers_per_joint)
frames = [syn.make_frame(calib, i % 50, syn.FrameSpec(persons=P))[0] for i in range(n)]
model, mlp = loop.build_models(gat_sd, prm, mlp_sd)
out = loop.run(frames[:20], model, mlp, warmup=5)
out = loop.run(frames, model, mlp, warmup=5)
out.pop('last')
print(out)
if os.environ.get('MPE_PROBE_PROFILE', '1') == '1':
    pr = cProfile.Profile(); pr.enable(); loop.run(frames[:100], model, mlp, warmup=0); pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(38); print(s.getvalue()[:9000])
