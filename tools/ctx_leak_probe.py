"""Device memory across create / use / destroy cycles of an Engine (mpe_create ... mpe_destroy): free memory before and after N cycles of a
24-frame context that runs a one-frame and a 24-frame batch through matching + both 3D stages (and, with PROBE_JSON=1, a JSON window through
the device-side parse).  PROBE_N cycles (15).  Round 6: delta 0.0 MB over 45 cycles without the JSON pipeline; with it a one-time ~200 MB
(page-locked staging the process keeps) and nothing that grows with the cycles (202 MB after 15, 218 MB after 45)."""
import importlib, sys, torch, gc
sys.path.insert(0, '.')
PKG='3d_multi_pose_estimator_amd'
syn=importlib.import_module(PKG+'.synthetic'); cal=importlib.import_module(PKG+'.calibration'); par=importlib.import_module(PKG+'.parameters'); pipeline=importlib.import_module(PKG+'.pipeline')
calib=cal.Calibration(par.parameters)
gat=syn.gat_state_dict(7, 902, logit_gain=25.0, logit_shift=0.948); prm=syn.gat_params(902); mlp=syn.mlp_state_dict(11,1260)
frames=[syn.make_frame(calib, 500+i, syn.FrameSpec(persons=3))[0] for i in range(24)]
frames=[{c:[f[c][0],f[c][1]] for c in f} for f in frames]
import os
JSON=os.environ.get('PROBE_JSON','1')=='1'
N=int(os.environ.get('PROBE_N','15'))
def cycle(n):
    eng=pipeline.Engine(par.parameters, calib, max_frames=n, max_persons_per_camera=6)
    eng.load_gat(gat, prm); eng.load_mlp(mlp)
    for k in (1, n):
        db=eng.to_device(eng.pack(frames[:k])); _,p,c=eng.match(db, want_scores=False); eng.mlp3d(db,p,c); eng.triangulate(db,p,c); eng.sync_status()
    if JSON: list(eng.stream_json(__import__('json').dumps([syn.make_frame(calib, 1, syn.FrameSpec(persons=2))[0]]*n).encode(), chunk_frames=n))
    eng.close(); del eng
torch.cuda.init()
cycle(24); gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()
free0,_=torch.cuda.mem_get_info()
for i in range(N):
    cycle(24)
gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()
free1,_=torch.cuda.mem_get_info()
print('free before %.1f MB, after %d create/use/destroy cycles %.1f MB, delta %.1f MB' % (free0/2**20, N, free1/2**20, (free0-free1)/2**20))
import resource; print('host maxrss MB', resource.getrusage(resource.RUSAGE_SELF).ru_maxrss/1024)
