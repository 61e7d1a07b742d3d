"""Engine.stream_json alone in a fresh process (few streams alive): frames/s and, with MPE_JSON_TIMING=1, the per-window
timeline.  python tools/json_stream_probe.py [frames] [chunk] [parser] [repeats] [contexts]
`repeats` > 1 runs the same call again in the same process: the first call of a fresh process finds a cool GPU."""
import importlib, json, os, sys, time
import os as _os; _os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')      # before HIP initialises: one hardware queue per busy stream (lib.py leaves the environment alone)
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = '3d_multi_pose_estimator_amd'
syn = importlib.import_module(PKG + '.synthetic'); cal = importlib.import_module(PKG + '.calibration')
par = importlib.import_module(PKG + '.parameters'); pipeline = importlib.import_module(PKG + '.pipeline')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
parser = sys.argv[3] if len(sys.argv) > 3 else 'device'
contexts = int(sys.argv[5]) if len(sys.argv) > 5 else 1
kw = {'contexts': contexts} if parser == 'device' else {}
calib = cal.Calibration(par.parameters)
uniq = [syn.make_frame(calib, i, syn.FrameSpec(persons=4))[0] for i in range(200)]
body = json.dumps([uniq[i % 200] for i in range(chunk)])[1:-1]
text = ('[' + ','.join([body] * (n // chunk)) + ']').encode()
warm = ('[' + ','.join([body] * 2) + ']').encode()
eng = pipeline.Engine(par.parameters, calib, max_frames=chunk, max_persons_per_camera=4)
eng.load_gat(syn.gat_state_dict(7, 902, logit_gain=25.0, logit_shift=0.948), syn.gat_params(902))
eng.load_mlp(syn.mlp_state_dict(11, 1260))
extra = [torch.cuda.Stream() for _ in range(int(os.environ.get('MPE_PROBE_EXTRA_STREAMS', '0')))]      # what bench.py has alive
for s_ in extra:
    with torch.cuda.stream(s_):
        torch.zeros(16, device='cuda').add_(1)
if os.environ.get('MPE_PROBE_RESIDENT'):                 # a resident batch through the engine first, as bench.py does
    fr = [{c: [f[c][0], f[c][1]] for c in f} for f in uniq]
    db = eng.to_device(eng.pack([fr[i % 200] for i in range(chunk)]))
    for _ in range(30):
        _, pp, nn_ = eng.match(db, want_scores=False); eng.mlp3d(db, pp, nn_)
    torch.cuda.synchronize()
sum(len(nn) for _, _, nn in eng.stream_json(warm, chunk_frames=chunk, parser=parser, **kw))
torch.cuda.synchronize()
for rep in range(int(sys.argv[4]) if len(sys.argv) > 4 else 1):
  t0 = time.perf_counter()
  got, stamps = 0, []
  for _, _, nn in eng.stream_json(text, chunk_frames=chunk, parser=parser, **kw):
      got += len(nn)
      stamps.append(time.perf_counter() - t0)
  t_last = time.perf_counter() - t0
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  if os.environ.get('MPE_JSON_TIMING'):
      gaps = [b - a for a, b in zip(stamps, stamps[1:])]
      print('first window out after %.2f ms; between windows: median %.2f ms, slowest %.2f ms; after the last window %.2f ms'
            % (1e3 * stamps[0], 1e3 * sorted(gaps)[len(gaps) // 2], 1e3 * max(gaps), 1e3 * (dt - stamps[-1])))
      print('gaps, ms: ' + ' '.join('%.1f' % (1e3 * g) for g in gaps))
  print('%s parser: %d frames, %.1f frames/s, %.2f ms per %d-frame window, %.2f GB/s of JSON' % (parser, got, got / dt, 1e3 * dt / (n // chunk), chunk, len(text) / dt / 1e9))
