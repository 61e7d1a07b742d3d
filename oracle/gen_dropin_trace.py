"""TEST INFRASTRUCTURE (build container only): the CALL SEQUENCE of the reference's per-frame loop, as a fixture.

Runs /root/reference/test/metrics_from_model.py UNCHANGED (as `__main__`, the layout and stand-ins of gen_harness_golden.py) on
the committed pinning file tests/golden/harness/syn_pinning_test.json with tracing wrappers installed around the symbols the
script imports -- MergedMultipleHumansDataset, GAT2.forward, get_person_proposal_from_network_output, PoseEstimatorDataset,
PoseEstimatorMLP.forward -- and stores, per processed frame, the sequence of calls with a shape-level description of every
argument (tensor shapes, dict keys or sizes, list lengths, scalars) in tests/golden/dropin/call_trace.json.

The package's one-frame-per-call loop (3d_multi_pose_estimator_amd/harness/dropin_loop.py) is written in its own form; that it
calls the mirrors in the reference's ORDER with arguments of the reference's SHAPES is what this fixture pins
(tests/test_gpu_dropin.py::test_frame_loop_calls_the_mirrors_like_the_reference_script).  No text of the reference is stored.

    python oracle/gen_dropin_trace.py
"""
import importlib
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import gen_harness_golden as H      # noqa: E402

OUT = os.environ.get('MPE_DROPIN_TRACE_OUT') or os.path.join(ROOT, 'tests', 'golden', 'dropin')      # (the regeneration test writes elsewhere)

PRELUDE = r'''
import sys, runpy, json
sys.dont_write_bytecode = True
sys.path[:0] = %(paths)r
import numpy as np, torch
import graph_generator, gat2, skeleton_matching_utils, pose_estimator_dataset_from_json, mlp

TRACE = []

def describe(value):
    if isinstance(value, torch.Tensor):
        return ['tensor', list(value.shape)]
    if isinstance(value, np.ndarray):
        return ['array', list(value.shape)]
    if isinstance(value, dict):
        return ['dict', sorted(value)] if 0 < len(value) <= 8 and all(isinstance(k, str) for k in value) else ['dict', len(value)]
    if isinstance(value, (list, tuple)):
        return ['list', len(value)]
    if isinstance(value, (bool, int, float, str)) or value is None:
        return value
    if hasattr(value, 'number_of_nodes'):
        return ['graph', int(value.number_of_nodes())]
    return type(value).__name__

def record(name, args, kwargs):
    TRACE.append([name, [describe(a) for a in args], {k: describe(v) for k, v in sorted(kwargs.items())}])

def traced_factory(module, attr, name):
    original = getattr(module, attr)
    def make(*args, **kwargs):
        record(name, args, kwargs)
        return original(*args, **kwargs)
    setattr(module, attr, make)

def traced_forward(cls, name):
    original = cls.forward
    def forward(self, *args, **kwargs):
        record(name, args, kwargs)
        return original(self, *args, **kwargs)
    cls.forward = forward

def traced_init(cls, name):                       # (the classes name themselves in super(): they stay classes)
    original = cls.__init__
    def __init__(self, *args, **kwargs):
        record(name, args, kwargs)
        original(self, *args, **kwargs)
    cls.__init__ = __init__

traced_init(graph_generator.MergedMultipleHumansDataset, 'MergedMultipleHumansDataset')
traced_factory(skeleton_matching_utils, 'get_person_proposal_from_network_output', 'get_person_proposal_from_network_output')
traced_init(pose_estimator_dataset_from_json.PoseEstimatorDataset, 'PoseEstimatorDataset')
traced_forward(gat2.GAT2, 'GAT2.__call__')
traced_forward(mlp.PoseEstimatorMLP, 'PoseEstimatorMLP.__call__')
sys.argv = %(argv)r
try:
    runpy.run_path(%(script)r, run_name='__main__')
finally:
    with open(%(out)r, 'w') as fh:
        json.dump(TRACE, fh)
'''


def main():
    syn = importlib.import_module(H.PKG + '.synthetic')
    par = importlib.import_module(H.PKG + '.parameters')
    params = par.parameters
    V, J = len(params.camera_names), len(params.joint_list)
    data_file = os.path.join(H.OUT, H.TEST_NAME)
    os.makedirs(OUT, exist_ok=True)
    with tempfile.TemporaryDirectory() as layout:
        mdir = os.path.join(layout, 'models_panoptic')
        os.makedirs(mdir)
        H.save_models(mdir, syn, V, J)
        os.makedirs(os.path.join(layout, 'test'), exist_ok=True)
        for name in ('tm_panoptic.pickle', 'human_pose.json'):
            os.symlink(os.path.join(H.REF, name), os.path.join(layout, name))
        raw = os.path.join(layout, 'trace.json')
        code = PRELUDE % {'paths': [H.SHIMS, os.path.join(H.REF, 'skeleton_matching'), os.path.join(H.REF, 'utils'), H.REF],
                          'argv': ['metrics_from_model.py', '--testfiles', data_file, '--tmdir', H.OUT, '--modelsdir', mdir,
                                   '--datastep', str(H.DATASTEP)],
                          'script': os.path.join(H.REF, 'test', 'metrics_from_model.py'), 'out': raw}
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1', HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='')
        res = subprocess.run([sys.executable, '-c', code], cwd=os.path.join(layout, 'test'), env=env, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError('metrics_from_model.py failed:\n' + res.stderr[-3000:])
        flat = json.load(open(raw))
    frames = []
    for event in flat:                                  # a frame's chain starts with the graph construction
        if event[0] == 'MergedMultipleHumansDataset':
            frames.append([])
        frames[-1].append(event)
    doc = {'script': 'test/metrics_from_model.py (unchanged, run as __main__)', 'testfile': H.TEST_NAME, 'datastep': H.DATASTEP,
           'note': 'per processed frame (frames without ground-truth bodies are skipped by the script before any call): '
                   '[symbol, positional argument descriptions, keyword argument descriptions]',
           'frames': frames}
    with open(os.path.join(OUT, 'call_trace.json'), 'w') as fh:
        json.dump(doc, fh, indent=0)
    print('%d frames, %d calls' % (len(frames), len(flat)))
    for ev in frames[0]:
        print(ev)


if __name__ == '__main__':
    main()
