"""TEST INFRASTRUCTURE (build container only): pin the evaluation harness
with numbers produced by the REFERENCE'S OWN SCRIPTS.

Runs /root/reference/test/metrics_from_model.py and metrics_from_triangulation.py UNCHANGED (as
`__main__`, cwd = /root/reference/test, third-party stand-ins of oracle/shims first on sys.path)
on

  * a synthetic Panoptic-format test file with ground truth (`syn_pinning_test.json`; the name
    gives the calibration file `tm_syn_pinning.pickle` the scripts look for, :109-111),
  * a dataset calibration that differs from the rig's by a rigid offset (so the GT path
    dataset-camera-1 -> world, :152-161, does real work),
  * deterministic weights saved in the reference's three file formats: skeleton_matching.prms
    (pickled dict incl. the activation modules, train_skeleton_matching.py:231-246),
    skeleton_matching.tch (state dict), pose_estimator.pytorch ({'model_state_dict': ...},
    train_pose_estimator.py:269-277),

and stores what they PRINT (AP / precision / recall per threshold, MEAN ERR) in
tests/golden/harness/harness_expected.json next to the two input files.  The weight files are
not committed (116 MB); tests/test_gpu_harness.py rebuilds them from the same generators
(3d_multi_pose_estimator_amd/synthetic.py) and runs the build's harness on the same inputs.

    python oracle/gen_harness_golden.py
"""
import importlib
import json
import os
import pickle
import re
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = '/root/reference'
SHIMS = os.path.join(ROOT, 'oracle', 'shims')
OUT = os.path.join(ROOT, 'tests', 'golden', 'harness')
PKG = '3d_multi_pose_estimator_amd'

GAT_NOISE_SEED, GAT_NOISE = 5, 1e-4
MLP_NOISE_SEED, MLP_NOISE = 3, 2e-4
DATASTEP = 3
TEST_NAME = 'syn_pinning_test.json'


def rigid_offset():
    """The dataset's root frame = the rig's root frame moved by this transform."""
    a, b = 0.3, -0.2
    Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry
    T[:3, 3] = [0.4, -0.1, 0.25]
    return T


def build_frames(calib, syn):
    """48 frames; with --datastep 3 the scripts read frames 0, 3, 6, ...: those carry the cases,
    the frames between them are different fillers (a wrong stride changes every number)."""
    def F(**kw):
        return syn.FrameSpec(identity_prob=True, **kw)
    cases = [F(persons=4, noise_px=1.0), F(persons=1, noise_px=2.0, joint_drop=0.1), F(persons=2, noise_px=1.0),
             F(persons=3, noise_px=0.5), F(persons=3, noise_px=1.0, spurious=1), F(persons=5, noise_px=1.5),
             F(persons=2, cameras=['trackerb', 'trackerd', 'trackere'], noise_px=1.0),
             F(persons=4, noise_px=0.5, joint_drop=0.2), F(persons=1, noise_px=3.0),
             F(persons=2, cameras=['trackerc']),                      # one camera: no graph -> skipped (:195-196)
             F(persons=4, noise_px=6.0), F(persons=2, noise_px=1.0),
             F(persons=4, noise_px=1.0), F(persons=3, noise_px=12.0), F(persons=2, noise_px=1.0, empty_cameras=('trackera',)),
             F(persons=6, noise_px=1.0)]
    R = rigid_offset()
    Rinv = np.linalg.inv(R)
    frames = []
    for i in range(48):
        spec = cases[i // DATASTEP] if i % DATASTEP == 0 else F(persons=2 + i % 2, noise_px=40.0)
        frame, _ = syn.make_frame(calib, 7000 + i, spec)
        for cam in frame:
            bodies = frame[cam][3]
            for bi, body in enumerate(bodies):
                # GT in the dataset's coordinates (cm): the scripts bring it back with
                # T_i(cam1) . T_dataset(root->cam1)
                for k in list(body.keys()):
                    w = np.array(body[k]) / 100.0
                    d = Rinv[:3, :3] @ w + Rinv[:3, 3]
                    body[k] = [float(c) * 100.0 for c in d]
                # every third body of frame 6 and 24 has no '-1' key: invalid GT (:170-173)
                if i in (6, 24) and bi % 3 == 0 and '-1' in body:
                    body['2'] = body.pop('-1')
        if i == 33:
            for cam in frame:                                          # no GT bodies: skipped (:137-138)
                frame[cam][3] = []
        frames.append(frame)
    return frames


def save_models(mdir, syn, V, J):
    import torch
    nf = 2 + V * J * 10
    prm = syn.gat_params(nf)
    prm = dict(prm, nonlinearity=torch.nn.LeakyReLU(), final_activation=torch.nn.Sigmoid())
    with open(os.path.join(mdir, 'skeleton_matching.prms'), 'wb') as fh:
        pickle.dump(prm, fh)
    gat = syn.matcher_gat_state_dict(nf, V, J, noise_seed=GAT_NOISE_SEED, noise_bound=GAT_NOISE)
    torch.save({k: torch.from_numpy(v) for k, v in gat.items()}, os.path.join(mdir, 'skeleton_matching.tch'))
    mlp = syn.decoder_mlp_state_dict(V, J, 14, noise_seed=MLP_NOISE_SEED, noise_bound=MLP_NOISE)
    torch.save({'model_state_dict': {k: torch.from_numpy(v) for k, v in mlp.items()}},
               os.path.join(mdir, 'pose_estimator.pytorch'))


def parse_sm(text):
    out = {}
    for line in text.splitlines():
        m = re.match(r'(rand score|homogeneity|completeness|v_measure) (\S+)', line)
        if m:
            out[m.group(1)] = float(m.group(2))
    return out


def parse_reprojection(text):
    out, cam = {}, None
    for line in text.splitlines():
        m = re.match(r'-+ CAMERA (\S+) -+', line)
        if m:
            cam = m.group(1)
            out[cam] = {}
        m = re.match(r'(est|triang|GT) (\S+) (\S+)', line)
        if m and cam:
            out[cam][m.group(1)] = [float(m.group(2)), float(m.group(3))]
    return out


def run_reference(script, data_file, tm_dir, models_dir):
    """The reference script as __main__, unchanged.  The working directory is a scratch copy of the
    reference's LAYOUT, not of its files: <layout>/test is the cwd, <layout>/tm_panoptic.pickle a
    symlink to the reference's calibration (parameters.transformations_path is cwd-relative), <layout>/human_pose.json likewise, and
    <layout>/models_panoptic the model directory that test/sm_metrics.py hard-codes (:81-84); the
    reference's own source directories are put on sys.path by absolute name."""
    layout = os.path.dirname(models_dir.rstrip('/'))
    os.makedirs(os.path.join(layout, 'test'), exist_ok=True)
    for name in ('tm_panoptic.pickle', 'human_pose.json'):        # data files the scripts open relative to the cwd
        link = os.path.join(layout, name)
        if not os.path.exists(link):
            os.symlink(os.path.join(REF, name), link)
    paths = [SHIMS, os.path.join(REF, 'skeleton_matching'), os.path.join(REF, 'utils'), REF]
    code = ('import sys, runpy; sys.dont_write_bytecode = True; sys.path[:0] = %r; '
            'sys.argv = [%r, "--testfiles", %r, "--tmdir", %r, "--modelsdir", %r, "--datastep", %r]; '
            'runpy.run_path(%r, run_name="__main__")'
            % (paths, script, data_file, tm_dir, models_dir, str(DATASTEP), os.path.join(REF, 'test', script)))
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1', HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='')
    res = subprocess.run([sys.executable, '-c', code], cwd=os.path.join(layout, 'test'), env=env, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError('%s failed:\n%s' % (script, res.stderr[-3000:]))
    return res.stdout


def parse_report(text):
    out = {'ap': {}}
    for line in text.splitlines():
        m = re.match(r'AP, precise and recall for (\d+) : (\S+) (\S+) (\S+)', line)
        if m:
            out['ap'][m.group(1)] = [float(m.group(2)), float(m.group(3)), float(m.group(4))]
        m = re.match(r'MEAN ERR \(mm\) (\S+)', line)
        if m:
            out['mpjpe_mm'] = float(m.group(1))
    return out


def main():
    syn = importlib.import_module(PKG + '.synthetic')
    cal = importlib.import_module(PKG + '.calibration')
    par = importlib.import_module(PKG + '.parameters')
    sys.path.insert(0, SHIMS)
    from pytransform3d.transform_manager import TransformManager
    params = par.parameters
    calib = cal.Calibration(params)
    os.makedirs(OUT, exist_ok=True)
    frames = build_frames(calib, syn)
    data_file = os.path.join(OUT, TEST_NAME)
    with open(data_file, 'w') as fh:
        json.dump(frames, fh)
    # dataset calibration: root_dataset -> camera = (root_rig -> camera) . R
    R = rigid_offset()
    tm = TransformManager()
    for i, cam in enumerate(params.camera_names):
        tm.add_transform('root', cam, calib.T_d[i] @ R)
    with open(os.path.join(OUT, 'tm_syn_pinning.pickle'), 'wb') as fh:
        pickle.dump(tm, fh)
    V, J = len(params.camera_names), len(params.joint_list)
    with tempfile.TemporaryDirectory() as layout:
        mdir = os.path.join(layout, 'models_panoptic')
        os.makedirs(mdir)
        save_models(mdir, syn, V, J)
        report = {}
        for key, script in (('model', 'metrics_from_model.py'), ('triangulation', 'metrics_from_triangulation.py')):
            text = run_reference(script, data_file, OUT, mdir)
            print(text)
            report[key] = parse_report(text)
            assert 'mpjpe_mm' in report[key] and len(report[key]['ap']) == 6, text
        text = run_reference('sm_metrics.py', data_file, OUT, mdir)
        print(text)
        report['sm_metrics'] = parse_sm(text)
        assert len(report['sm_metrics']) == 4, text
        text = run_reference('reprojection_error.py', data_file, OUT, mdir)
        print(text)
        report['reprojection_error'] = parse_reprojection(text)
        assert len(report['reprojection_error']) == V, text
    report['inputs'] = {'testfile': TEST_NAME, 'tm': 'tm_syn_pinning.pickle', 'datastep': DATASTEP,
                        'gat': {'kind': 'matcher', 'noise_seed': GAT_NOISE_SEED, 'noise_bound': GAT_NOISE},
                        'mlp': {'kind': 'decoder', 'noise_seed': MLP_NOISE_SEED, 'noise_bound': MLP_NOISE}}
    with open(os.path.join(OUT, 'harness_expected.json'), 'w') as fh:
        json.dump(report, fh, indent=1)
    print(json.dumps(report, indent=1))


if __name__ == '__main__':
    main()
