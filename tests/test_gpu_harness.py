"""GPU: the evaluation harness end to end -- on synthetic frames, and on the committed test file
against the numbers the REFERENCE'S OWN scripts printed for it (tests/golden/harness/)."""
import importlib
import json
import os

import pytest

from conftest import GOLDEN, harness_model_files

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('script,key', [('metrics_from_model', 'model'), ('metrics_from_triangulation', 'triangulation')])
def test_harness_reproduces_the_reference_scripts_report(script, key, tmp_path):
    """the build's harness, driven exactly like the reference script
    (--testfiles --tmdir --modelsdir --datastep; weights read from skeleton_matching.prms/.tch and
    pose_estimator.pytorch, calibration from tm_<a>_<b>.pickle), prints what
    /root/reference/test/<script>.py printed for the same files (oracle/gen_harness_golden.py):
    AP / precision / recall at every threshold exactly, MPJPE within 0.01 mm."""
    hd = os.path.join(GOLDEN, 'harness')
    with open(os.path.join(hd, 'harness_expected.json')) as fh:
        exp = json.load(fh)
    mdir = harness_model_files(str(tmp_path), exp['inputs'])
    m = importlib.import_module('3d_multi_pose_estimator_amd.harness.' + script)
    out = m.main(['--testfiles', os.path.join(hd, exp['inputs']['testfile']), '--tmdir', hd, '--modelsdir', mdir,
                  '--datastep', str(exp['inputs']['datastep']), '--batch', '7'])
    want = exp[key]
    assert abs(out['mpjpe_mm'] - want['mpjpe_mm']) < 0.01, (out['mpjpe_mm'], want['mpjpe_mm'])
    for th, triple in want['ap'].items():
        assert out['ap'][th] == pytest.approx(triple, rel=1e-12, abs=1e-12), (th, out['ap'][th], triple)


@pytest.mark.parametrize('extra', [['--gat-acc64'], ['--mlp-precision', 'f64'], ['--gat-acc64', '--mlp-precision', 'max_accuracy']])
def test_harness_precision_flags_keep_the_reference_scripts_report(extra, tmp_path):
    """The harness's own precision flags (f64 sums in every GAT GEMM; the MLP in mode 4 / 5) on the committed test file: the report of
    /root/reference/test/metrics_from_model.py for the same files is reproduced under each -- AP / precision / recall exactly, MPJPE within
    0.01 mm (a14 / a15 with the more accurate arithmetic)."""
    hd = os.path.join(GOLDEN, 'harness')
    with open(os.path.join(hd, 'harness_expected.json')) as fh:
        exp = json.load(fh)
    mdir = harness_model_files(str(tmp_path), exp['inputs'])
    m = importlib.import_module('3d_multi_pose_estimator_amd.harness.metrics_from_model')
    out = m.main(['--testfiles', os.path.join(hd, exp['inputs']['testfile']), '--tmdir', hd, '--modelsdir', mdir,
                  '--datastep', str(exp['inputs']['datastep']), '--batch', '7'] + extra)
    want = exp['model']
    assert abs(out['mpjpe_mm'] - want['mpjpe_mm']) < 0.01, (out['mpjpe_mm'], want['mpjpe_mm'])
    for th, triple in want['ap'].items():
        assert out['ap'][th] == pytest.approx(triple, rel=1e-12, abs=1e-12), (th, out['ap'][th], triple)


def test_triangulation_harness_recovers_ground_truth():
    """With the ground-truth pairing as scores, undistort + DLT must recover the synthetic 3D
    joints: MPJPE far below a millimetre on noise-free projections (the lens model of the
    generator and the 5-iteration undistortion differ by ~1e-3 px)."""
    m = importlib.import_module('3d_multi_pose_estimator_amd.harness.metrics_from_triangulation')
    out = m.main(['--synthetic', '40', '--random-weights', '--teacher-scores', '--batch', '16'])
    assert out['mpjpe_mm'] < 0.5
    assert out['ap']['25'][2] > 0.95          # recall at 25 mm


def test_model_harness_runs():
    m = importlib.import_module('3d_multi_pose_estimator_amd.harness.metrics_from_model')
    out = m.main(['--synthetic', '24', '--random-weights', '--batch', '16'])
    assert '25' in out['ap']


def test_sm_metrics_harness():
    """Clustering-quality harness (reference test/sm_metrics.py): the ground-truth pairing as
    scores must give a perfect grouping; the metric code is exercised end to end."""
    m = importlib.import_module('3d_multi_pose_estimator_amd.harness.sm_metrics')
    out = m.main(['--synthetic', '32', '--random-weights', '--teacher-scores', '--batch', '16'])
    assert out['rand score'] > 0.999 and out['v_measure'] > 0.999
    out2 = m.main(['--synthetic', '16', '--random-weights', '--batch', '16'])
    assert 0.0 <= out2['homogeneity'] <= 1.0


def test_reprojection_error_harness():
    """reference test/reprojection_error.py: triangulated joints of correctly grouped, noise-free
    detections reproject onto the detections (sub-pixel; the radial-only model of the metric
    ignores the small tangential terms of the generator's lens model)."""
    m = importlib.import_module('3d_multi_pose_estimator_amd.harness.reprojection_error')
    out = m.main(['--synthetic', '16', '--random-weights', '--teacher-scores', '--batch', '16'])
    tri = [v for (kind, cam), v in out.items() if kind == 'triang']
    assert len(tri) == 5 and max(t[1] for t in tri) < 2.0
    assert any(kind == 'est' for kind, _ in out)


def _expected():
    hd = os.path.join(GOLDEN, 'harness')
    with open(os.path.join(hd, 'harness_expected.json')) as fh:
        return hd, json.load(fh)


def test_sm_metrics_harness_reproduces_the_reference_script(tmp_path):
    """f2: harness/sm_metrics.py on the engine against the four numbers
    /root/reference/test/sm_metrics.py printed for the same files."""
    hd, exp = _expected()
    mdir = harness_model_files(str(tmp_path), exp['inputs'])
    m = importlib.import_module('3d_multi_pose_estimator_amd.harness.sm_metrics')
    out = m.main(['--testfiles', os.path.join(hd, exp['inputs']['testfile']), '--tmdir', hd, '--modelsdir', mdir,
                  '--datastep', str(exp['inputs']['datastep']), '--batch', '7'])
    for k, v in exp['sm_metrics'].items():
        assert out[k] == pytest.approx(v, rel=1e-12, abs=1e-12), k


def test_reprojection_harness_reproduces_the_reference_script(tmp_path):
    """f3: harness/reprojection_error.py on the engine: per-camera medians of the reprojection
    error of the MLP estimate and of the triangulation as /root/reference/test/reprojection_error.py
    printed them (the means are dominated by a few points projected through z ~ 0: log scale)."""
    import numpy as np
    hd, exp = _expected()
    mdir = harness_model_files(str(tmp_path), exp['inputs'])
    m = importlib.import_module('3d_multi_pose_estimator_amd.harness.reprojection_error')
    out = m.main(['--testfiles', os.path.join(hd, exp['inputs']['testfile']), '--tmdir', hd, '--modelsdir', mdir,
                  '--datastep', str(exp['inputs']['datastep']), '--batch', '7'])
    for cam, kinds in exp['reprojection_error'].items():
        for kind, (mean, median) in kinds.items():
            g = out[(kind, cam)]
            assert g[1] == pytest.approx(median, rel=2e-4), (cam, kind, g, median)
            assert abs(np.log10(g[0]) - np.log10(mean)) < 0.5, (cam, kind, g, mean)


def test_sm_metrics_without_gt_harness_identity_clash(tmp_path, monkeypatch):
    """f2, second script (pinned to the reference's printed numbers in tests/test_gpu_generated.py): a property on top --
    scenes composed from files whose individuals carry DIFFERENT identity codes are grouped well; with one individual's
    file given twice the two copies share a code and the score drops."""
    syn = importlib.import_module('3d_multi_pose_estimator_amd.synthetic')
    cal = importlib.import_module('3d_multi_pose_estimator_amd.calibration')
    par = importlib.import_module('3d_multi_pose_estimator_amd.parameters').parameters
    calib = cal.Calibration(par)
    hd, exp = _expected()
    mdir = harness_model_files(str(tmp_path), exp['inputs'])
    monkeypatch.chdir(tmp_path)
    files = []
    for k in range(3):                                  # three individuals, person index k -> prob level k
        frames = []
        for i in range(12):
            full, _ = syn.make_frame(calib, 9000 + i, syn.FrameSpec(persons=3, noise_px=1.0, identity_prob=True, permute=False))
            one = {}
            for cam in full:
                sk = json.loads(full[cam][0])
                one[cam] = [json.dumps([sk[k]]), full[cam][1]]
            frames.append(one)
        path = tmp_path / ('person%d.json' % k)
        path.write_text(json.dumps(frames))
        files.append(str(path))
    m = importlib.import_module('3d_multi_pose_estimator_amd.harness.sm_metrics_without_gt')
    out = m.main(['--testfiles'] + files + ['--modelsdir', mdir, '--datastep', '2', '--batch', '4', '--seed', '11'])
    assert out['n_data'] >= 6
    assert out['rand score'] > 0.85 and out['homogeneity'] > 0.95
    out2 = m.main(['--testfiles', files[0], files[0], files[1], '--modelsdir', mdir, '--datastep', '2', '--batch', '4', '--seed', '11'])
    assert out2['rand score'] < out['rand score']       # two "individuals" with the same identity code get mixed
