"""GPU: the evaluation harness end to end on synthetic frames."""
import importlib

import pytest

pytestmark = pytest.mark.gpu


def test_triangulation_harness_recovers_ground_truth():
    """With the ground-truth pairing as scores, undistort + DLT must recover the synthetic 3D
    joints: MPJPE far below a millimetre on noise-free projections (the lens model of the
    generator and the 5-iteration undistortion differ by ~1e-3 px)."""
    m = importlib.import_module('3d_multi_pose_estimator_amd.harness.metrics_from_triangulation')
    out = m.main(['--synthetic', '40', '--random-weights', '--teacher-scores', '--batch', '16'])
    assert out['mpjpe_mm'] < 0.5
    assert out[25][2] > 0.95          # recall at 25 mm


def test_model_harness_runs():
    m = importlib.import_module('3d_multi_pose_estimator_amd.harness.metrics_from_model')
    out = m.main(['--synthetic', '24', '--random-weights', '--batch', '16'])
    assert 25 in out


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
