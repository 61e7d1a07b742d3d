import importlib
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
PKG = '3d_multi_pose_estimator_amd'


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pkg(sub=None):
    return importlib.import_module(PKG + ('.' + sub if sub else ''))


def oracle():
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    return importlib.import_module('oracle_np')


VARIANT_CASES = {
    'panoptic': ['c1_2view_1person', 'c2_5x4_clean', 'c2_5x4_messy', 'c2_5x4_reordered', 'c2_3x2', 'c4_5x10'],
    'arplab': ['arp_6x3', 'arp_robot_pair'],
    # the reference's camera-subset preset (parameters.py:110-112): six cameras configured, the two robot cameras used
    'arprobot': ['arp_robot_only', 'arp_robot_only_all_streams'],
    'ring23': ['ring23x3', 'ring23_sparse'],
}
CASES = VARIANT_CASES['panoptic']
ALL_CASES = [(v, c) for v, cs in VARIANT_CASES.items() for c in cs]
# round 6: 40 (PANOPTIC), 30 (ARPLAB) and 8 (23 cameras) frames of random shape through the reference (oracle/gen_golden.py: random_shape_specs) -- one case, used by the stage tests
# that take any case; not part of CASES (tests that walk the hand-made cases keep their size)
FUZZ_CASES = [('panoptic', 'fz_random_shapes'), ('arplab', 'fz_random_shapes'), ('ring23', 'fz_random_shapes')]
ALL_CASES_FZ = ALL_CASES + FUZZ_CASES


def golden_dir(variant='panoptic'):
    return GOLDEN if variant == 'panoptic' else os.path.join(GOLDEN, variant)


class Env:
    """Parameters, calibration and deterministic weights of one fixture variant."""

    def __init__(self, variant):
        self.variant = variant
        with open(os.path.join(golden_dir(variant), 'meta.json')) as fh:
            self.meta = json.load(fh)
        par, cal, syn = pkg('parameters'), pkg('calibration'), pkg('synthetic')
        if variant == 'panoptic':
            self.params = par.parameters
            self.calib = cal.Calibration(self.params)
        elif variant in ('arplab', 'arprobot'):
            self.params = par.select('ARPLAB' if variant == 'arplab' else 'ARPLAB_ROBOT')
            self.calib = cal.Calibration(self.params)          # package copy of tm_arp
        else:
            self.params = par.select('RING23')
            self.calib = cal.Calibration(self.params, syn.ring_transform_manager(self.params))
        self._gat = self._mlp = self._mlp_room = None

    @property
    def gat(self):
        if self._gat is None:
            syn = pkg('synthetic')
            m = self.meta
            self._gat = (syn.gat_state_dict(m['gat_seed'], m['num_feats'], logit_gain=m['logit_gain'],
                                            logit_shift=m['logit_shift']), syn.gat_params(m['num_feats']))
        return self._gat

    @property
    def mlp(self):
        if self._mlp is None:
            self._mlp = pkg('synthetic').mlp_state_dict(self.meta['mlp_seed'], self.meta['mlp_in'])
        return self._mlp


    @property
    def mlp_room(self):
        """The capture-volume MLP of the fixtures (`mlp_out_room` / `poses_room`): outputs within a few metres."""
        if self._mlp_room is None:
            m = self.meta['room_mlp']
            p = self.params
            self._mlp_room = pkg('synthetic').decoder_mlp_state_dict(
                len(p.used_cameras), len(p.joint_list), p.numbers_per_joint, noise_seed=m['noise_seed'], noise_bound=m['noise_bound'])
        return self._mlp_room


_envs = {}


def env(variant='panoptic'):
    if variant not in _envs:
        _envs[variant] = Env(variant)
    return _envs[variant]


@pytest.fixture(scope='session')
def meta():
    return env().meta


@pytest.fixture(scope='session')
def calib():
    return env().calib


@pytest.fixture(scope='session')
def gat_weights():
    return env().gat


@pytest.fixture(scope='session')
def mlp_weights():
    return env().mlp


def load_case(name, variant='panoptic'):
    d = golden_dir(variant)
    arr = np.load(os.path.join(d, name + '.npz'))
    with open(os.path.join(d, name + '.frames.json')) as fh:
        frames = json.load(fh)
    return arr, frames


def harness_model_files(out_dir, inputs):
    """The model files of the harness pinning test in the reference's three formats
    (skeleton_matching.prms / .tch, pose_estimator.pytorch), rebuilt from the deterministic
    generators named in harness_expected.json (the files themselves are 116 MB)."""
    import pickle

    import torch
    syn = pkg('synthetic')
    par = pkg('parameters').parameters
    V, J = len(par.camera_names), len(par.joint_list)
    nf = 2 + V * J * 10
    mdir = os.path.join(out_dir, 'models')
    os.makedirs(mdir, exist_ok=True)
    prm = dict(syn.gat_params(nf), nonlinearity=torch.nn.LeakyReLU(), final_activation=torch.nn.Sigmoid())
    with open(os.path.join(mdir, 'skeleton_matching.prms'), 'wb') as fh:
        pickle.dump(prm, fh)
    g, m = inputs['gat'], inputs['mlp']
    assert g['kind'] == 'matcher' and m['kind'] == 'decoder'
    gat = syn.matcher_gat_state_dict(nf, V, J, noise_seed=g['noise_seed'], noise_bound=g['noise_bound'])
    torch.save({k: torch.from_numpy(v) for k, v in gat.items()}, os.path.join(mdir, 'skeleton_matching.tch'))
    mlp = syn.decoder_mlp_state_dict(V, J, par.numbers_per_joint, noise_seed=m['noise_seed'], noise_bound=m['noise_bound'])
    torch.save({'model_state_dict': {k: torch.from_numpy(v) for k, v in mlp.items()}},
               os.path.join(mdir, 'pose_estimator.pytorch'))
    return mdir


def generated_fixture():
    """tests/golden/generated/: what /root/reference/test/sm_metrics_without_gt.py built and printed on the four committed
    single-person files with `random` seeded (oracle/gen_generated_golden.py) -> (expected dict, arrays, file paths,
    probabilities_set of the script, :99-104)."""
    d = os.path.join(GOLDEN, 'generated')
    with open(os.path.join(d, 'generated_expected.json')) as fh:
        exp = json.load(fh)
    arr = np.load(os.path.join(d, 'generated_graphs.npz'))
    files = [os.path.join(d, f) for f in exp['files']]
    lengths = []
    for f in files:
        with open(f) as fh:
            lengths.append(len(json.load(fh)))
    probs = [0.8] + [0.8 * n / lengths[0] for n in lengths[1:]]
    return exp, arr, files, probs


def generated_gat_weights(exp):
    syn = pkg('synthetic')
    par = pkg('parameters').parameters
    V, J = len(par.used_cameras_skeleton_matching), len(par.joint_list)
    nf = 2 + V * J * 10
    return syn.matcher_gat_state_dict(nf, V, J, noise_seed=exp['gat']['noise_seed'], noise_bound=exp['gat']['noise_bound']), syn.gat_params(nf)


def proposals_as_rows(proposals, cams):
    return [[-1 if d[c] is None else int(d[c]) for c in cams] for d in proposals]
