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


@pytest.fixture(scope='session')
def meta():
    with open(os.path.join(GOLDEN, 'meta.json')) as fh:
        return json.load(fh)


@pytest.fixture(scope='session')
def calib():
    return pkg('calibration').Calibration(pkg('parameters').parameters)


@pytest.fixture(scope='session')
def gat_weights(meta):
    syn = pkg('synthetic')
    sd = syn.gat_state_dict(meta['gat_seed'], meta['num_feats'], logit_gain=meta['logit_gain'],
                            logit_shift=meta['logit_shift'])
    return sd, syn.gat_params(meta['num_feats'])


@pytest.fixture(scope='session')
def mlp_weights(meta):
    return pkg('synthetic').mlp_state_dict(meta['mlp_seed'], meta['mlp_in'])


def load_case(name):
    arr = np.load(os.path.join(GOLDEN, name + '.npz'))
    with open(os.path.join(GOLDEN, name + '.frames.json')) as fh:
        frames = json.load(fh)
    return arr, frames


CASES = ['c1_2view_1person', 'c2_5x4_clean', 'c2_5x4_messy', 'c2_5x4_reordered', 'c2_3x2', 'c4_5x10']
