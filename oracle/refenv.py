"""TEST INFRASTRUCTURE (build container only): import the reference's hot-path modules.

Puts the third-party stand-ins of oracle/shims first on sys.path, changes into
/root/reference/test (the reference resolves its calibration pickle relative to the
cwd, parameters.py:73) and imports the modules test/metrics_from_model.py imports
(:12-24).  /root/reference is read-only, so byte-code writing is disabled.  Used by
oracle/gen_golden.py and by nothing that runs on the GPU box.
"""
import importlib
import os
import sys

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))


def available():
    return os.path.isdir(os.path.join(REF, 'skeleton_matching'))


def load(parameters_dir=None):
    """Return a dict of the reference modules.  `parameters_dir`, if given, is a
    directory holding an alternative ``parameters.py`` (same schema) that shadows the
    reference's own, the way a user edits CONFIGURATION there."""
    if not available():
        raise RuntimeError('reference tree not present')
    sys.dont_write_bytecode = True
    os.chdir(os.path.join(REF, 'test'))
    paths = [os.path.join(HERE, 'shims')]
    if parameters_dir:
        paths.append(parameters_dir)
    paths += [os.path.join(REF, 'skeleton_matching'), os.path.join(REF, 'utils'), REF]
    for p in reversed(paths):
        if p in sys.path:
            sys.path.remove(p)
        sys.path.insert(0, p)
    import torch
    torch.set_grad_enabled(False)
    names = ['parameters', 'gat2', 'graph_generator', 'pose_estimator_dataset_from_json', 'mlp',
             'skeleton_matching_utils', 'pose_estimator_utils']
    import contextlib
    import io
    mods = {}
    with contextlib.redirect_stdout(io.StringIO()):
        for n in names:
            mods[n] = importlib.import_module(n)
    return mods
