"""MI355X-native inference path of gnns4hri/3D_multi_pose_estimator.

The directory name starts with a digit, so import it with
``importlib.import_module('3d_multi_pose_estimator_amd')`` or through the root-level
alias module ``mpe_amd``.
"""
__version__ = '0.1.0'
