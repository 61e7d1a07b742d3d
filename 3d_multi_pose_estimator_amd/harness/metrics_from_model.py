"""Counterpart of the reference's test/metrics_from_model.py on the MI355X path
(GAT matching -> clustering -> MLP 3D).  Usage mirrors the reference:

    python -m 3d_multi_pose_estimator_amd.harness.metrics_from_model --testfiles F.json --tmdir DIR --modelsdir DIR

or, with nothing but this repository:  --synthetic 200 --random-weights
"""
from .common import build_parser, run


def main(argv=None):
    args = build_parser('Print accuracy and time metrics of the skeleton-matching and pose estimation models '
                        '(CMU Panoptic only)').parse_args(argv)
    return run(args, 'mlp')


if __name__ == '__main__':
    main()
