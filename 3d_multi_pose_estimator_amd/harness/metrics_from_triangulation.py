"""Counterpart of the reference's test/metrics_from_triangulation.py on the MI355X path
(GAT matching -> clustering -> pairwise DLT with the 5 cm median filter)."""
from .common import build_parser, run


def main(argv=None):
    args = build_parser('Print accuracy and time metrics of the skeleton-matching model and triangulation '
                        '(CMU Panoptic only)').parse_args(argv)
    return run(args, 'tri')


if __name__ == '__main__':
    main()
