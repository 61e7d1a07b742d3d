"""pytransform3d.transform_manager.TransformManager (1.9.1) — the two paths the
reference uses: a stored edge is returned as is, the reverse edge is
``numpy.linalg.inv`` of the stored matrix (pytransform3d.transformations.invert_transform)."""
import numpy as np


class TransformManager:
    def __init__(self, strict_check=True, check=True):
        self.strict_check = strict_check
        self.check = check
        self.transforms = {}

    def __setstate__(self, state):
        self.__dict__.update(state)

    def add_transform(self, from_frame, to_frame, A2B):
        self.transforms[(from_frame, to_frame)] = np.asarray(A2B, dtype=np.float64)
        return self

    def get_transform(self, from_frame, to_frame):
        if (from_frame, to_frame) in self.transforms:
            return self.transforms[(from_frame, to_frame)]
        if (to_frame, from_frame) in self.transforms:
            return np.linalg.inv(self.transforms[(to_frame, from_frame)])
        raise KeyError('Cannot compute path from frame %r to frame %r' % (from_frame, to_frame))
