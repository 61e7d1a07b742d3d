"""API mirror of the reference's utils/mlp.py: same module tree (so ``pose_estimator.pytorch``'s
``model_state_dict`` loads unchanged, keys ``layers.{1,3,...,17}.{weight,bias}``); forward() runs
the chain of fp32 MFMA GEMMs with fused bias + LeakyReLU(0.1) (mpe_mlp_forward)."""
import torch
from torch import nn

from . import runtime


class PoseEstimatorMLP(nn.Module):
    def __init__(self, input_dimensions, output_dimensions):
        super().__init__()
        print('MLP input size', input_dimensions)
        negative_slope = 0.1
        widths = [input_dimensions, 3072, 3072, 2048, 2048, 1024, 1024, 1024, 1024]
        mods = [nn.Flatten()]
        for a, b in zip(widths[:-1], widths[1:]):
            mods += [nn.Linear(a, b), nn.LeakyReLU(negative_slope=negative_slope)]
        mods.append(nn.Linear(widths[-1], output_dimensions))
        self.layers = nn.Sequential(*mods)
        self.negative_slope = negative_slope
        self._engine = None
        self._version = None

    def _ensure_engine(self):
        w = self.__dict__.get('_param_watch')
        if w is None:
            w = self.__dict__['_param_watch'] = runtime.ParamWatch(self)      # (gat2.GAT2._state_version)
        ver = w.version()
        if self._engine is not None and ver == self._version:
            return self._engine
        if self._engine is not None:
            self._engine.close()
        eng = runtime.new_engine(max_frames=int(runtime.os.environ.get('MPE_MLP_MAX_ROWS', '64')))
        eng.load_mlp({k: v.detach().cpu() for k, v in self.state_dict().items()}, self.negative_slope)
        self._engine, self._version = eng, ver
        return eng

    def forward(self, x):
        eng = self._ensure_engine()
        x = x.reshape(x.shape[0], -1).to(eng.device, torch.float32)
        return eng.mlp_forward(x)
