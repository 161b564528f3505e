"""Bare HRNet-W48-S backbone. Mirror of reference lib/models/hrnet.py: class HRNet (:275), forward (:419-446) = stem, layer1,
stages 2-3 and the 1x1 `reduce` conv on the LOWEST-resolution branch -> [S, DIM_MODEL, H/16, W/16]; the `final_layer` the
constructor builds (:317-323) is never applied.  get_pose_net(cfg, is_train) (:480).  Consumed by models.interformer through
models.backbone.build_backbone when MODEL.SINGLEFORMER is unset (interformer.py:143-144)."""
import torch

from .. import arch
from ._base import I2RModule


class HRNet(I2RModule):
    def __init__(self, cfg, **kwargs):
        super().__init__(cfg, arch.hrnet_spec(cfg, ""))

    def engine(self):
        eng = super().engine()
        assert eng.name == "hrnet"
        return eng

    def _engine_name(self):
        return "hrnet"

    def forward(self, x):
        """x [S, 3, H, W] -> reduce(y_list[-1]) [S, d, H/16, W/16] (hrnet.py:419-446)."""
        with torch.no_grad():
            return self.engine().forward_backbone(x)


def get_pose_net(cfg, is_train, **kwargs):
    if is_train:
        raise NotImplementedError("i2r_amd implements the inference path only (is_train=False)")
    return HRNet(cfg, **kwargs).eval()
