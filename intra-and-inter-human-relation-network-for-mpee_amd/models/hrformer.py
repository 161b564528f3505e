"""HRFormer-B intra-human stage. Mirror of reference lib/models/hrformer.py: class HRFormer (:2470-2480) = HRT backbone
(:1735-2092) + TopDownSimpleHead with 0 deconvs and a 1x1 final conv (:2215-2348); forward returns (x_tmp[0], heatmaps) =
(branch-0 features [S, 78, H/4, W/4], heatmaps [S, J, H/4, W/4]); get_pose_net(cfg, is_train, model_path, e2e_flag) (:2487) with the
architecture constants of :2489-2525 (arch_hrformer.STAGES) -- the factory InterFormer.__init__ reaches through
eval('models.' + cfg.MODEL.SINGLEFORMER + '.get_pose_net') (interformer.py:139)."""
import torch

from .. import arch_hrformer
from ._base import I2RModule


class HRFormer(I2RModule):
    def __init__(self, cfg, **kwargs):
        super().__init__(cfg, arch_hrformer.hrformer_spec(cfg, ""))

    def _engine_name(self):
        return "hrformer"

    def forward(self, x):
        """x [S, 3, H, W] -> (features, heatmaps)  (hrformer.py:2477-2480)"""
        with torch.no_grad():
            return self.engine().forward_single(x)


def get_pose_net(cfg, is_train, model_path="", e2e_flag=False, **kwargs):
    if is_train:
        raise NotImplementedError("i2r_amd implements the inference path only (is_train=False)")
    return HRFormer(cfg, **kwargs).eval()
