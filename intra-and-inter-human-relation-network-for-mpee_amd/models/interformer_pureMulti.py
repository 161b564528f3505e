"""Vanilla I2R-Net (HRNet-W48-S -> inter-human encoder -> shared deconv x2 -> head).
Mirror of reference lib/models/interformer_pureMulti.py: class TransPoseH (:419), get_pose_net (:816)."""
from .. import arch
from ._base import I2RModule


class TransPoseH(I2RModule):
    def __init__(self, cfg, **kwargs):
        super().__init__(cfg, arch.vanilla_spec(cfg))


def get_pose_net(cfg, is_train, **kwargs):
    if is_train:
        raise NotImplementedError("i2r_amd implements the inference path only (is_train=False)")
    return TransPoseH(cfg, **kwargs).eval()
