"""2-stage I2R-Net (first stage -> max-pool -> inter-human encoder -> DeConv -> residual -> head).
Mirror of reference lib/models/interformer.py: class InterFormer (:130), get_pose_net (:326).
forward returns {'single','multi'} when INTER_SUPERVISION and not SINGLEFORMER_FIX (:320-323)."""
from .. import arch
from ._base import I2RModule


class InterFormer(I2RModule):
    def __init__(self, cfg, is_train=False, **kwargs):
        super().__init__(cfg, arch.interformer_spec(cfg))


def get_pose_net(cfg, is_train, **kwargs):
    if is_train:
        raise NotImplementedError("i2r_amd implements the inference path only (is_train=False)")
    return InterFormer(cfg, is_train, **kwargs).eval()
