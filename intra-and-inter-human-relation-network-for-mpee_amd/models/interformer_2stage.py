"""Older 2-stage wiring used by experiments/coco/interformer_coco_tph_192_p4_b4.yaml.
Mirror of reference lib/models/interformer_2stage.py: class InterFormer (:208), get_pose_net (:425)."""
from .. import arch
from ._base import I2RModule


class InterFormer(I2RModule):
    def __init__(self, cfg, is_train=False, **kwargs):
        super().__init__(cfg, arch.interformer_2stage_spec(cfg))


def get_pose_net(cfg, is_train, **kwargs):
    if is_train:
        raise NotImplementedError("i2r_amd implements the inference path only (is_train=False)")
    return InterFormer(cfg, is_train, **kwargs).eval()
