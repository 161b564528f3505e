"""TransPose-H intra-human stage. Mirror of reference lib/models/transpose_h.py: TransPoseH (:416),
get_pose_net(cfg, is_train, pretrained_path, is_end2end) (:691).  Stand-alone use returns
(features, heatmaps) in the reference; here it is only built as InterFormer.singleformer, whose parameters
live under the ``singleformer.`` prefix of the 2-stage module (models/interformer.py)."""
from .. import arch


def param_spec(cfg, prefix=""):
    return arch.transpose_h_spec(cfg, prefix)


def get_pose_net(cfg, is_train, pretrained_path="", is_end2end=False, **kwargs):
    raise NotImplementedError(
        "transpose_h is consumed through models.interformer (MODEL.SINGLEFORMER: transpose_h); "
        "a stand-alone TransPose-H forward is not part of the I2R-Net inference path")
