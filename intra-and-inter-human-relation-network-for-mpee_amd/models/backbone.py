"""Mirror of reference lib/models/backbone.py: HRNetBackbone (:9) wraps hrnet.get_pose_net as `.body`; build_backbone (:18)."""
import torch.nn as nn

from . import hrnet


class HRNetBackbone(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.body = hrnet.HRNet(cfg).eval()  # (the reference passes is_train=True only to trigger init_weights, backbone.py:12)

    def forward(self, x):
        return self.body(x)


def build_backbone(cfg):
    return HRNetBackbone(cfg)
