"""In-container harness that imports the UPSTREAM reference on CPU (test infrastructure only).

Only used in the build container (where /root/reference exists) to
  (i)  validate the CPU restatement in oracle/i2r_cpu.py layer by layer, and
  (ii) generate the golden vectors committed under tests/golden/ (see oracle/make_golden.py).

Nothing here travels to the GPU box in a usable form: it needs /root/reference. The reference
needs a few packages this image lacks (yacs, timm, mmcv, torchvision); they are only used as
layer *builders* / init helpers, so tiny stand-in modules returning plain torch.nn
layers are registered in sys.modules before `import models`. The stand-ins carry no arithmetic.
"""
import logging
import os
import sys
import types

import torch
import torch.nn as nn

REF_ROOT = os.environ.get("I2R_REFERENCE_ROOT", "/root/reference")
REF_LIB = os.path.join(REF_ROOT, "lib")


def have_reference():
    return os.path.isdir(os.path.join(REF_LIB, "models"))


def _mod(name):
    m = types.ModuleType(name)
    sys.modules[name] = m
    return m


def _install_stubs():
    if "timm" not in sys.modules:
        timm = _mod("timm")
        timm_models = _mod("timm.models")
        layers = _mod("timm.models.layers")
        timm.models = timm_models
        timm_models.layers = layers

        def to_2tuple(x):
            return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

        def trunc_normal_(t, mean=0.0, std=1.0, a=-2.0, b=2.0):
            return nn.init.trunc_normal_(t, mean=mean, std=std, a=a, b=b)

        layers.to_2tuple = to_2tuple
        layers.trunc_normal_ = trunc_normal_

    if "torchvision" not in sys.modules:
        tv = _mod("torchvision")
        tvm = _mod("torchvision.models")
        tv.models = tvm

        class _BB(nn.Module):  # torchvision BasicBlock layout (conv-bn-relu-conv-bn +x relu)
            def __init__(self, c):
                super().__init__()
                self.conv1 = nn.Conv2d(c, c, 3, 1, 1, bias=False)
                self.bn1 = nn.BatchNorm2d(c)
                self.relu = nn.ReLU(inplace=True)
                self.conv2 = nn.Conv2d(c, c, 3, 1, 1, bias=False)
                self.bn2 = nn.BatchNorm2d(c)

            def forward(self, x):
                o = self.relu(self.bn1(self.conv1(x)))
                o = self.bn2(self.conv2(o))
                return self.relu(o + x)

        class _R18(nn.Module):  # only children()[:5] are used by position_embedding.py:16-17
            def __init__(self):
                super().__init__()
                self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
                self.bn1 = nn.BatchNorm2d(64)
                self.relu = nn.ReLU(inplace=True)
                self.maxpool = nn.MaxPool2d(3, 2, 1)
                self.layer1 = nn.Sequential(_BB(64), _BB(64))
                self.layer2 = nn.Identity()

        def resnet18(pretrained=False, **kw):
            return _R18()

        tvm.resnet18 = resnet18

    if "mmcv" not in sys.modules:
        mmcv = _mod("mmcv")
        cnn = _mod("mmcv.cnn")
        runner = _mod("mmcv.runner")
        ckpt = _mod("mmcv.runner.checkpoint")
        utils = _mod("mmcv.utils")
        parrots = _mod("mmcv.utils.parrots_wrapper")
        mmcv.cnn, mmcv.runner, mmcv.utils = cnn, runner, utils
        runner.checkpoint = ckpt
        utils.parrots_wrapper = parrots

        def build_conv_layer(cfg, *a, **k):
            return nn.Conv2d(*a, **k)

        def build_norm_layer(cfg, num_features, postfix=""):
            typ = (cfg or {}).get("type", "BN")
            assert typ in ("BN", "SyncBN"), typ
            return "bn" + str(postfix), nn.BatchNorm2d(num_features)

        def build_upsample_layer(cfg, *a, **k):
            cfg = dict(cfg)
            typ = cfg.pop("type")
            assert typ == "deconv", typ
            cfg.update(k)
            return nn.ConvTranspose2d(*a, **cfg)

        def constant_init(module, val, bias=0):
            if hasattr(module, "weight") and module.weight is not None:
                nn.init.constant_(module.weight, val)
            if hasattr(module, "bias") and module.bias is not None:
                nn.init.constant_(module.bias, bias)

        def normal_init(module, mean=0, std=1, bias=0):
            if hasattr(module, "weight") and module.weight is not None:
                nn.init.normal_(module.weight, mean, std)
            if hasattr(module, "bias") and module.bias is not None:
                nn.init.constant_(module.bias, bias)

        def kaiming_init(module, *a, **k):
            nn.init.kaiming_normal_(module.weight)

        cnn.build_conv_layer = build_conv_layer
        cnn.build_norm_layer = build_norm_layer
        cnn.build_upsample_layer = build_upsample_layer
        cnn.constant_init = constant_init
        cnn.normal_init = normal_init
        cnn.kaiming_init = kaiming_init
        ckpt.load_state_dict = lambda module, sd, strict=False, logger=None: module.load_state_dict(sd, strict=strict)
        utils.get_logger = lambda name, log_file=None, log_level=logging.INFO: logging.getLogger(name)
        parrots._BatchNorm = nn.modules.batchnorm._BatchNorm


_REF = None


def import_reference():
    """Returns the reference's `models` package (lib/models/__init__.py:16-23)."""
    global _REF
    if _REF is not None:
        return _REF
    if not have_reference():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    _install_stubs()
    # the reference package is called `models` / `utils`; make sure nothing of ours shadows it
    for name in ("models", "utils"):
        if name in sys.modules and not getattr(sys.modules[name], "__file__", "").startswith(REF_LIB):
            raise RuntimeError("module %r already imported from elsewhere" % name)
    if REF_LIB not in sys.path:
        sys.path.insert(0, REF_LIB)
    import models  # noqa

    _REF = models
    return models


def build_reference_model(cfg):
    """cfg: an attr-dict config (i2r_amd.config.load_config) -> reference nn.Module in eval()."""
    models = import_reference()
    net = eval("models." + cfg.MODEL.NAME + ".get_pose_net")(cfg, is_train=False)
    net.eval()
    return net
