"""GPU: heat-map error of the 16-bit MFMA modes against the fp32 CPU oracle, per model family (sets the stated tolerances)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import i2r_amd  # noqa
import i2r_cpu
from _golden import setup, CASES
from i2r_amd import models

for tag in ("w48_l213", "tph_l21", "hrt_l21", "hrt288_l2"):
    cfg, sd, x, m, length, g = setup(tag)
    ref = i2r_cpu.forward(sd, cfg, x, m, length)
    ref = ref if isinstance(ref, dict) else {"multi": ref}
    net = eval("models." + cfg.MODEL.NAME + ".get_pose_net")(cfg, is_train=False)
    net.load_state_dict(sd, strict=True)
    net = net.cuda()
    for prec in ("fp32", "bf16", "fp16"):
        y = net.set_precision(prec)(x.cuda(), m.cuda(), length)
        y = y if isinstance(y, dict) else {"multi": y}
        msg = []
        for k in y:
            d = (y[k].cpu() - ref[k])
            msg.append("%s: max-abs %.3e  rel-to-max %.3e  rms/rms %.3e" % (k, d.abs().max().item(), d.abs().max().item() / ref[k].abs().max().item(),
                                                                            d.pow(2).mean().sqrt().item() / ref[k].pow(2).mean().sqrt().item()))
        print(tag, prec, " | ".join(msg))
