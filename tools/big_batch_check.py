"""GPU: large / odd batches -- more than 64 token groups (the lane-parallel group search takes two trips), 150 single-person images,
and a 200-crop batch; every image's rows must equal the rows it gets when run alone."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import i2r_amd  # noqa
from i2r_amd import config, synth, arch, models
cfg = config.load_config(sys.argv[1] if len(sys.argv) > 1 else "w48_pure_en6")
sd = synth.make_state_dict(arch.param_spec(cfg))
net = eval("models." + cfg.MODEL.NAME + ".get_pose_net")(cfg, is_train=False)
net.load_state_dict(sd, strict=True)
net = net.cuda()
def multi(y):
    return y["multi"] if isinstance(y, dict) else y
for length in ([1] * 150, [3, 1, 2] * 30 + [5] * 4, [2] * 70):
    x, m, _ = synth.make_inputs(length, 256, 192, 3)
    x, m = x.cuda(), m.cuda()
    y = multi(net(x, m, length))
    torch.cuda.synchronize()
    worst, o = 0.0, 0
    for i, n in enumerate(length):
        if i % 17 == 0:
            ya = multi(net(x[o:o + n], m[o:o + n], [n]))
            worst = max(worst, (ya - y[o:o + n]).abs().max().item())
        o += n
    print("S=%d groups=%d finite=%s worst vs alone %.2e" % (sum(length), len(length), bool(torch.isfinite(y).all()), worst))
