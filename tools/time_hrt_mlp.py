"""GPU tuning aid: time the fused HRFormer MLP-block kernel (i2r_hrt_mlp_block), both variants, on the branch shapes of configs 4 / 5.
usage: time_hrt_mlp.py [bf16|fp16] [crops] [192|288]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import i2r_amd  # noqa
from i2r_amd import cabi, engine, synth
if os.environ.get("I2R_TOOL_LIB"):  # an A/B library variant (tools/ab/build_variant.sh)
    cabi._LIB = cabi.load_library(os.path.join(ROOT, os.environ["I2R_TOOL_LIB"]))
DEV = torch.device("cuda:0")
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
big = len(sys.argv) > 3 and sys.argv[3] == "288"
shapes = ((78, 96, 72), (156, 48, 36), (312, 24, 18)) if big else ((78, 64, 48), (156, 32, 24), (312, 16, 12))
for c, h, w, variant in [sh + (v,) for sh in shapes for v in (1, 2) if v == 2 or sh[0] <= 156]:
    hid = 4 * c
    sd = {"b.norm2.weight": torch.ones(c), "b.norm2.bias": torch.zeros(c),
          "b.mlp.fc1.weight": torch.from_numpy(synth._sym(1, "f1%d" % c, (hid, c, 1, 1), 0.1)), "b.mlp.fc1.bias": torch.zeros(hid),
          "b.mlp.dw3x3.weight": torch.from_numpy(synth._sym(1, "dw%d" % c, (hid, 1, 3, 3), 0.3)), "b.mlp.dw3x3.bias": torch.zeros(hid),
          "b.mlp.fc2.weight": torch.from_numpy(synth._sym(1, "f2%d" % c, (c, hid, 1, 1), 0.05)), "b.mlp.fc2.bias": torch.zeros(c)}
    for k, ch in (("norm1", hid), ("norm2", hid), ("norm3", c)):
        sd.update({"b.mlp.%s.weight" % k: torch.ones(ch), "b.mlp.%s.bias" % k: torch.zeros(ch), "b.mlp.%s.running_mean" % k: torch.zeros(ch),
                   "b.mlp.%s.running_var" % k: torch.ones(ch)})
    P = engine.Program(DEV)
    mb = engine.Packer(sd, DEV, prec).mlp_block_lp("b", c)
    x = P.alloc(n, h, w, c)
    x.t.normal_()
    y = x
    for _ in range(4):
        y = P.hrt_mlp(y, mb, variant=variant)
    P.finalize()
    for _ in range(3):
        P.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        P.run()
    e1.record()
    torch.cuda.synchronize()
    print("C=%d %dx%d n=%d variant %d: %.1f us per launch" % (c, h, w, n, variant, e0.elapsed_time(e1) / 40 * 1e3))
