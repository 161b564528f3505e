"""GPU tuning aid: time the fused HRFormer attention-block kernel (i2r_hrt_attn_block), both variants, on the branch shapes of configs 4 / 5.
usage: time_hrt_attn.py [bf16|fp16] [crops] [192|288]"""
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
shapes = ((78, 2, 96, 72), (156, 4, 48, 36), (312, 8, 24, 18), (624, 16, 12, 9)) if big else ((78, 2, 64, 48), (156, 4, 32, 24), (312, 8, 16, 12), (624, 16, 8, 6))
for c, heads, h, w, variant in [sh + (v,) for sh in shapes for v in (1, 2) if v == 2 or sh[0] <= 156]:
    sd = {"b.norm1.weight": torch.ones(c), "b.norm1.bias": torch.zeros(c)}
    for k in ("q_proj", "k_proj", "v_proj", "out_proj"):
        sd["b.attn.attn.%s.weight" % k] = torch.from_numpy(synth._sym(1, k + str(c), (c, c), 0.1))
        sd["b.attn.attn.%s.bias" % k] = torch.zeros(c)
    P = engine.Program(DEV)
    ab = engine.Packer(sd, DEV, prec).attn_block_lp("b", c, heads)
    x = P.alloc(n, h, w, c)
    x.t.normal_()
    y = x
    for _ in range(4):
        y = P.hrt_attn(y, ab, variant=variant)
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
