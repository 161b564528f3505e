"""GPU tuning aid: time of the encoder stacks of one workload's program (bench.stack_timing) with a tuning library and its env switches
(I2R_ENC_QT = 1 | 2 forces one / two query tiles per workgroup).  usage: I2R_TOOL_LIB=tools/ab/lib_enc.so python tools/enc_ab.py [workload] [precision]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import i2r_amd  # noqa
from i2r_amd import cabi
if os.environ.get("I2R_TOOL_LIB"):
    cabi._LIB = cabi.load_library(os.path.join(ROOT, os.environ["I2R_TOOL_LIB"]))
from i2r_amd import config, synth, arch, engine
DEV = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "tph_192_p6_b4"
prec = sys.argv[2] if len(sys.argv) > 2 else "fp32"
wl = bench.WORKLOADS[name]
cfg = config.load_config(name)
sd = synth.make_state_dict(arch.param_spec(cfg))
eng = engine.Engine(cfg, sd, DEV, precision=prec)
length = wl["length"]
x, pm, _ = synth.make_inputs(length, cfg.MODEL.IMAGE_SIZE[1], cfg.MODEL.IMAGE_SIZE[0], 0)
for _ in range(3):
    y = eng.forward(x.to(DEV), pm.to(DEV), length)
torch.cuda.synchronize()
P = next(iter(eng.programs.values()))[0]
ms = bench.stack_timing(P, prec)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    eng.forward(x.to(DEV), pm.to(DEV), length)
e1.record(); torch.cuda.synchronize()
yy = y
while not torch.is_tensor(yy):
    yy = list(yy.values())[-1] if isinstance(yy, dict) else yy[-1]
print("%s %s: encoder stacks %.3f ms per forward, forward %.3f ms  (I2R_ENC_QT=%s I2R_ENC_QF=%s)  output checksum %.6f max %.4f" % (name, prec, ms, e0.elapsed_time(e1) / 10, os.environ.get("I2R_ENC_QT", "-"), os.environ.get("I2R_ENC_QF", "-"), yy.float().abs().mean().item(), yy.float().abs().max().item()))
