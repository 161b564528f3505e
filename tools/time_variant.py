"""GPU: ms per forward of a base config under KEY VALUE overrides (the variant configs of tests/_golden.py VARIANTS), 32 crops.
usage: python tools/time_variant.py <config> [KEY VALUE ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ast
import torch
import i2r_amd  # noqa
from i2r_amd import config, synth, arch, engine
DEV = torch.device("cuda:0")
name = sys.argv[1]
opts = [ast.literal_eval(v) if i % 2 and v[:1] in "0123456789[TF" else v for i, v in enumerate(sys.argv[2:])]
cfg = config.load_config(name, opts)
sd = synth.make_state_dict(arch.param_spec(cfg))
eng = engine.Engine(cfg, sd, DEV)
length = [4] * 8
x, pm, _ = synth.make_inputs(length, cfg.MODEL.IMAGE_SIZE[1], cfg.MODEL.IMAGE_SIZE[0], 0)
x, pm = x.to(DEV), pm.to(DEV)
for _ in range(3):
    eng.forward(x, pm, length)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    eng.forward(x, pm, length)
e1.record()
torch.cuda.synchronize()
print("%s %s: %.3f ms per forward of %d crops" % (name, opts, e0.elapsed_time(e1) / 10, sum(length)))
