"""GPU tuning aid: per-wave phase time stamps of the fp32 encoder layer kernel (enc_layer4_k) inside the bench workload."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
DEV = torch.device("cuda:0")
buf = torch.zeros(1 << 20, dtype=torch.int64, device=DEV)
os.environ["I2R_ENC_STAMP"] = hex(buf.data_ptr())
os.environ["I2R_ENC_STAMP_FUSED"] = sys.argv[2] if len(sys.argv) > 2 else "1"
REPS = 1
import i2r_amd  # noqa
from i2r_amd import config, synth, arch, engine
name = sys.argv[1] if len(sys.argv) > 1 else "w48_pure_en6"
cfg = config.load_config(name)
sd = synth.make_state_dict(arch.param_spec(cfg))
eng = engine.Engine(cfg, sd, DEV)
length = [4] * 8
x, pm, _ = synth.make_inputs(length, 256, 192, 0)
x, pm = x.to(DEV), pm.to(DEV)
for _ in range(3):
    eng.forward(x, pm, length)
torch.cuda.synchronize()
T = buf.cpu().numpy().reshape(-1, 8).astype(np.float64)
T = T[T[:, 0] > 0]
names = ["lookup+loads+qproj", "attention", "merge", "outproj+LN1", "FFN1", "FFN2+LN2", "store+next KV"]
for rep in range(REPS):
    t = T[:, rep * 8:rep * 8 + 8]
    print("pass %d: waves %d" % (rep, len(t)))
    for i, nm in enumerate(names):
        dd = t[:, i + 1] - t[:, i]
        print("  %-20s mean %8.0f  p10 %8.0f  p90 %8.0f" % (nm, dd.mean(), np.percentile(dd, 10), np.percentile(dd, 90)))
    dd = t[:, 7] - t[:, 0]
    print("  %-20s mean %8.0f  p10 %8.0f  p90 %8.0f" % ("total", dd.mean(), np.percentile(dd, 10), np.percentile(dd, 90)))
