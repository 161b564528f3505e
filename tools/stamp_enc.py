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
raw = buf.cpu().numpy().reshape(-1, 8).astype(np.float64)
names = ["lookup+loads+qproj", "attention", "merge(+hand-off)", "outproj+LN1", "FFN1", "FFN2+LN2", "store+next KV"]
nblk = 512 if raw[511 * 4, 0] > 0 else int((raw[:, 0] > 0).sum() // 4)
raw = raw[:nblk * 4].reshape(nblk, 4, 8)
blk = np.arange(nblk)
q = blk >> 3
split_launch = nblk == 512 and name == "w48_pure_en6"
kinds = {"all": np.ones(nblk, bool)}
if split_launch:  # partial key split: per XCD 32 whole tiles first, then the halves; a half that left early has stamp 7 == 0
    early = raw[:, 0, 7] == 0
    hf = os.environ.get("HALVES_FIRST", "1") == "1"
    isw = (q >= 32) if hf else (q < 32)
    kinds = {"whole": isw, "half, finishes the tile": ~isw & ~early, "half, leaves early": ~isw & early}
t0 = raw[:, :, 0][raw[:, :, 0] > raw[:, :, 0].max() - 1e6].min()  # (the last stamped launch)
for kn, sel in kinds.items():
    t = raw[sel].reshape(-1, 8)
    last = 3 if kn.endswith("early") else 7
    print("%s: %d workgroups; start %.0f .. %.0f, end %.0f .. %.0f (cycles after the first wave of the launch)" % (
        kn, sel.sum(), t[:, 0].min() - t0, t[:, 0].max() - t0, t[:, last].min() - t0, t[:, last].max() - t0))
    for i, nm in enumerate(names[:last]):
        dd = t[:, i + 1] - t[:, i]
        print("  %-20s mean %8.0f  p10 %8.0f  p90 %8.0f" % (nm, dd.mean(), np.percentile(dd, 10), np.percentile(dd, 90)))
    dd = t[:, last] - t[:, 0]
    print("  %-20s mean %8.0f  p10 %8.0f  p90 %8.0f" % ("total", dd.mean(), np.percentile(dd, 10), np.percentile(dd, 90)))
