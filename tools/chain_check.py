"""GPU: repeat the vanilla forward with the persistent chain launches and report the chains' error words + result drift."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import i2r_amd  # noqa
from i2r_amd import config, synth, arch, engine
DEV = torch.device("cuda:0")
cfg = config.load_config("w48_pure_en6")
sd = synth.make_state_dict(arch.param_spec(cfg))
os.environ["I2R_TUNING"] = "1"  # engine._tune reads the A/B switches only with it
os.environ["I2R_CONV_CHAIN"] = "0"
ref_eng = engine.Engine(cfg, sd, DEV)
os.environ["I2R_CONV_CHAIN"] = "1"
eng = engine.Engine(cfg, sd, DEV)
for S_len in ([4] * 8, [2, 1], [3], [1] * 5):
    x, pm, _ = synth.make_inputs(S_len, 256, 192, 0)
    x, pm = x.to(DEV), pm.to(DEV)
    ref = ref_eng.forward(x, pm, S_len).clone()
    bad, worst, codes = 0, 0.0, set()
    for it in range(30):
        y = eng.forward(x, pm, S_len)
        torch.cuda.synchronize()
        P = eng.programs[(sum(S_len), 256, 192, False)][0]
        for f, n in getattr(P, "chain_flags", []):
            codes.add(int(f[n].item()))
        d = (y - ref).abs().max().item()
        worst = max(worst, d)
        bad += d != 0.0
    print("length", S_len, ": runs differing from the per-layer result", bad, "/ 30, worst", worst, "error words", sorted(codes))
