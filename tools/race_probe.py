"""GPU probe for the part-batch forward (Engine._forward): a one-program forward, then the FIRST part-batch forward of the same
engine, per-image error against the oracle.  VARIANT env: zeros (torch.empty -> zeros in the engine), sync_build (device sync after
every program build)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import i2r_amd
from i2r_amd import synth, models, engine
import i2r_cpu
from _golden import setup
variant = os.environ.get("VARIANT", "")
if "zeros" in variant:
    _empty = torch.empty
    engine.torch.empty = lambda *a, **k: torch.zeros(*a, **k)
cfg, sd, _, _, _, _ = setup("tph_l21")
full = [6, 4, 4, 2, 2, 1, 1, 1, 2, 5]
x, m, _ = synth.make_inputs(full, 256, 192, seed=3)
ref = i2r_cpu.forward(sd, cfg, x, m, full)
net = models.interformer.get_pose_net(cfg, is_train=False); net.load_state_dict(sd, strict=True); net = net.cuda()
eng = net.engine()
import time
if "sync_" in variant or "pre_" in variant or "time" in variant:
    b0 = eng._build
    def b1(*a, **k):
        on_default = torch.cuda.current_stream() == torch.cuda.default_stream()
        if on_default and "pre_sync" in variant:
            torch.cuda.synchronize()
        if on_default and "pre_sleep" in variant:
            time.sleep(0.5)
        t0 = time.perf_counter()
        r = b0(*a, **k)
        if "time" in variant:
            print("build on %s stream: %.1f ms" % ("default" if on_default else "side", (time.perf_counter() - t0) * 1e3))
        on_default = torch.cuda.current_stream() == torch.cuda.default_stream()
        if "sync_build" in variant or ("sync_a" in variant and on_default) or ("sync_b" in variant and not on_default):
            torch.cuda.synchronize()
        return r
    eng._build = b1
import contextlib
ctx = (lambda: torch.cuda.stream(CTX)) if "ctx" in variant else contextlib.nullcontext
CTX = torch.cuda.Stream()
def report(tag):
    with ctx():
        y = net(x.cuda(), m.cuda(), full)
    torch.cuda.synchronize()
    o, errs = 0, []
    for n in full:
        errs.append("%.0e" % (y["single"][o:o + n].cpu() - ref["single"][o:o + n]).abs().max().item())
        o += n
    print(variant or "plain", tag, len(eng.last_programs), " ".join(errs), flush=True)
eng.SPLIT_MIN_CROPS = 10 ** 9
report("one  ")
eng.SPLIT_MIN_CROPS = 24
report("split")
report("again")
