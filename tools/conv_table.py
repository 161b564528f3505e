"""GPU: per-launch table of the conv launches of one config (kernel variant, shapes, time, TFLOP/s), heaviest first."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import torch
import i2r_amd  # noqa
from i2r_amd import config, synth, arch, engine, cabi
import bench
DEV = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "hrt_192_p4_b4"
cfg = config.load_config(name)
sd = synth.make_state_dict(arch.param_spec(cfg))
eng = engine.Engine(cfg, sd, DEV)
length = [4] * 8
H, W = cfg.MODEL.IMAGE_SIZE[1], cfg.MODEL.IMAGE_SIZE[0]
x, pm, _ = synth.make_inputs(length, H, W, 0)
eng.forward(x.to(DEV), pm.to(DEV), length)
P = next(iter(eng.programs.values()))[0]
L = cabi.lib()
cur = torch.cuda.current_stream().cuda_stream
streams = (C.c_void_p * 4)(cur, cur, cur, cur)
rows = {}
for rep in range(3):
    for i, (kind, lane, st) in enumerate(P.ops):
        if kind not in (cabi.OP_CONV, cabi.OP_CONV_GROUP):
            continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        cabi.check(L.i2r_run_program(C.cast(C.byref(P._c_ops, i * C.sizeof(cabi.Op)), C.POINTER(cabi.Op)), 1, streams, None), "op")
        e1.record()
        torch.cuda.synchronize()
        if rep == 0:
            continue
        ms = [st] if kind == cabi.OP_CONV else [st.d[j].contents for j in range(st.n)]
        nm, fl = bench._op_name_flop(kind, st)
        key = (nm, tuple((m.cin, m.cout, m.ntaps, m.stride, m.conv_h, m.conv_w, m.tile_h, m.tile_w, m.rep) for m in ms))
        r = rows.setdefault(key, [0, 0.0, fl])
        r[0] += 1
        r[1] += e0.elapsed_time(e1)
tot = sum(r[1] for r in rows.values()) / 2
print("conv launches: %.2f ms per step" % tot)
for (nm, shp), (n, ms, fl) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print("%6.3f ms/step  x%-3d %5.1f TF  %-28s %s" % (ms / 2, n // 2, fl * n / ms / 1e9 / 1, nm, " + ".join("%d->%d k%d s%d @%dx%d t%dx%d%s" % (a, b, k, s, h, w, th, tw, " up%d" % rp if rp > 1 else "") for a, b, k, s, h, w, th, tw, rp in shp)))
