"""GPU experiment: keep a few CUs free of the high-resolution lanes' big kernels.  The low-resolution lanes of an HRFormer stage are chains
of small dependent launches and the longest lanes of every region (tools/op_list.py); their workgroups queue behind the long-running
workgroups of the fused block kernels of lanes 0 / 1.  Lanes 0 / 1 run on streams created with hipExtStreamCreateWithCUMask that leave
`reserve` CUs out (per 32-bit word of the mask, two layouts tried), lanes 2 / 3 on ordinary streams.
usage: cu_mask_try.py [workload] [precision]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import i2r_amd  # noqa
from i2r_amd import config, synth, arch, engine

DEV = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "hrt_192_p4_b4"
wl = bench.WORKLOADS[name]
prec = sys.argv[2] if len(sys.argv) > 2 else wl["precision"]
cfg = config.load_config(name)
sd = synth.make_state_dict(arch.param_spec(cfg))
length = list(wl["length"])
W, H = cfg.MODEL.IMAGE_SIZE
x, pm, _ = synth.make_inputs(length, H, W, 0)
x, pm = x.to(DEV), pm.to(DEV)
hip = C.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]


def masked_stream(words):
    s = C.c_void_p()
    arr = (C.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), len(words), arr)
    assert rc == 0, "hipExtStreamCreateWithCUMask -> %d" % rc
    return torch.cuda.ExternalStream(s.value, device=DEV)


def run(tag, lane0, lanes):
    eng = engine.Engine(cfg, sd, DEV, precision=prec)
    if lanes is not None:
        eng.side_streams = lanes
    ctx = torch.cuda.stream(lane0) if lane0 is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        for _ in range(5):
            y = eng.forward(x, pm, length)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 30
        for _ in range(K):
            y = eng.forward(x, pm, length)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
    print("%-46s %.3f ms / forward  (checksum %.6f)" % (tag, dt * 1e3, float(y.float().abs().mean()) if not isinstance(y, dict) else float(y["multi"].abs().mean())), flush=True)


full = [0xFFFFFFFF] * 8
run("default streams", None, None)
run("default streams (again)", None, None)
run("all lanes on full-mask ExternalStreams", masked_stream(full), [masked_stream(full) for _ in range(3)])
# one 32-bit word: is the mask applied per XCC (32 CUs each) and replicated?
for words, tag in (([0xFFFFFFFF], "1 word, full"), ([0x0FFFFFFF], "1 word, 28 of 32"), ([0xFFFFFFFF, 0xFFFFFFFF], "2 words, full"),
                   ([0x0FFFFFFF] * 8, "8 words, 28 of 32 each")):
    run("lanes 0/1 masked: " + tag, masked_stream(words), [masked_stream(words), torch.cuda.Stream(), torch.cuda.Stream()])
run("default streams (end)", None, None)
