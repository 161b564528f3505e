"""GPU tuning aid: what the per-step collective costs at N = 1 (one-rank RCCL group): forward only / + async heat-map gather waited one step
later (bench.py's step) / + key-point gather / + gather on a side stream.  usage: gather_cost.py [workload]"""
import os, sys, time, socket
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
import bench
import i2r_amd  # noqa
from i2r_amd import caller, config, synth
from i2r_amd import dist as i2r_dist
name = sys.argv[1] if len(sys.argv) > 1 else "hrt_192_p4_b4"
with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
wl = bench.WORKLOADS[name]
cfg, sd, net = bench.build_net(name, wl["precision"], dev)
W_, H_ = cfg.MODEL.IMAGE_SIZE
length = list(wl["length"])
x, m, _ = synth.make_inputs(length, H_, W_, seed=0)
x, m = x.to(dev), m.to(dev)
counts = [sum(length)]

def fwd():
    y = net(x, m, length)
    return y["multi"] if isinstance(y, dict) else y

def run(mode, steps=40):
    pend = [None]
    def step():
        y = fwd()
        h = None
        if mode == "heat":
            h = i2r_dist.gather_heatmaps_async(y, counts)
        elif mode == "kp":
            p, v = caller.decode(y, None, None, cfg.TEST.BLUR_KERNEL, transform_back=False)
            h = i2r_dist.gather_keypoints(p, v, counts, async_op=True)
        elif mode == "copy":
            h = y.clone()
        if pend[0] is not None and hasattr(pend[0], "wait"):
            pend[0].wait()
        pend[0] = h
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

for mode in ("none", "copy", "heat", "kp", "none"):
    print("%s %-5s %.3f ms / step" % (name, mode, run(mode)))
dist.destroy_process_group()
