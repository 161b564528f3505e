#!/usr/bin/env python
"""bench.py -- throughput of the I2R-Net inference hot path on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" is ONE forward of the vanilla I2R-Net (HRNet-W48-S, 256x192, 6 encoder layers, fp32) over one
synthetic batch of 8 images x 4 persons = 32 crops per GPU (BASELINE.json configs[1]); inputs are resident in
HBM before the timed region; weights are the seeded synthetic set.  With N > 1 every rank runs its own 8 images
(weak scaling, images are the independent unit) and the per-crop heatmaps are all-gathered over RCCL each step.

One JSON line is printed by rank 0:
  value      = crops/s of the whole job (all ranks), from the max-over-ranks wall time of exactly K steps
  roofline   = the dominant kernel (fp32-MFMA implicit-GEMM conv): algorithmic FLOPs per launch / average launch
               duration, both from a per-launch HIP-event timing pass inside this script, vs the 157.3 TFLOP/s fp32
               matrix peak (MI355X_MICROARCH.md)
  cpu_baseline = the CPU oracle (oracle/i2r_cpu.py, a port of the reference forward) timed on this host, rank 0 only
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import i2r_amd  # noqa: E402,F401
from i2r_amd import arch, cabi, config, models, synth  # noqa: E402
from i2r_amd import dist as i2r_dist  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, 256 CU x 2.4 GHz
IMAGES_PER_GPU, PERSONS = 8, 4
GFLOP_PER_CROP = 19.085 + 0.0849 * PERSONS  # BASELINE.md section 3 (N = 4 persons / image)


def conv_kernel_name(members):
    """rocprof-visible instantiation name conv_igemm_f32<MT, NT, CAP, PF>, asked from the library itself."""
    import ctypes as C
    arr = (C.POINTER(cabi.ConvDesc) * len(members))(*[C.pointer(m) for m in members])
    buf = C.create_string_buffer(96)
    cabi.check(cabi.lib().i2r_conv_kernel_name(arr, len(members), buf, 96), "i2r_conv_kernel_name")
    return buf.value.decode()


def conv_flop(d):
    return 2.0 * d.n_img * d.conv_h * d.conv_w * d.cout * d.cin * d.ntaps


def _op_name_flop(kind, st):
    if kind == cabi.OP_CONV:
        return conv_kernel_name([st]), conv_flop(st)
    if kind == cabi.OP_CONV_GROUP:
        members = [st.d[i].contents for i in range(st.n)]
        return conv_kernel_name(members), sum(conv_flop(m) for m in members)
    if kind == cabi.OP_CONV_CHAIN:
        members = [st.descs[i].contents for i in range(st.n_layers * st.n_members)]
        return "conv_chain_f32<%d, %d, %d, %d>" % (st.mt, st.nt, st.cap, st.pf), sum(conv_flop(m) for m in members)
    name = {cabi.OP_STEM: "stem_conv_k", cabi.OP_MAXPOOL: "maxpool_k", cabi.OP_HEAD: "head_k",
            cabi.OP_ENC_KV: "enc_kv_k", cabi.OP_ENC_LAYER: "enc_layer4_k", cabi.OP_LAYERNORM: "layernorm_k",
            cabi.OP_WINATTN: "window_attn_k", cabi.OP_DWCONV: "dwconv3x3_k", cabi.OP_UPSAMPLE: "upsample_add_k"}[kind]
    return name, 0.0


def per_launch_timing(program, reps=3):
    """Replay the program with HIP events on the launch stream between RUNS of consecutive launches of the same kernel (e.g. the six
    encoder layers, the 8 convs of a branch block) -> per-kernel (launch count, total ms, total flop).  Timing a run as a whole
    keeps the kernels back to back as in the real step; one event pair per launch would add its own few microseconds to each."""
    import ctypes as C
    L = cabi.lib()
    cur = torch.cuda.current_stream().cuda_stream
    streams = (C.c_void_p * 4)(cur, cur, cur, cur)
    ops = [(i, kind, st) for i, (kind, lane, st) in enumerate(program.ops) if kind not in (cabi.OP_FORK, cabi.OP_JOIN)]
    named = [(i,) + _op_name_flop(kind, st) for i, kind, st in ops]  # single-stream pass: lanes collapse onto the current stream
    runs = []  # [name, [op indices], flop]
    for i, name, flop in named:
        if runs and runs[-1][0] == name:
            runs[-1][1].append(i)
            runs[-1][2] += flop
        else:
            runs.append([name, [i], flop])
    stats = {}
    for rep in range(reps + 1):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(runs) + 1)]
        evs[0].record()
        for r, (name, idx, flop) in enumerate(runs):
            for i in idx:
                cabi.check(L.i2r_run_program(C.cast(C.byref(program._c_ops, i * C.sizeof(cabi.Op)), C.POINTER(cabi.Op)), 1,
                                             streams, None), "op %d" % i)
            evs[r + 1].record()
        torch.cuda.synchronize()
        if rep == 0:
            continue  # warm-up pass
        for r, (name, idx, flop) in enumerate(runs):
            s = stats.setdefault(name, [0, 0.0, 0.0])
            s[0] += len(idx)
            s[1] += evs[r].elapsed_time(evs[r + 1])
            s[2] += flop
    return stats, reps


def hbm_traffic():
    """HBM bytes per launch of the dominant (grouped stage-3 conv) kernel from the committed PMC passes
    (profiles/round1_hbm_traffic.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE doubled per
    MI355X_MICROARCH.md).  bench.py itself cannot collect PMC counters; null when the profile is absent."""
    try:
        with open(os.path.join(ROOT, "profiles", "round1_hbm_traffic.json")) as f:
            return round(json.load(f)["hbm_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(cfg, sd, budget_s=20.0):
    """The CPU oracle (a port of the reference forward) timed on this host; bounded to ~budget_s seconds."""
    import i2r_cpu
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    threads = min(cores, 32)  # torch CPU convs stop scaling (and oversubscribe badly) far below 256 threads
    torch.set_num_threads(threads)
    x, m, length = synth.make_inputs([1], 256, 192)
    t0 = time.perf_counter()
    i2r_cpu.forward(sd, cfg, x, m, length)  # warm-up + cost probe on ONE crop
    probe = time.perf_counter() - t0
    persons = PERSONS if probe * PERSONS * 3 < budget_s else 1
    x, m, length = synth.make_inputs([persons], 256, 192)
    n, t0 = 0, time.perf_counter()
    while True:
        i2r_cpu.forward(sd, cfg, x, m, length)
        n += 1
        if time.perf_counter() - t0 > budget_s * 0.6 or n >= 40:
            break
    dt = time.perf_counter() - t0
    return {"value": round(n * persons / dt, 3), "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": "%d forwards of 1 image x %d person(s), fp32, oracle/i2r_cpu.py on torch %s CPU, %d threads "
                      "(%d cores visible)" % (n, persons, torch.__version__, threads, cores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="w48_pure_en6", help="workload config (default = BASELINE configs[1]); others are "
                    "exploratory: tph_192_p6_b4, hrt_192_p4_b4, coco_hrt_288_p2_b4")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16", "fp16"],
                    help="MFMA operand type of the conv kernels (default fp32 = the BASELINE configs[1] parity mode)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        raise SystemExit("--gpus %d needs one process per GPU: launch with python -m torch.distributed.run "
                         "--nproc-per-node %d ... bench.py --gpus %d" % (args.gpus, args.gpus, args.gpus))
    assert world == args.gpus, "WORLD_SIZE=%d but --gpus %d" % (world, args.gpus)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)  # RCCL

    cfg = config.load_config(args.config)
    if args.config != "w48_pure_en6":
        args.no_cpu_baseline = True
    sd = synth.make_state_dict(arch.param_spec(cfg))
    net = eval("models." + cfg.MODEL.NAME + ".get_pose_net")(cfg, is_train=False)
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).set_precision(args.precision)

    # global workload: world * 8 images of 4 persons; this rank's contiguous shard
    length_all = [PERSONS] * (IMAGES_PER_GPU * world)
    lo, hi, off = i2r_dist.shard_images(length_all, rank, world)
    length = length_all[lo:hi]
    counts = [sum(length_all[a:b]) for a, b in
              [i2r_dist.shard_images(length_all, r, world)[:2] for r in range(world)]]
    W_, H_ = cfg.MODEL.IMAGE_SIZE
    x, m, _ = synth.make_inputs(length, H_, W_, seed=rank)
    x, m = x.to(dev), m.to(dev)

    pending = [None]

    def step():
        """one forward over this rank's images; N > 1: the all-gather of step k is waited for after step k+1 has been issued"""
        y = net(x, m, length)
        if isinstance(y, dict):
            y = y["multi"]
        if world > 1:
            h = i2r_dist.gather_heatmaps_async(y, counts)
            if pending[0] is not None:
                y = pending[0].wait()
            pending[0] = h
        return y

    def drain(y):
        if pending[0] is not None:
            y = pending[0].wait()
            pending[0] = None
        return y

    for _ in range(args.warmup):
        y = step()
    drain(y)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        y = step()
    y = drain(y)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    assert torch.isfinite(y).all()

    crops_per_step = sum(length_all)
    value = crops_per_step * args.steps / dt
    out = {
        "metric": "images/sec (256x192 crops) I2R-Net HRNet-W48 inference", "value": round(value, 2), "unit": "images/sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": {"fp32": "f32", "bf16": "bf16", "fp16": "f16"}[args.precision], "data": "synthetic",
        "config": {"workload": "vanilla I2R-Net HRNet-W48-S 256x192, 6 encoder layers, fp32, random weights "
                               "(BASELINE configs[1]: w48_pure_en6)" if (args.config == "w48_pure_en6" and args.precision == "fp32")
                               else "%s (exploratory, %s MFMA operands)" % (args.config, args.precision),
                   "images_per_gpu": IMAGES_PER_GPU, "persons_per_image": PERSONS, "crops_per_gpu_step": sum(length),
                   "parallelism": "dp%d (images sharded, RCCL all-gather of heatmaps)" % world if world > 1 else "single GPU",
                   "gflop_per_crop": round(GFLOP_PER_CROP, 3)},
        "model_tflops": round(value * GFLOP_PER_CROP / 1e3, 2) if args.config == "w48_pure_en6" else None,
    }
    if rank == 0:
        if not args.no_roofline:
            prog = next(iter(net.engine().programs.values()))[0]
            stats, reps = per_launch_timing(prog)
            total_ms = sum(s[1] for s in stats.values())
            dom = max((k for k in stats if k.startswith("conv_")), key=lambda k: stats[k][1])
            cnt, ms, flop = stats[dom]
            ach = flop / (ms * 1e-3) / 1e12
            conv_ms = sum(s[1] for k, s in stats.items() if k.startswith("conv_"))
            conv_flop = sum(s[2] for k, s in stats.items() if k.startswith("conv_"))
            out["roofline"] = {
                "bound": "mfma", "kernel": dom, "achieved": round(ach, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": hbm_traffic(),
                "launches_per_step": cnt // reps, "avg_launch_us": round(ms / cnt * 1e3, 2),
                "gflop_per_launch": round(flop / cnt / 1e9, 4),
                "share_of_step_kernel_time": round(ms / total_ms, 3),
                "all_conv_tflops": round(conv_flop / (conv_ms * 1e-3) / 1e12, 2),
                "per_kernel_ms_per_step": {k: round(s[1] / reps, 3) for k, s in sorted(stats.items(), key=lambda kv: -kv[1][1])},
            }
            if args.config == "w48_pure_en6":
                # attention blocks (north_star): QKV/out projections + QK^T/AV + FFN of the 6 encoder layers, algorithmic FLOPs
                d_, dff_, tok = 96, 192, 192
                per_tok = 2 * d_ * d_ * 4 + 2 * 2 * d_ * dff_ + 2 * 2 * d_ * (PERSONS * tok)
                att_flop = per_tok * sum(length) * tok * cfg.MODEL.ENCODER_LAYERS
                att_ms = sum(s[1] for k, s in stats.items() if k.startswith("enc_")) / reps
                att = att_flop / (att_ms * 1e-3) / 1e12
                out["roofline"]["attention_blocks"] = {"kernels": "enc_kv_k + enc_layer4_k", "gflop_per_step": round(att_flop / 1e9, 3),
                                                       "ms_per_step": round(att_ms, 3), "achieved": round(att, 2),
                                                       "peak": FP32_MFMA_PEAK_TFLOPS, "frac": round(att / FP32_MFMA_PEAK_TFLOPS, 4)}
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, sd)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
