#!/usr/bin/env python
"""bench.py -- throughput of the I2R-Net inference hot path on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config NAME] [--precision P] [--pipeline]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" is ONE forward over one synthetic batch (inputs resident in HBM before the timed region, seeded synthetic weights).
Default workload = BASELINE.json configs[1]: vanilla I2R-Net (HRNet-W48-S, 256x192, 6 encoder layers), fp32, 8 images x 4 persons
= 32 crops per GPU.  --config selects the other BASELINE workloads with THEIR batch shapes and dtypes (WORKLOADS below):
configs[2] tph_192_p6_b4 bf16 (16 ragged images, 1-6 persons), configs[3] hrt_192_p4_b4 bf16 (4 images x 4), configs[4]
coco_hrt_288_p2_b4 fp16 (one image of 12 persons at 384x288).  With N > 1 every rank runs its own batch of that shape (weak scaling,
images are the independent unit) and the per-crop results are all-gathered over RCCL each step.

One JSON line is printed by rank 0:
  value        crops/s of the whole job (all ranks), from the max-over-ranks wall time of exactly K steps
  roofline     the dominant kernel (implicit-GEMM conv on the matrix pipe): algorithmic FLOPs per launch / average launch duration,
               both from a per-launch HIP-event timing pass inside this script, vs the dense MFMA peak of the operand type
               (MI355X_MICROARCH.md: fp32 157.3 TFLOP/s, bf16/fp16 2500 TFLOP/s); attention_blocks = the encoder kernels
  parity       max-abs difference of the first image of the timed batch against the CPU oracle (fp32: the 1e-3 bar of BASELINE.json)
  cpu_baseline the CPU oracle (oracle/i2r_cpu.py, a port of the reference forward) timed on this host, rank 0, N = 1 only
--pipeline times the validate() step around the forward as one unit: uint8 image -> affine crops + bbox masks -> flip-test forward
-> key-point decode, all on the device (lib/core/function.py:124-200); see pipeline_step().
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

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import i2r_amd  # noqa: E402,F401
from i2r_amd import arch, cabi, config, models, synth  # noqa: E402
from i2r_amd import dist as i2r_dist  # noqa: E402

MFMA_PEAK_TFLOPS = {"fp32": 157.3, "bf16": 2500.0, "fp16": 2500.0}  # MI355X_MICROARCH.md (dense)
DTYPE_NAME = {"fp32": "f32", "bf16": "bf16", "fp16": "f16"}


def _config3_length():
    """BASELINE config 3: 16 images, length_i = rng(seed 0).integers(1, 7)"""
    return [int(v) for v in np.random.default_rng(0).integers(1, 7, size=16)]


# name -> per-GPU batch (persons per image), BASELINE dtype, algorithmic GFLOP per crop as a function of persons per image
WORKLOADS = {
    "w48_pure_en6": dict(length=[4] * 8, precision="fp32", gflop=lambda n: 19.085 + 0.0849 * n,
                         label="vanilla I2R-Net HRNet-W48-S 256x192, 6 encoder layers, fp32, random weights (BASELINE configs[1]: w48_pure_en6)"),
    "tph_192_p6_b4": dict(length=_config3_length(), precision="bf16", gflop=lambda n: 34.956 + 0.0283 * n,
                          label="I2R-Net TransPose-H first stage 256x192 p6_b4, 16 CrowdPose-shaped images of 1-6 persons (BASELINE configs[2])"),
    "hrt_192_p4_b4": dict(length=[4] * 4, precision="bf16", gflop=lambda n: 28.048 + 0.023 * n,
                          label="I2R-Net HRFormer-B 256x192 p4_b4, per-GPU batch 16 = 4 images x 4 persons (BASELINE configs[3])"),
    "coco_hrt_288_p2_b4": dict(length=[12], precision="fp16", gflop=lambda n: 61.439 + 0.1165 * n,
                               label="I2R-Net HRFormer-B 384x288, one image of 12 persons (BASELINE configs[4])"),
}


def refuse_tuning_env():
    """A bench number must not depend on a tuning switch: no I2R_* variable may be set (the product library ignores them anyway --
    the kernel-side hooks only exist in a -DI2R_TUNING build -- but the Python side has two: I2R_CONV_CHAIN, I2R_BRANCH_LANES)."""
    bad = sorted(k for k in os.environ if k.startswith("I2R_") and k != "I2R_REFERENCE_ROOT")
    if bad:
        raise SystemExit("bench.py refuses to run with tuning variables set: %s" % ", ".join(bad))


def conv_kernel_name(members):
    """rocprof-visible instantiation name conv_igemm_f32<MT, NT, CAP, PF>, asked from the library itself."""
    import ctypes as C
    arr = (C.POINTER(cabi.ConvDesc) * len(members))(*[C.pointer(m) for m in members])
    buf = C.create_string_buffer(96)
    cabi.check(cabi.lib().i2r_conv_kernel_name(arr, len(members), buf, 96), "i2r_conv_kernel_name")
    return buf.value.decode()


def conv_flop(d):
    return 2.0 * d.n_img * d.conv_h * d.conv_w * d.cout * d.cin * d.ntaps


def conv_bytes(d):
    """ALGORITHMIC HBM bytes of one conv launch member (SURVEY 8d convention: every operand element once): input map(s), packed
    weights, output, and each residual map that is present, in their storage types (fp32, or 16 bit in the bf16 / fp16 modes)."""
    e_in, e_out = (2 if d.in_f16 else 4), (2 if d.out_f16 else 4)
    b = d.n_img * d.in_h * d.in_w * d.cin * e_in * (2 if d.in2 else 1)
    b += d.ntaps * d.cin * d.cout * (4 if d.dtype == 0 else 2)
    o = d.n_img * d.conv_h * d.conv_w * d.rep * d.rep * d.cout * e_out
    return float(b + o * (1 + sum(1 for r in (d.res1, d.res2, d.res_post) if r)))


def _op_bytes(kind, st):
    if kind == cabi.OP_CONV:
        return conv_bytes(st)
    if kind == cabi.OP_CONV_GROUP:
        return sum(conv_bytes(st.d[i].contents) for i in range(st.n))
    return 0.0


HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def _op_name_flop(kind, st):
    if kind == cabi.OP_CONV:
        return conv_kernel_name([st]), conv_flop(st)
    if kind == cabi.OP_CONV_GROUP:
        members = [st.d[i].contents for i in range(st.n)]
        return conv_kernel_name(members), sum(conv_flop(m) for m in members)
    if kind == cabi.OP_CONV_CHAIN:
        members = [st.descs[i].contents for i in range(st.n_layers * st.n_members)]
        return "conv_chain_f32<%d, %d, %d, %d>" % (st.mt, st.nt, st.cap, st.pf), sum(conv_flop(m) for m in members)
    if kind in (cabi.OP_ENC_KV, cabi.OP_ENC_LAYER):
        lp = st.dtype != 0
        return {cabi.OP_ENC_KV: "enc_kv_lp_k" if lp else "enc_kv_k", cabi.OP_ENC_LAYER: "enc_layer_lp_k" if lp else "enc_layer4_k"}[kind], 0.0
    name = {cabi.OP_STEM: "stem_mfma_k", cabi.OP_MAXPOOL: "maxpool_k", cabi.OP_HEAD: "head_mfma_k", cabi.OP_LAYERNORM: "layernorm_k",
            cabi.OP_WINATTN: "window_attn_k", cabi.OP_DWCONV: "dwconv3x3_k", cabi.OP_UPSAMPLE: "upsample_add_k",
            cabi.OP_PE_RES_STEM: "pe_res_stem_k", cabi.OP_HRT_ATTN: "hrt_attn_block_k", cabi.OP_HRT_MLP: "hrt_mlp_block_k", cabi.OP_FUSE_UP: "fuse_up_add_k"}.get(kind, "op%d" % kind)
    return name, 0.0


def per_launch_timing(program, reps=3):
    """Replay the program with HIP events on the launch stream between RUNS of consecutive launches of the same kernel (e.g. the six
    encoder layers, the 8 convs of a branch block) -> per-kernel (launch count, total ms, total flop).  Timing a run as a whole
    keeps the kernels back to back as in the real step; one event pair per launch would add its own few microseconds to each."""
    import ctypes as C
    L = cabi.lib()
    cur = torch.cuda.current_stream().cuda_stream
    streams = (C.c_void_p * 4)(cur, cur, cur, cur)
    ops = [(i, kind, st) for i, (kind, lane, st) in enumerate(program.ops) if kind not in cabi.SYNC_OPS]
    named = [(i,) + _op_name_flop(kind, st) + (_op_bytes(kind, st),) for i, kind, st in ops]  # single-stream pass: lanes collapse onto the current stream
    runs = []  # [name, [op indices], flop, bytes]
    for i, name, flop, nbytes in named:
        if runs and runs[-1][0] == name:
            runs[-1][1].append(i)
            runs[-1][2] += flop
            runs[-1][3] += nbytes
        else:
            runs.append([name, [i], flop, nbytes])
    stats = {}
    for rep in range(reps + 1):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(runs) + 1)]
        evs[0].record()
        for r, (name, idx, flop, nbytes) in enumerate(runs):
            for i in idx:
                cabi.check(L.i2r_run_program(C.cast(C.byref(program._c_ops, i * C.sizeof(cabi.Op)), C.POINTER(cabi.Op)), 1,
                                             streams, None), "op %d" % i)
            evs[r + 1].record()
        torch.cuda.synchronize()
        if rep == 0:
            continue  # warm-up pass
        for r, (name, idx, flop, nbytes) in enumerate(runs):
            s = stats.setdefault(name, [0, 0.0, 0.0, 0.0])
            s[0] += len(idx)
            s[1] += evs[r].elapsed_time(evs[r + 1])
            s[2] += flop
            s[3] += nbytes
    return stats, reps


def stack_timing(program, prefix="enc_", reps=3):
    """ms per replay of the ops whose kernel name starts with `prefix`, timed as WHOLE contiguous ranges (one event pair around each
    maximal run of such ops, e.g. enc_kv_k + the six enc_layer4_k launches of an encoder stack): an event pair costs a few
    microseconds of idle queue, which per_launch_timing's pair per KERNEL run would charge twice to a seven-launch stack whose
    first kernel takes 10 us.  Same launches, same stream, same order as in the timed step."""
    import ctypes as C
    L = cabi.lib()
    cur = torch.cuda.current_stream().cuda_stream
    streams = (C.c_void_p * 4)(cur, cur, cur, cur)
    ops = [(i, kind, st) for i, (kind, lane, st) in enumerate(program.ops) if kind not in cabi.SYNC_OPS]
    ranges = []
    for i, kind, st in ops:
        name = _op_name_flop(kind, st)[0]
        if not name.startswith(prefix):
            continue
        if ranges and ranges[-1][-1] == i - 1:
            ranges[-1].append(i)
        else:
            ranges.append([i])
    if not ranges:
        return 0.0
    total = 0.0
    for rep in range(reps + 1):
        pairs = []
        for idx in ranges:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in idx:
                cabi.check(L.i2r_run_program(C.cast(C.byref(program._c_ops, i * C.sizeof(cabi.Op)), C.POINTER(cabi.Op)), 1,
                                             streams, None), "op %d" % i)
            e1.record()
            pairs.append((e0, e1))
        torch.cuda.synchronize()
        if rep:
            total += sum(a.elapsed_time(b) for a, b in pairs)
    return total / reps


def attention_flop(program):
    """algorithmic FLOPs of the encoder stacks of a program (north_star 'attention blocks'): per layer and token the q/k/v/out
    projections (8 d^2), the FFN (4 d dff) and QK^T + AV over the token's own group (4 d L_g)"""
    total = 0.0
    for st in program.enc_stacks:
        offs = st["current"]
        lens = [offs[i + 1] - offs[i] for i in range(len(offs) - 1)]
        for d, _ in st["descs"]:
            dm, dff = d.d, 192
            total += sum(L * (8.0 * dm * dm + 4.0 * dm * dff + 4.0 * dm * L) for L in lens)
    return total


def hbm_traffic(cname):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
    runs, FETCH_SIZE doubled per MI355X_MICROARCH.md).  bench.py itself cannot collect PMC counters: this is a constant read from
    profiles/ (named in traffic_source), null when no profile of this round exists for the workload."""
    for rnd in ("round2", "round1"):
        path = os.path.join(ROOT, "profiles", "%s_hbm_traffic%s.json" % (rnd, "" if cname == "w48_pure_en6" else "_" + cname))
        try:
            with open(path) as f:
                return round(json.load(f)["hbm_bytes_per_launch"]), os.path.relpath(path, ROOT)
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def oracle_parity(cfg, sd, x, m, length, y, precision):
    """first image of the timed batch through the CPU oracle (the checker, never the thing measured)"""
    import i2r_cpu
    n = length[0]
    torch.set_num_threads(min(len(os.sched_getaffinity(0)), 32))
    ref = i2r_cpu.forward(sd, cfg, x[:n].cpu(), m[:n].cpu(), [n])
    ref = ref["multi"] if isinstance(ref, dict) else ref
    # the image must be re-run alone: its crops depend on its own image only, so the batch rows are the same numbers
    diff = (y[:n].cpu() - ref).abs().max().item()
    out = {"max_abs": float("%.3e" % diff), "vs": "oracle/i2r_cpu.py fp32 on image 0 of the timed batch (%d crops)" % n,
           "ref_max_abs": round(ref.abs().max().item(), 3)}
    if precision == "fp32":
        out["tolerance"] = 1e-3
        out["ok"] = diff < 1e-3
    else:
        out["rel_max"] = float("%.3e" % (diff / ref.abs().max().item()))
        out["tolerance"] = "tests/test_model_gpu.py LP_TOL (%s): max-abs <= %s of max|ref|" % (precision, {"bf16": "5 %", "fp16": "1 %"}[precision])
        out["ok"] = out["rel_max"] <= {"bf16": 5e-2, "fp16": 1e-2}[precision]
    return out


def cpu_baseline(cfg, sd, H, W, length, budget_s=20.0):
    """The CPU oracle (a port of the reference forward) timed on this host on the SAME batch shape as the GPU line (`length`: persons per
    image) when that fits the time budget, else on its largest image alone; bounded to ~budget_s seconds."""
    import i2r_cpu
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    threads = min(cores, 32)  # torch CPU convs stop scaling (and oversubscribe badly) far below 256 threads
    torch.set_num_threads(threads)
    x, m, l1 = synth.make_inputs([1], H, W)
    t0 = time.perf_counter()
    i2r_cpu.forward(sd, cfg, x, m, l1)  # warm-up + cost probe on ONE crop
    probe = time.perf_counter() - t0
    if probe * sum(length) * 3 < budget_s:
        sample, what = list(length), "the timed batch shape (%d images, %d crops)" % (len(length), sum(length))
    elif probe * max(length) * 3 < budget_s:
        sample, what = [max(length)], "1 image x %d person(s)" % max(length)
    else:
        sample, what = [1], "1 image x 1 person"
    x, m, ls = synth.make_inputs(sample, H, W)
    n, t0 = 0, time.perf_counter()
    while True:
        i2r_cpu.forward(sd, cfg, x, m, ls)
        n += 1
        if time.perf_counter() - t0 > budget_s * 0.6 or n >= 40:
            break
    dt = time.perf_counter() - t0
    return {"value": round(n * sum(sample) / dt, 3), "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": "%d forwards of %s at %dx%d, fp32, oracle/i2r_cpu.py on torch %s CPU, %d threads "
                      "(%d cores visible)" % (n, what, H, W, torch.__version__, threads, cores)}


def make_pipeline(net, cfg, length, H, W, dev, seed):
    """The validate() step around the forward as one device-side unit (lib/core/function.py:124-200, JointsDataset.py:296-333):
    uint8 image + person boxes -> affine crops + bbox masks (i2r_crop_affine / i2r_box_mask) -> flip-test forward -> key points
    (i2r_decode).  Synthetic 640x480 images, boxes from the seeded generator; returns step() -> (preds [S,J,2], maxvals [S,J,1])."""
    from i2r_amd import caller, input as i2r_input
    rng = np.random.default_rng(seed)
    ih, iw = 480, 640
    images, boxes = [], []
    for n in length:
        images.append(torch.from_numpy(rng.integers(0, 256, size=(ih, iw, 3), dtype=np.uint8)).to(dev))
        b = np.stack([rng.uniform(20, iw * 0.5, n), rng.uniform(20, ih * 0.5, n), rng.uniform(60, iw * 0.45, n), rng.uniform(90, ih * 0.45, n)], 1)
        boxes.append(b)
    ds = cfg.DATASET.DATASET.lower() if cfg.DATASET.DATASET.lower() in caller.FLIP_PAIRS else ("crowdpose" if cfg.MODEL.NUM_JOINTS == 14 else "coco")
    pairs = caller.FLIP_PAIRS[ds]
    cs = [[i2r_input.box_to_center_scale(b, (W, H)) for b in bs] for bs in boxes]
    centers = np.concatenate([np.stack([c for c, _ in one]) for one in cs])
    scales = np.concatenate([np.stack([s for _, s in one]) for one in cs])

    def step():
        xs, ms = [], []
        for img, bs, one in zip(images, boxes, cs):  # per image, as JointsDataset.__getitem__ does (host: 2x3 affine solve per person)
            x, m = i2r_input.person_inputs(img, [c for c, _ in one], [s for _, s in one], bs, (W, H),
                                           color_rgb=bool(cfg.DATASET.COLOR_RGB), device=dev)
            xs.append(x)
            ms.append(m)
        x, m, lens = i2r_input.collate(list(zip(xs, ms)))
        hm = net.forward_flip(x, m, lens, pairs)
        return caller.decode(hm, centers, scales, cfg.TEST.BLUR_KERNEL)
    return step


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="w48_pure_en6", choices=sorted(WORKLOADS),
                    help="workload (default = BASELINE configs[1]); the others are BASELINE configs[2..4] with their own batch shapes")
    ap.add_argument("--precision", default=None, choices=["fp32", "bf16", "fp16"],
                    help="MFMA operand type (default: the workload's BASELINE dtype -- fp32 / bf16 / bf16 / fp16)")
    ap.add_argument("--pipeline", action="store_true", help="time image -> crops -> flip-test forward -> key points instead of the bare forward")
    ap.add_argument("--gather", default="keypoints", choices=["keypoints", "heatmaps"],
                    help="N > 1: payload of the per-step all-gather (decoded key points [S,J,3], or the heat maps)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    args = ap.parse_args()
    refuse_tuning_env()

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

    wl = WORKLOADS[args.config]
    precision = args.precision or wl["precision"]
    cfg = config.load_config(args.config)
    sd = synth.make_state_dict(arch.param_spec(cfg))
    net = eval("models." + cfg.MODEL.NAME + ".get_pose_net")(cfg, is_train=False)
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).set_precision(precision)

    # global workload: every rank's batch has the workload's shape (weak scaling); this rank's contiguous shard of the image list
    per_gpu = list(wl["length"])
    length_all = per_gpu * world
    bounds = i2r_dist.shard_bounds(length_all, world) if world > 1 else [0, len(length_all)]
    if world > 1:  # identical shapes -> the balanced cuts are the per-GPU batches
        assert [bounds[r + 1] - bounds[r] for r in range(world)] == [len(per_gpu)] * world, bounds
    length = length_all[bounds[rank]:bounds[rank + 1]]
    counts = [sum(length_all[bounds[r]:bounds[r + 1]]) for r in range(world)]
    W_, H_ = cfg.MODEL.IMAGE_SIZE
    x, m, _ = synth.make_inputs(length, H_, W_, seed=rank)
    x, m = x.to(dev), m.to(dev)
    gflop_per_step = sum(n * wl["gflop"](n) for n in length_all)

    from i2r_amd import caller
    pending = [None]
    pipe = make_pipeline(net, cfg, length, H_, W_, dev, seed=rank) if args.pipeline else None

    def step():
        """one forward over this rank's images; N > 1: the all-gather of step k is waited for after step k+1 has been issued"""
        if pipe is not None:
            preds, maxv = pipe()
            y = torch.cat([preds, maxv], 2)
            if world > 1:
                h = i2r_dist.gather_heatmaps_async(y, counts)
        else:
            y = net(x, m, length)
            if isinstance(y, dict):
                y = y["multi"]
            if world > 1:
                if args.gather == "keypoints":  # decode on the device, gather [S, J, 3] (168 B/crop) instead of 172 KB/crop
                    preds, maxv = caller.decode(y, None, None, cfg.TEST.BLUR_KERNEL, transform_back=False)
                    h = i2r_dist.gather_keypoints(preds, maxv, counts, async_op=True)
                else:
                    h = i2r_dist.gather_heatmaps_async(y, counts)
        if world > 1:
            if pending[0] is not None:
                pending[0].wait()
            pending[0] = h
        return y

    def drain():
        if pending[0] is not None:
            pending[0].wait()
            pending[0] = None

    for _ in range(args.warmup):
        y = step()
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        y = step()
    drain()
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
        "metric": "images/sec (%dx%d crops) I2R-Net inference" % (H_, W_) if args.config != "w48_pure_en6"
                  else "images/sec (256x192 crops) I2R-Net HRNet-W48 inference",
        "value": round(value, 2), "unit": "images/sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE_NAME[precision], "data": "synthetic",
        "config": {"workload": wl["label"] + ("" if precision == wl["precision"] else " -- run with %s MFMA operands" % precision)
                               + (" -- PIPELINE: uint8 image -> crops + masks -> flip-test forward -> key points" if args.pipeline else ""),
                   "images_per_gpu": len(length), "persons_per_image": length if len(set(length)) > 1 else length[0],
                   "crops_per_gpu_step": sum(length),
                   "parallelism": "dp%d (images sharded, RCCL all-gather of %s)" % (world, "key points" if (args.pipeline or args.gather == "keypoints") else "heat maps")
                                  if world > 1 else "single GPU",
                   "gflop_per_step_per_gpu": round(gflop_per_step / world, 2)},
        "model_tflops": round(gflop_per_step * (2 if args.pipeline else 1) * args.steps / dt / 1e3, 2),
    }
    if rank == 0:
        eng = net.engine()
        if not args.no_roofline:
            key = next(k for k in eng.programs if k[3] == bool(args.pipeline))
            prog = eng.programs[key][0]
            stats, reps = per_launch_timing(prog)
            total_ms = sum(s[1] for s in stats.values())
            dom = max((k for k in stats if k.startswith("conv_")), key=lambda k: stats[k][1])
            cnt, ms, flop, nbytes = stats[dom]
            ach = flop / (ms * 1e-3) / 1e12
            conv_ms = sum(s[1] for k, s in stats.items() if k.startswith("conv_"))
            conv_flop_ = sum(s[2] for k, s in stats.items() if k.startswith("conv_"))
            peak = MFMA_PEAK_TFLOPS[precision]
            traffic, traffic_src = hbm_traffic(args.config)
            out["roofline"] = {
                "bound": "mfma", "kernel": dom, "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
                "launches_per_step": cnt // reps, "avg_launch_us": round(ms / cnt * 1e3, 2),
                "gflop_per_launch": round(flop / cnt / 1e9, 4),
                "share_of_step_kernel_time": round(ms / total_ms, 3),
                "all_conv_tflops": round(conv_flop_ / (conv_ms * 1e-3) / 1e12, 2),
                "per_kernel_ms_per_step": {k: round(s[1] / reps, 3) for k, s in sorted(stats.items(), key=lambda kv: -kv[1][1])},
            }
            # which roof bounds the dominant kernel: its arithmetic intensity (algorithmic FLOP / algorithmic HBM byte, both per
            # launch) against the machine balance peak FLOP/s : 8 TB/s.  fp32 convs sit far above it (MFMA-bound); with 16-bit
            # operands the matrix peak is 16x higher and the same launches fall BELOW it: their roof is HBM.
            r = out["roofline"]
            gbs = nbytes / (ms * 1e-3) / 1e9
            ai, balance = flop / nbytes, peak * 1e12 / (HBM_PEAK_GBS * 1e9)
            r["gbytes_per_launch"] = round(nbytes / cnt / 1e9, 4)
            r["intensity_flop_per_byte"], r["machine_balance_flop_per_byte"] = round(ai, 1), round(balance, 1)
            if ai < balance:
                r["mfma_view"] = {"achieved": r["achieved"], "peak": peak, "unit": "TFLOP/s", "frac": r["frac"]}
                r.update({"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)})
            else:
                r["hbm_view"] = {"achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)}
            att_flop = attention_flop(prog)
            att_k = sorted(k for k in stats if k.startswith("enc_"))
            att_ms = stack_timing(prog)  # (each encoder stack timed as one unit; the per-kernel split stays in per_kernel_ms_per_step)
            if att_flop and att_ms:
                att = att_flop / (att_ms * 1e-3) / 1e12
                att_peak = MFMA_PEAK_TFLOPS["fp32" if "enc_layer4_k" in att_k and "enc_layer_lp_k" not in att_k else precision]
                out["roofline"]["attention_blocks"] = {"kernels": " + ".join(att_k), "gflop_per_step": round(att_flop / 1e9, 3),
                                                       "ms_per_step": round(att_ms, 3), "achieved": round(att, 2),
                                                       "peak": att_peak, "frac": round(att / att_peak, 4)}
        if not args.no_parity and not args.pipeline:
            y1 = net(x, m, length)
            y1 = y1["multi"] if isinstance(y1, dict) else y1
            out["parity"] = oracle_parity(cfg, sd, x, m, length, y1, precision)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(cfg, sd, H_, W_, length)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
