#!/usr/bin/env python
"""bench.py -- throughput of the I2R-Net inference hot path on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config NAME] [--precision P] [--pipeline] [--ragged-stream] [--scaling strong]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" is ONE forward over one synthetic batch (inputs resident in HBM before the timed region, seeded synthetic weights).
Default workload = BASELINE.json configs[1]: vanilla I2R-Net (HRNet-W48-S, 256x192, 6 encoder layers), fp32, 8 images x 4 persons
= 32 crops per GPU.  --config selects the other BASELINE workloads with THEIR batch shapes and dtypes (WORKLOADS below):
configs[2] tph_192_p6_b4 bf16 (16 ragged images, 1-6 persons), configs[3] hrt_192_p4_b4 bf16 (4 images x 4), configs[4]
coco_hrt_288_p2_b4 fp16 (one image of 12 persons at 384x288).

N > 1: one process per GPU.  Under torch.distributed.run (WORLD_SIZE set) this process is one rank; called plainly as
`python bench.py --gpus N` it launches the N ranks itself (self_launch: re-executes under torch.distributed.run on 127.0.0.1).
Every rank runs its own batch of the workload's shape (weak scaling, images are the independent unit) and the per-crop results are
all-gathered over RCCL each step: the key points decoded on the device, [S, J, 3] (168 B per crop: what validate() keeps of a batch;
--gather heatmaps gathers the predicted heat maps, the payload BASELINE.json's north_star names, 172 KB per crop).  Decode and gather
are issued on a side stream under the next forward (dist.PostStep).  The line's `value` is timed with the --gather payload;
`gather_alt` is the same K steps re-timed with the other payload.

--scaling strong (default weak): the total job is FIXED -- 64 images with 1-6 persons -- and cut into contiguous image shards balanced by
crop count (dist.shard_bounds); the ranks' crop counts differ, the line carries them and the imbalance (`shards`).

One JSON line is printed by rank 0:
  value        crops/s of the whole job (all ranks), from the max-over-ranks wall time of exactly K steps
  roofline     the kernel with the LARGEST share of the step (over all kernels), measured IN SITU: the forward of the timed region is re-run
               with a HIP stop event bound to every dispatch and a start marker in front of it, on the launch's own stream
               (i2r_run_program_timed) -- avg_launch_us = mean kernel duration with lanes / sibling programs in flight, busy_ms_per_step =
               time with at least one launch of the kernel in flight, frac = EXECUTED FLOPs / algorithmic HBM bytes of all its launches
               (op_model; for the Winograd kernel the executed multiply-adds = the direct convolution's / 2.25, so frac <= 1 -- the
               direct-convolution FLOPs it delivers are filed under direct_equivalent) over that busy time against the roof its
               arithmetic intensity selects (MI355X_MICROARCH.md: fp32 MFMA 157.3 TFLOP/s, bf16/fp16 2500 TFLOP/s, HBM 8 TB/s); the
               markers slow the forward (forward_ms_with_timing_events), so frac is a LOWER bound; `standalone` = the same launches alone
               on one stream; `concurrent_programs` = the part-batch programs fork -> join between two events (unperturbed);
               `kernels` = the same figures for the five largest kernels; attention_blocks = the encoder kernels
  lanes        what the engine's stream probe found (candidate streams, alone / together spin times, roles), whether the device-side
               lane synchronisation is in use and that none of its waits timed out
  collective_overhead  (default line) BASELINE configs[3] timed alternately as the plain step and as the data-parallel step (device decode +
               RCCL all-gather through a one-rank group, on a side stream under the next forward) in this process
  other_workloads  (default single-GPU line only) BASELINE configs[2..4] at their own batch shapes and dtypes, the ragged stream and the
               pipeline, 10-30 steps each, measured in this process after the headline's timed region: value, ms_per_step, dominant kernel
               with its roofline fraction, parity against the CPU oracle
  parity       max-abs difference of the first image of the timed batch against the CPU oracle (fp32: the 1e-3 bar of BASELINE.json)
  cpu_baseline the CPU oracle (oracle/i2r_cpu.py, a port of the reference forward) timed on this host, rank 0, N = 1 only
--pipeline times the validate() step around the forward as one unit: uint8 image -> affine crops + bbox masks -> flip-test forward
-> key-point decode, all on the device (lib/core/function.py:124-200); see make_pipeline().
--ragged-stream times what validate() really feeds the model (lib/core/function.py:124-140): a stream of batches whose crop count
changes with every batch (16 images of 1-6 persons); see ragged_stream().
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
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
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E ~8 TB/s
WINO_CUT = 2.25  # Winograd F(2x2, 3x3): 36 / 16 multiply-adds of the direct convolution per executed one
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


# --scaling strong: the fixed job -- 64 images, persons per image = rng(seed 0).integers(1, 7) (the config-3 distribution, 4 x 16 images)
STRONG_LENGTH = [int(v) for v in np.random.default_rng(0).integers(1, 7, size=64)]


def refuse_tuning_env():
    """A bench number must not depend on a tuning switch: no I2R_* variable may be set (the product library ignores them anyway --
    the kernel-side hooks only exist in a -DI2R_TUNING build -- but the Python side has two: I2R_CONV_CHAIN, I2R_BRANCH_LANES)."""
    bad = sorted(k for k in os.environ if k.startswith("I2R_") and k != "I2R_REFERENCE_ROOT")
    if bad:
        raise SystemExit("bench.py refuses to run with tuning variables set: %s" % ", ".join(bad))


# ------------------------------------------------------------------------------------------------------------------------------
# algorithmic model of every launch of a program: (rocprof-visible kernel name, FLOP, HBM bytes, matrix pipe)
# ------------------------------------------------------------------------------------------------------------------------------
def conv_kernel_name(members):
    """rocprof-visible instantiation name conv_igemm_f32<MT, NT, CAP, PF>, asked from the library itself."""
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


def _esz(dt):
    return 2 if dt else 4


def op_model(kind, st, precision, enc_lens=None):
    """(kernel name, algorithmic FLOP, algorithmic HBM bytes, matrix pipe) of one launch.  FLOPs count multiply-adds of the reference's
    dense contractions as 2 (conv 2 pixels cout cin taps; encoder token: 8 d^2 projections + 4 d dff FFN + 4 d L_group attention;
    HRFormer window of 49 tokens: 8 C^2 49 + 4 C 49^2; MLP pixel: fc1 + fc2 16 C^2 + depth-wise 18 x 4C); element-wise kernels get their
    adds/compares.  Bytes: every operand element once in its storage type.  pipe: 'fp32' | 'bf16' | 'fp16' = the MFMA operand type the
    kernel runs its contractions on, None = no matrix work (the roof of such a kernel is HBM)."""
    lp = precision if precision != "fp32" else None
    if kind == cabi.OP_CONV:
        return conv_kernel_name([st]), conv_flop(st), conv_bytes(st), (precision if st.dtype else "fp32")
    if kind == cabi.OP_CONV_GROUP:
        ms = [st.d[i].contents for i in range(st.n)]
        return conv_kernel_name(ms), sum(conv_flop(m) for m in ms), sum(conv_bytes(m) for m in ms), (precision if ms[0].dtype else "fp32")
    if kind == cabi.OP_CONV_CHAIN:
        ms = [st.descs[i].contents for i in range(st.n_layers * st.n_members)]
        return ("conv_chain_f32<%d, %d, %d, %d>" % (st.mt, st.nt, st.cap, st.pf), sum(conv_flop(m) for m in ms),
                sum(conv_bytes(m) for m in ms), "fp32")
    if kind in (cabi.OP_ENC_KV, cabi.OP_ENC_LAYER):
        islp = st.dtype != 0
        d, dff, cs, n = st.d, st.dff_pad, st.cs, st.n_tok
        lens = enc_lens or [n]
        kv_e = 2 if islp else 4
        if kind == cabi.OP_ENC_KV:  # k = (src + pos) Wk + bk, v = src Wv + bv
            return ("enc_kv_lp_k" if islp else "enc_kv_k", 4.0 * d * d * n, float(n * cs * 4 * (2 if st.pos else 1) + 2 * n * cs * kv_e + 2 * d * d * kv_e),
                    lp if islp else "fp32")
        flop = n * (4.0 * d * d + 4.0 * d * dff) + sum(4.0 * d * L * L for L in lens)  # q + out proj, FFN, QK^T + PV over the token's group
        if st.next_w_in:
            flop += 4.0 * d * d * n  # the NEXT layer's k / v projection, fused into this launch
        nbytes = n * cs * 4 * (3 if st.pos else 2) + 2 * n * cs * kv_e * (2 if st.next_w_in else 1) + (4 * d * d + 2 * d * dff) * kv_e
        name = ("enc_layer_lp4_k" if (st.n_qtiles192 > 0 and st.n_qtiles64 >= 512) else "enc_layer_lp_k") if islp else "enc_layer4_k"
        return name, flop, float(nbytes), lp if islp else "fp32"
    if kind == cabi.OP_MH_ATTN:  # QK^T + PV over the token's group as executed (heads of hp = head dim padded to 16)
        lens = enc_lens or []
        n = sum(lens)
        return "enc_mh_attn_k", sum(4.0 * st.heads * st.hp * L * L for L in lens), float(n * (st.qk_cs + st.v_cs + st.out_cs) * 4), "fp32"
    if kind == cabi.OP_HRT_ATTN:
        nwin = st.n_img * ((st.h + 6) // 7) * ((st.w_ + 6) // 7)
        c = st.c
        return ("hrt_attn_head_k" if st.variant == 2 else "hrt_attn_block_k", nwin * (8.0 * c * c * 49 + 4.0 * c * 49 * 49), float(2 * st.n_img * st.h * st.w_ * st.cs * 4 + 4 * c * c * 2), lp)
    if kind == cabi.OP_HRT_MLP:
        npix, c = st.n_img * st.h * st.w_, st.c
        return ("hrt_mlp_wide_k" if st.variant == 2 else "hrt_mlp_block_k", npix * (16.0 * c * c + 18.0 * 4 * c), float(2 * npix * st.cs * 4 + 8 * c * c * 2 + 10 * 4 * c * 4), lp)
    if kind == cabi.OP_WINATTN:
        nwin = st.n_img * ((st.h + 6) // 7) * ((st.w_ + 6) // 7)
        return "window_attn_k", nwin * 4.0 * st.c * 49 * 49, float(st.n_img * st.h * st.w_ * st.cs * 4 * 4), "fp32"  # (cs = hs; q|k|v in, o out)
    if kind == cabi.OP_STEM:
        oh, ow = (st.in_h - 1) // 2 + 1, (st.in_w - 1) // 2 + 1
        return ("stem_mfma_k" if st.cout == 64 else "stem_conv_k", 2.0 * st.n_img * oh * ow * st.cout * 9 * st.cin,
                float(st.n_src * st.cin * st.in_h * st.in_w * 4 + st.n_img * oh * ow * st.out_cs * _esz(st.out_dt)), "fp32")
    if kind == cabi.OP_PE_RES_STEM:
        oh, ow = (st.in_h - 1) // 2 + 1, (st.in_w - 1) // 2 + 1
        return ("pe_res_stem_k", st.n_img * (st.in_h * st.in_w * 54.0 + oh * ow * 2.0 * 147 * st.cout),
                float(st.n_src * st.in_h * st.in_w * 4 + st.n_img * oh * ow * st.out_cs * 4), None)
    if kind == cabi.OP_ROWS_GATHER:
        return "rows_gather_k", 0.0, float(2 * st.n_out * st.floats_per_crop * 4), None
    if kind == cabi.OP_VIEW_SCRAMBLE:
        return "view_scramble_k", 0.0, float(st.n_out * st.hw * st.cs * 4 + st.n_images * st.max_persons * st.hw * st.cs * 4), None
    if kind == cabi.OP_PE_CAT_VEC:  # window maxima of the mask, one th*tw-long dot product per vector element, the broadcast store
        P = st.th * st.tw
        return ("pe_cat_vec_k", float(st.n_img * (st.in_h * st.in_w + 2.0 * P * st.vec)),
                float(st.n_valid * st.in_h * st.in_w * 4 + st.n_img * P * (st.c_end - st.c0) * 4 + st.vec * (P + 1) * 4), None)
    if kind == cabi.OP_HEAD:
        npix = st.n_img * st.h * st.w_
        return ("head_mfma_k" if st.cin <= 128 else "head_k", 2.0 * npix * st.cin * st.cout, float(npix * (st.in_cs + st.cout) * 4), "fp32")
    if kind == cabi.OP_MAXPOOL:
        oh, ow = (st.in_h - 1) // 2 + 1, (st.in_w - 1) // 2 + 1
        return "maxpool_k", 9.0 * st.n_img * oh * ow * st.c, float(st.n_img * (st.in_h * st.in_w * st.in_cs + oh * ow * st.out_cs) * 4), None
    if kind == cabi.OP_LAYERNORM:
        return "layernorm_k", 8.0 * st.npix * st.c, float(st.npix * st.cs * (4 + _esz(st.out_dt))), None
    if kind == cabi.OP_DWCONV:
        oh, ow = (st.in_h - 1) // st.stride + 1, (st.in_w - 1) // st.stride + 1
        return ("dwconv3x3_k", 18.0 * st.n_img * oh * ow * st.c, float(st.n_img * (st.in_h * st.in_w + oh * ow) * st.cs * _esz(st.dt) + 10 * st.cs * 4), None)
    if kind == cabi.OP_UPSAMPLE:
        n_out = st.n_img * st.low_h * st.low_w * st.scale * st.scale
        scales = [st.scale] + ([st.scale2] if st.low2 else []) + ([st.scale3] if st.low2 and st.low3 else [])
        return ("upsample_add_k", 8.0 * n_out * st.c * len(scales), float((sum(n_out // (s_ * s_) for s_ in scales) + 2 * n_out) * st.cs * 4), None)
    if kind == cabi.OP_FUSE_UP:
        n_out = st.n_img * st.h * st.w
        low = n_out // (st.s1 * st.s1) + (n_out // (st.s2 * st.s2) if st.t2 else 0)
        return "fuse_up_add_k", (2.0 if st.t2 else 1.0) * n_out * st.cs, float((2 * n_out + low) * st.cs * _esz(st.dt)), None
    if kind == cabi.OP_CONV1X1_PAIR:
        n = st.n_pix
        return ("conv1x1_pair_k<%d, %d, %d>" % (st.k_a // 16, st.cb_out // 16, st.mt or 2), 2.0 * n * st.ca_out * (st.k_a + st.cb_out),
                float(n * (st.x_cs + st.y_cs * (2 if st.res else 1) + st.cb_out) * 4 + st.ca_out * (st.k_a + st.cb_out) * 4), "fp32")
    if kind == cabi.OP_CONV1X1_LP:
        n = st.n_pix
        return ("conv1x1_lp_k", 2.0 * n * st.cin_pad * st.cout_pad,
                float(n * (st.x_cs * _esz(st.in_16) + st.out_cs * _esz(st.out_16) * (1 + bool(st.res1) + bool(st.res2) + bool(st.res_post))) + st.cin_pad * st.cout_pad * 2), precision)
    return "op%d" % kind, 0.0, 0.0, None


def _enc_lens(program):
    """address of every encoder descriptor of the program -> token-group lengths (the attention term of op_model needs them)"""
    out = {}
    for st in program.enc_stacks:
        offs = st["current"]
        lens = [offs[i + 1] - offs[i] for i in range(len(offs) - 1)]
        for d, _ in st["descs"]:
            out[C.addressof(d)] = lens
        for a in st.get("mh", ()):
            out[C.addressof(a)] = lens
    return out


def _launch_ops(program):
    return [(i, kind, st) for i, (kind, lane, st) in enumerate(program.ops) if kind not in cabi.SYNC_OPS]


def _run_one(L, program, i, streams):
    cabi.check(L.i2r_run_program(C.cast(C.byref(program._c_ops, i * C.sizeof(cabi.Op)), C.POINTER(cabi.Op)), 1, streams, None), "op %d" % i)


def _op_models(program, precision):
    """[(op index, kernel name, FLOP, bytes, pipe)] of the program's launches, cached on the program (the grouping may change between
    forwards: the cache is keyed by the encoder stacks' current offsets)"""
    lens = _enc_lens(program)
    key = (precision, tuple(tuple(v) for v in lens.values()))
    cached = getattr(program, "_bench_models", None)
    if cached is None or cached[0] != key:
        rows = [(i,) + op_model(kind, st, precision, lens.get(C.addressof(st)) if kind in (cabi.OP_ENC_KV, cabi.OP_ENC_LAYER, cabi.OP_MH_ATTN) else None)
                for i, kind, st in _launch_ops(program)]
        program._bench_models = cached = (key, rows)
    return cached[1]


def _union_ms(iv):
    """total length of the union of [a, b] intervals"""
    tot, end = 0.0, None
    for a, b in sorted(iv):
        if end is None or a > end:
            tot += b - a
            end = b
        elif b > end:
            tot += b - end
            end = b
    return tot


def in_situ_timing(fwd, precision, reps=5):
    """Per-launch durations INSIDE the real forward.  `fwd()` is the product's forward (the callable the timed region ran); while
    engine.Program.timing_log is armed every Program.run() goes through i2r_run_program_timed: every launch goes out with a STOP event bound
    to its dispatch and a START marker in front of it, on the launch's OWN stream -- lanes, part-batch sibling programs, fork / join /
    xsync exactly as in the timed step.  elapsed(start, stop) = the kernel's own duration in situ (tools/probe/event_timing.hip: 20.9 /
    50.9 us for kernels of 20 / 50 us).  The marker costs its stream ~5 us per launch: the timed forward is 10-20 % slower than the real
    one (forward_ms_with_timing_events beside ms_per_step), so a kernel's busy time is an UPPER bound and `frac` a LOWER bound of what the
    real step reaches.  A rocprofv3 kernel trace of the command agrees where nothing overlaps (the tail's encoder layers, layer1) and
    cannot be compared where streams do: its interception makes the host the bottleneck and the part-batch programs run one after the
    other (tools/summarize_round6.py puts in-situ, standalone and rocprofv3 figures side by side).
    -> stats {kernel name: dict(cnt, dur_ms = sum of in-situ durations, union_ms = time with at least one launch of the kernel in
    flight, flop, bytes, pipe)} summed over `reps` forwards (one more, untimed, comes first), and the mean wall ms of a timed forward."""
    from i2r_amd import engine
    stats, wall = {}, []
    for rep in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        engine.Program.timing_log = []
        try:
            e0.record()
            fwd()
            e1.record()
            torch.cuda.synchronize()
            log = engine.Program.timing_log
        finally:
            engine.Program.timing_log = None
        if rep == 0:
            continue
        wall.append(e0.elapsed_time(e1))
        iv = {}
        for P, t0, t1, lane_streams in log:
            models = {i: (name, flop, nbytes, pipe) for i, name, flop, nbytes, pipe in _op_models(P, precision)}
            ready = {}  # stream handle -> time (ms after e0) at which everything this stream has to wait for is complete
            slot_t = {}  # record slot -> `ready` of the recording stream at the record

            def mx(*v):
                v = [x for x in v if x is not None]
                return max(v) if v else None
            for i, (kind, lane, st) in enumerate(P.ops):
                if kind == cabi.OP_LANE_FLAGS:
                    continue
                if kind in (cabi.OP_RECORD, cabi.OP_WAIT):
                    sk, slot = lane_streams[lane & 3], (lane >> 8) & 7
                    if kind == cabi.OP_RECORD:
                        slot_t[slot] = ready.get(sk)
                    else:
                        ready[sk] = mx(ready.get(sk), slot_t.get(slot))
                    continue
                if kind in cabi.SYNC_OPS:  # what the sync op makes each stream wait for (csrc/i2r_api.hip: run_program)
                    ls = [lane_streams[l] for l in range(4) if lane & (1 << l)]
                    s0 = lane_streams[0]
                    if kind == cabi.OP_FORK:
                        for l in ls:
                            ready[l] = mx(ready.get(l), ready.get(s0))
                    elif kind == cabi.OP_JOIN:
                        ready[s0] = mx(ready.get(s0), *[ready.get(l) for l in ls])
                    else:
                        m = mx(*[ready.get(l) for l in ls])
                        for l in ls:
                            ready[l] = m
                    continue
                sk = lane_streams[lane]
                if t1[i] is None:  # (not timed in this pass)
                    continue
                end = e0.elapsed_time(t1[i])
                start = e0.elapsed_time(t0[i]) if t0[i] is not None else ready.get(sk)
                if start is None or start > end:
                    start = end
                ready[sk] = end
                name, flop, nbytes, pipe = models[i]
                s = stats.setdefault(name, dict(cnt=0, dur_ms=0.0, union_ms=0.0, flop=0.0, bytes=0.0, pipe=pipe))
                s["cnt"] += 1
                s["dur_ms"] += end - start
                s["flop"] += flop
                s["bytes"] += nbytes
                iv.setdefault(name, []).append((start, end))
        for name, lst in iv.items():
            stats[name]["union_ms"] += _union_ms(lst)
    return stats, reps, sum(wall) / len(wall)


def standalone_timing(programs, precision, reps=3):
    """the OLD view, kept beside the in-situ one: every run of equal launches replayed alone on ONE stream (lanes collapsed, no sibling
    program) between two events -> {kernel: [launches, ms, flop, bytes]} over `reps` replays: what a kernel takes with the chip to itself"""
    stats = {}
    for program in (programs if isinstance(programs, (list, tuple)) else [programs]):
        for k, v in _per_launch_timing_one(program, precision, reps)[0].items():
            s = stats.setdefault(k, [0, 0.0, 0.0, 0.0])
            for i in range(4):
                s[i] += v[i]
    return stats


def _named_runs(program, precision):
    runs = []  # [name, [op indices], flop, bytes, pipe]
    for i, name, flop, nbytes, pipe in _op_models(program, precision):
        if runs and runs[-1][0] == name:
            runs[-1][1].append(i)
            runs[-1][2] += flop
            runs[-1][3] += nbytes
        else:
            runs.append([name, [i], flop, nbytes, pipe])
    return runs


def _per_launch_timing_one(program, precision, reps=3):
    """Replay the program with HIP events on the launch stream between RUNS of consecutive launches of the same kernel (e.g. the six
    encoder layers, the 8 convs of a branch block) -> per-kernel [launch count, total ms, total flop, total bytes, pipe].  Timing a run
    as a whole keeps the kernels back to back as in the real step; one event pair per launch would add its own few microseconds to
    each.  Single-stream pass: stream lanes collapse onto the current stream (kernels of different lanes do not overlap here)."""
    L = cabi.lib()
    cur = torch.cuda.current_stream().cuda_stream
    streams = (C.c_void_p * 4)(cur, cur, cur, cur)
    runs = _named_runs(program, precision)
    stats = {}
    for rep in range(reps + 1):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(runs) + 1)]
        evs[0].record()
        for r, (name, idx, flop, nbytes, pipe) in enumerate(runs):
            for i in idx:
                _run_one(L, program, i, streams)
            evs[r + 1].record()
        torch.cuda.synchronize()
        if rep == 0:
            continue  # warm-up pass
        for r, (name, idx, flop, nbytes, pipe) in enumerate(runs):
            s = stats.setdefault(name, [0, 0.0, 0.0, 0.0, pipe, 0])
            s[0] += len(idx)
            s[1] += evs[r].elapsed_time(evs[r + 1])
            s[2] += flop
            s[3] += nbytes
            s[5] += len(idx)
    return stats, reps


def stack_timing(programs, precision, prefix="enc_", reps=3):
    return sum(_stack_timing_one(P, precision, prefix, reps) for P in (programs if isinstance(programs, (list, tuple)) else [programs]))


def _stack_timing_one(program, precision, prefix="enc_", reps=3):
    """ms per replay of the ops whose kernel name starts with `prefix`, timed as WHOLE contiguous ranges (one event pair around each
    maximal run of such ops, e.g. enc_kv_k + the six enc_layer4_k launches of an encoder stack): an event pair costs a few
    microseconds of idle queue, which per_launch_timing's pair per KERNEL run would charge twice to a seven-launch stack whose
    first kernel takes 10 us.  Same launches, same stream, same order as in the timed step."""
    L = cabi.lib()
    cur = torch.cuda.current_stream().cuda_stream
    streams = (C.c_void_p * 4)(cur, cur, cur, cur)
    ranges = []
    for i, kind, st in _launch_ops(program):
        if not op_model(kind, st, precision)[0].startswith(prefix):
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
                _run_one(L, program, i, streams)
            e1.record()
            pairs.append((e0, e1))
        torch.cuda.synchronize()
        if rep:
            total += sum(a.elapsed_time(b) for a, b in pairs)
    return total / reps


def _kbase(name):
    """kernel name without template arguments / operand-type suffix: conv_igemm_lp<3, 3, 8, 1>/bf16 -> conv_igemm_lp"""
    return name.split("<")[0].split("/")[0]


def hbm_traffic(cname, kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of the
    workload's own bench command, FETCH_SIZE doubled per MI355X_MICROARCH.md).  bench.py itself cannot collect PMC counters: this is a
    constant read from profiles/ (named in traffic_source).  Round-5 files hold every kernel of the workload by BASE name, the bytes
    being the launch-weighted mean over all instantiations that ran (what `algorithmic_bytes` -- the launch-weighted mean of bench's
    own byte model over the same launches -- compares with); older files hold one instantiation.  -> (bytes, source, what) or Nones."""
    base = _kbase(kernel)
    for rnd in ("round6", "round5", "round4", "round3", "round2", "round1"):
        path = os.path.join(ROOT, "profiles", "%s_hbm_traffic%s.json" % (rnd, "" if cname == "w48_pure_en6" else "_" + cname))
        try:
            with open(path) as f:
                j = json.load(f)
            if "by_kernel" in j:
                e = j["by_kernel"].get(base)
                if e is None:
                    continue
                return round(e["hbm_bytes_per_launch"]), os.path.relpath(path, ROOT), "launch-weighted mean over %d launches of %d instantiation(s) of %s" % (
                    e["launches_averaged"], len(e["instantiations"]), base)
            if _kbase(j.get("kernel", "")) == base:
                return round(j["hbm_bytes_per_launch"]), os.path.relpath(path, ROOT), "one instantiation: " + j["kernel"]
        except (OSError, KeyError, ValueError):
            continue
    return None, None, None


def _kernel_view(name, s, reps, total_ms, precision, alone=None):
    """roofline figures of one kernel from its in_situ_timing entry.  `achieved` / `frac` are what the kernel EXECUTES on the roof that
    bounds it, so frac <= 1 always: (FLOPs or algorithmic bytes of ALL its launches) / (the time during which at least one of its launches
    is in flight, `busy_ms_per_step`) -- with part-batch programs or lanes side by side, launches that overlap share that time, which is
    the chip-level rate of the kernel; `avg_launch_us` is the plain mean of the per-launch in-situ durations (what rocprofv3 shows per
    launch) and `launches_in_flight` their ratio.  For the Winograd kernel (csrc/i2r_conv_wino.hip) the executed FLOPs are the
    algorithmic ones / 2.25 (16 multiply-adds per 2x2 output tile and (cin, cout) pair where the direct convolution of SURVEY 8d needs 36;
    checked against PMC SQ_INSTS_MFMA x 2048 FLOP in profiles/round3_pmc_sq_grouped_conv.json); the direct-convolution FLOPs it DELIVERS per
    second are reported separately as `direct_equivalent` and are not a pipe fraction.  `alone` = the kernel's standalone_timing entry."""
    cnt, dur, ms, flop, nbytes, pipe = s["cnt"], s["dur_ms"], s["union_ms"], s["flop"], s["bytes"], s["pipe"]
    wino = name.startswith("conv_wino")
    flop_exec = flop / WINO_CUT if wino else flop
    tf = flop_exec / (ms * 1e-3) / 1e12
    gbs = nbytes / (ms * 1e-3) / 1e9
    out = {"kernel": name,  # (16-bit conv instantiations carry their operand type in the name: conv_igemm_lp<..>/bf16)
           "launches_per_step": cnt // reps, "avg_launch_us": round(dur / cnt * 1e3, 2), "launches_in_flight": round(dur / ms, 2),
           "busy_ms_per_step": round(ms / reps, 3), "share_of_step_kernel_time": round(dur / total_ms, 3),
           "gflop_per_launch": round(flop_exec / cnt / 1e9, 4), "gbytes_per_launch": round(nbytes / cnt / 1e9, 4)}
    peak = MFMA_PEAK_TFLOPS[pipe] if pipe else None
    ai = flop_exec / nbytes if nbytes else 0.0
    balance = peak * 1e12 / (HBM_PEAK_GBS * 1e9) if peak else float("inf")
    mfma = {"achieved": round(tf, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 4)} if peak else None
    hbm = {"achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)}
    out["intensity_flop_per_byte"] = round(ai, 1)
    if peak:
        out["machine_balance_flop_per_byte"] = round(balance, 1)
    if wino:
        out["algorithm"] = ("winograd F(2x2,3x3): gflop_per_launch / achieved / frac count the multiply-adds the matrix pipe executes "
                            "(= direct-convolution FLOPs / 2.25)")
        out["direct_equivalent"] = {"gflop_per_launch": round(flop / cnt / 1e9, 4), "tflops_delivered": round(tf * WINO_CUT, 2),
                                    "note": "SURVEY 8d direct-convolution FLOPs per second; a speed-up over a direct kernel, not a fraction of the pipe"}
    # which roof bounds the kernel: its arithmetic intensity (executed FLOP / algorithmic HBM byte) against the machine balance
    # peak FLOP/s : 8 TB/s of the pipe it computes on.  fp32 convs sit far above it (MFMA-bound); with 16-bit operands the matrix
    # peak is 16x higher and the same launches can fall BELOW it: their roof is HBM.  Kernels without matrix work: HBM.
    mfma_bound = bool(peak and ai >= balance)
    if mfma_bound:
        out.update(bound="mfma", **mfma)
        out["hbm_view"] = hbm
    else:
        out.update(bound="hbm", **hbm)
        if mfma:
            out["mfma_view"] = mfma
    # per-launch fraction: ONE launch's work over its own in-situ duration (a launch that shares the chip with siblings scores low here)
    per = (flop_exec / cnt / (dur / cnt * 1e-3) / 1e12 / peak) if mfma_bound else (nbytes / cnt / (dur / cnt * 1e-3) / 1e9 / HBM_PEAK_GBS)
    out["frac_per_launch_in_situ"] = round(per, 4)
    if alone is not None and alone[0]:
        a_us = alone[1] / alone[0] * 1e3
        a_flop = alone[2] / (WINO_CUT if wino else 1.0)
        a_frac = (a_flop / (alone[1] * 1e-3) / 1e12 / peak) if mfma_bound else (alone[3] / (alone[1] * 1e-3) / 1e9 / HBM_PEAK_GBS)
        out["standalone"] = {"avg_launch_us": round(a_us, 2), "frac": round(a_frac, 4),
                             "what": "the same launches replayed alone on one stream (no lanes, no sibling program): the kernel with the chip to itself"}
    return out


def concurrent_phase(fwd, eng, precision, reps=10):
    """The part-batch programs of a forward (Engine._split_bounds: towers / whole programs of independent image groups side by side on
    their own streams) as ONE unit: the span from the fork to the join between two timing events on the caller's stream -- two markers
    per forward, so the forward runs at its product speed -- against the executed FLOPs of all their launches.  This is the chip-level
    rate of that phase with everything overlapping as it does in the product; the per-kernel in-situ figures (one marker per launch,
    timed forward 15-25 % slower) are lower bounds beside it.  -> dict or None when the forward does not split."""
    progs = list(eng.last_concurrent)
    if len(progs) < 2:
        return None
    ev = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]
    eng.phase_events = ev
    try:
        fwd()
        torch.cuda.synchronize()
        span = []
        for _ in range(reps):
            for _ in range(3):  # (back to back: the events hold the LAST forward's span, issued with the host running ahead as in the timed loop)
                fwd()
            torch.cuda.synchronize()
            span.append(ev[0].elapsed_time(ev[1]))
    finally:
        eng.phase_events = None
    flop = sum(f / (WINO_CUT if name.startswith("conv_wino") else 1.0) for P in progs for _, name, f, _, _ in _op_models(P, precision))
    ms = sorted(span)[len(span) // 2]
    tf = flop / (ms * 1e-3) / 1e12
    return {"programs": len(progs), "launches": sum(len(_op_models(P, precision)) for P in progs), "span_ms": round(ms, 3),
            "executed_gflop": round(flop / 1e9, 2), "tflops_executed": round(tf, 2), "frac_of_mfma_peak": round(tf / MFMA_PEAK_TFLOPS[precision], 4),
            "what": "fork -> join of the concurrent part-batch programs, two timing events per forward (median of %d forwards, each the last of three issued back to back)" % reps}


def roofline_report(fwd, prog, precision, cname, eng=None):
    """fwd: the forward whose programs `prog` are (the callable of the timed region); see in_situ_timing"""
    stats, reps, wall_ms = in_situ_timing(fwd, precision)
    alone = standalone_timing(prog, precision)
    total_ms = sum(s["dur_ms"] for s in stats.values())
    order = sorted(stats, key=lambda k: -stats[k]["dur_ms"])
    dom = order[0]  # the kernel with the largest share of the step's kernel time, whatever it is
    r = _kernel_view(dom, stats[dom], reps, total_ms, precision, alone.get(dom))
    r["timing"] = ("in situ: a HIP stop event bound to every dispatch + a start marker in front of it, on the launch's own stream inside the product "
                   "forward (i2r_run_program_timed), %d forwards; avg_launch_us = mean kernel duration with lanes / sibling programs in flight; the markers "
                   "slow the forward (forward_ms_with_timing_events beside ms_per_step), so busy times are upper bounds and frac a lower bound -- "
                   "concurrent_programs (two events per forward) is the unperturbed figure of the part-batch phase as a whole; standalone = the same "
                   "launches alone on one stream (what a rocprofv3 kernel trace shows for the part-batch programs: under it the host is the bottleneck "
                   "and they run one after the other).  Measured and dropped: events on the dominant kernel's launches only -- mixing "
                   "hipExtLaunchKernelGGL and plain launches in one stream cost MORE (w48 forward 6.3 ms, no overlap left)" % reps)
    r["forward_ms_with_timing_events"] = round(wall_ms, 3)
    if eng is not None:
        cp = concurrent_phase(fwd, eng, precision)
        if cp is not None:
            r["concurrent_programs"] = cp
    # the three fractions of the dominant kernel's roof side by side (all measured in this run)
    r["fractions"] = {"in_situ": r["frac"], "in_situ_per_launch": r["frac_per_launch_in_situ"],
                      "standalone": r.get("standalone", {}).get("frac"),
                      "what": "in_situ: all launches of the kernel / the time with >= 1 of them in flight, inside a forward slowed by one event marker per "
                              "launch (lower bound); in_situ_per_launch: one launch / its own duration while it shares the chip; standalone: the kernel with "
                              "the chip to itself"}
    if "concurrent_programs" in r and r["bound"] == "mfma":
        r["fractions"]["concurrent_programs_all_kernels"] = r["concurrent_programs"]["frac_of_mfma_peak"]
        r["fractions"]["what"] += ("; concurrent_programs_all_kernels: EVERY kernel of the part-batch programs together, fork -> join between two events "
                                   "in an unperturbed forward -- the dominant kernel, more efficient than the glue around it, runs above this average")
    # `traffic` (PMC, per launch) next to `algorithmic_bytes` (bench's byte model, per launch, same launch-weighted mean over the kernel's
    # launches): their ratio is computable from the line
    traffic, traffic_src, traffic_what = hbm_traffic(cname, r["kernel"])
    r["traffic"], r["traffic_source"], r["traffic_what"] = traffic, traffic_src, traffic_what
    r["algorithmic_bytes"] = round(stats[dom]["bytes"] / stats[dom]["cnt"])
    r["traffic_over_algorithmic"] = round(traffic / r["algorithmic_bytes"], 3) if traffic and r["algorithmic_bytes"] else None
    conv = [k for k in stats if k.startswith("conv_")]
    if conv:
        t_conv = sum(stats[k]["union_ms"] for k in conv) * 1e-3  # (sum of the kernels' busy times: an upper bound of the convs' share of the step)
        # all convolution launches together, in EXECUTED matrix-pipe FLOPs (Winograd launches counted / 2.25) over the fp32 / 16-bit peak
        ex = sum(stats[k]["flop"] / (WINO_CUT if k.startswith("conv_wino") else 1.0) for k in conv)
        r["all_conv"] = {"tflops_executed": round(ex / t_conv / 1e12, 2), "frac_of_mfma_peak": round(ex / t_conv / 1e12 / MFMA_PEAK_TFLOPS[precision], 4),
                         "tflops_direct_equivalent": round(sum(stats[k]["flop"] for k in conv) / t_conv / 1e12, 2), "busy_ms_per_step": round(t_conv / reps * 1e3, 3)}
    r["kernels"] = [_kernel_view(k, stats[k], reps, total_ms, precision, alone.get(k)) for k in order[:5]]
    for kv in r["kernels"]:
        for drop in ("hbm_view", "mfma_view", "machine_balance_flop_per_byte", "gbytes_per_launch", "algorithm", "direct_equivalent"):
            kv.pop(drop, None)
    r["per_kernel_ms_per_step"] = {k: round(stats[k]["dur_ms"] / reps, 3) for k in order}
    r["per_kernel_avg_launch_us"] = {k: round(stats[k]["dur_ms"] / stats[k]["cnt"] * 1e3, 2) for k in order}
    # The SUM of the in-situ durations over the step's wall time = how many launches are in flight on average (lanes / part-batch
    # programs side by side); filled in by the caller, which knows the step's wall time.
    r["kernel_time_sum_ms_per_step"] = round(total_ms / reps, 3)
    r["_executed_gflop_per_step"] = sum(stats[k]["flop"] / (WINO_CUT if k.startswith("conv_wino") else 1.0) for k in stats) / reps / 1e9
    att_k = sorted(k for k in stats if k.startswith("enc_"))
    att_flop = sum(stats[k]["flop"] for k in att_k) / reps
    att_ms = sum(stats[k]["union_ms"] for k in att_k) / reps  # in situ: time with an encoder kernel of that name in flight
    if att_flop and att_ms:
        att = att_flop / (att_ms * 1e-3) / 1e12
        att_peak = MFMA_PEAK_TFLOPS["fp32" if "enc_layer4_k" in att_k and not any(k_.startswith("enc_layer_lp") for k_ in att_k) else precision]
        r["attention_blocks"] = {"kernels": " + ".join(att_k), "gflop_per_step": round(att_flop / 1e9, 3), "ms_per_step": round(att_ms, 3),
                                 "achieved": round(att, 2), "peak": att_peak, "frac": round(att / att_peak, 4), "timing": "in situ (sum of the kernels' busy times)"}
        att_alone = stack_timing(prog, precision)  # (each encoder stack replayed alone as one unit)
        if att_alone:
            r["attention_blocks"]["standalone"] = {"ms_per_step": round(att_alone, 3), "frac": round(att_flop / (att_alone * 1e-3) / 1e12 / att_peak, 4)}
    return r


LP_TOL = {"bf16": 3e-2, "fp16": 6e-3}  # = tests/test_model_gpu.py LP_TOL: max-abs error as a fraction of max|ref|


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
        out["tolerance"] = "tests/test_model_gpu.py LP_TOL (%s): max-abs <= %g %% of max|ref|" % (precision, LP_TOL[precision] * 100)
        out["ok"] = out["rel_max"] <= LP_TOL[precision]
    return out


def cpu_baseline(cfg, sd, H, W, length, budget_s=20.0):
    """The CPU oracle (a port of the reference forward) timed on this host on the SAME batch shape as the GPU line (`length`: persons per
    image) when that fits the time budget, else on its largest image alone; bounded to ~budget_s seconds.  Threads: the fastest of a few
    candidate counts up to all visible cores (see below) -- `cores` reports what was used, `sample` what was tried."""
    import i2r_cpu
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    # thread count: BASELINE.md asks for the host's cores; torch's CPU convolutions stop scaling long before the 256 hardware threads of
    # the GPU box (and oversubscribe badly beyond), so the candidates {32, 64, all visible / 2, all visible} are each timed on one small
    # forward and the fastest is used -- `cores` reports what was used, `sample` what was tried
    x1, m1, l1 = synth.make_inputs([min(2, max(length))], H, W)
    tried = {}
    for t in sorted({min(cores, 32), min(cores, 64), max(1, cores // 2), cores}):
        torch.set_num_threads(t)
        i2r_cpu.forward(sd, cfg, x1, m1, l1)  # warm-up (thread pool, allocator)
        t0 = time.perf_counter()
        i2r_cpu.forward(sd, cfg, x1, m1, l1)
        tried[t] = time.perf_counter() - t0
        if tried[t] > 4.0 * min(tried.values()):
            break  # (clearly past the scaling knee: do not spend the budget on slower settings)
    threads = min(tried, key=tried.get)
    torch.set_num_threads(threads)
    probe = tried[threads] / l1[0]
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
                      "(%d cores visible; seconds per 2-crop forward by thread count: %s)"
                      % (n, what, H, W, torch.__version__, threads, cores, ", ".join("%d: %.2f" % (t, v) for t, v in sorted(tried.items())))}


def cpu_baseline_short(cfg, sd, H, W, length, threads=32, budget_s=3.0):
    """The CPU oracle on ONE image of the workload (its largest, capped at 4 persons) for a few seconds: the `cpu_baseline` of an
    other_workloads entry (the headline's own leg probes thread counts and runs the whole batch; this one must stay short)."""
    import i2r_cpu
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    threads = min(threads, cores)
    torch.set_num_threads(threads)
    n_p = min(max(length), 4)
    x, m, ls = synth.make_inputs([n_p], H, W)
    i2r_cpu.forward(sd, cfg, x, m, ls)  # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        i2r_cpu.forward(sd, cfg, x, m, ls)
        n += 1
        if time.perf_counter() - t0 > budget_s * 0.5 or n >= 10:
            break
    dt = time.perf_counter() - t0
    return {"value": round(n * n_p / dt, 3), "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": "%d forward(s) of 1 image x %d person(s) at %dx%d, fp32, oracle/i2r_cpu.py on torch %s CPU, %d threads (%d cores visible)"
                      % (n, n_p, H, W, torch.__version__, threads, cores)}


def make_pipeline(net, cfg, length, H, W, dev, seed):
    """The validate() step around the forward as one device-side unit (lib/core/function.py:124-200, JointsDataset.py:296-333):
    uint8 images + person boxes -> ALL affine crops + bbox masks of the batch in one launch, written straight into the collated tensors
    (input.person_inputs_batch: one pinned upload of the crop / image tables, i2r_person_inputs_cv2) -> flip-test forward -> key points
    (i2r_decode).  Synthetic 640x480 images, boxes from the seeded generator; returns step() -> (preds [S,J,2], maxvals [S,J,1]).
    The host's share per step: box -> centre / scale (the dataset's _box2cs), the batched 3-point affine solves, one table upload."""
    from i2r_amd import caller, input as i2r_input
    rng = np.random.default_rng(seed)
    ih, iw = 480, 640
    images, boxes = [], []
    for n in length:
        images.append(torch.from_numpy(rng.integers(0, 256, size=(ih, iw, 3), dtype=np.uint8)).to(dev))
        b = np.stack([rng.uniform(20, iw * 0.5, n), rng.uniform(20, ih * 0.5, n), rng.uniform(60, iw * 0.45, n), rng.uniform(90, ih * 0.45, n)], 1)
        boxes.append(b)
    pairs = caller.FLIP_PAIRS[_dataset_name(cfg)]

    def step():
        cs = [[i2r_input.box_to_center_scale(b, (W, H)) for b in bs] for bs in boxes]   # (per person, as JointsDataset.__getitem__ does)
        x, m, lens, cen, scl = i2r_input.person_inputs_batch(images, [[c for c, _ in one] for one in cs], [[s for _, s in one] for one in cs],
                                                            boxes, (W, H), color_rgb=bool(cfg.DATASET.COLOR_RGB), device=dev)
        hm = net.forward_flip(x, m, lens, pairs)
        return caller.decode(hm, cen, scl, cfg.TEST.BLUR_KERNEL)
    return step


def ragged_stream(net, cfg, dev, H, W, n_batches, flip_pairs=None, seed=0):
    """What validate() really feeds the model (lib/core/function.py:124-140): a stream of batches of TEST.BATCH_SIZE_PER_GPU = 16 images
    whose crop count S = sum(length) changes with every batch (length_i = rng.integers(1, 7): S in about 30..70).  All batches are
    resident on the device before timing.  Returns the report dict: crops/s of a COLD pass (programs built on first use, as the first
    epoch of a validate() run sees them) and of a WARM pass over the same stream, program builds, the padded-crop fraction (a batch
    runs in the smallest pre-built program that holds it: Engine.capacity), and the fixed-S line of the same mean S for comparison."""
    rng = np.random.default_rng(seed)
    batches = []
    for b in range(n_batches):
        length = [int(v) for v in rng.integers(1, 7, size=16)]
        x, m, _ = synth.make_inputs(length, H, W, seed=1000 + b)
        batches.append((x.to(dev), m.to(dev), length))
    eng = net.engine()

    def fwd(x, m, length):
        return net.forward_flip(x, m, length, flip_pairs) if flip_pairs is not None else net(x, m, length)

    def one_pass():
        torch.cuda.synchronize()
        b0, t0 = eng.n_builds, time.perf_counter()
        for x, m, length in batches:
            y = fwd(x, m, length)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, eng.n_builds - b0, y
    crops = sum(sum(b[2]) for b in batches)
    cold_s, cold_builds, _ = one_pass()
    warm_s, warm_builds, y = one_pass()
    assert torch.isfinite(y["multi"] if isinstance(y, dict) else y).all()
    padded = sum(eng.capacity(sum(b[2])) - sum(b[2]) for b in batches)
    # fixed-S reference: every batch has the stream's mean crop count (same number of images)
    mean_s = int(round(crops / n_batches))
    base, extra = divmod(mean_s, 16)
    fixed_len = [base + 1] * extra + [base] * (16 - extra)
    x, m, _ = synth.make_inputs(fixed_len, H, W, seed=999)
    x, m = x.to(dev), m.to(dev)
    for _ in range(3):
        fwd(x, m, fixed_len)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_batches):
        fwd(x, m, fixed_len)
    torch.cuda.synchronize()
    fixed_s = time.perf_counter() - t0
    return {"batches": n_batches, "images_per_batch": 16, "crops_total": crops,
            "crops_per_batch_min_mean_max": [min(sum(b[2]) for b in batches), round(crops / n_batches, 1), max(sum(b[2]) for b in batches)],
            "cold": {"crops_per_s": round(crops / cold_s, 1), "ms_per_batch": round(cold_s / n_batches * 1e3, 3), "program_builds": cold_builds},
            "warm": {"crops_per_s": round(crops / warm_s, 1), "ms_per_batch": round(warm_s / n_batches * 1e3, 3), "program_builds": warm_builds},
            "padded_crop_fraction": round(padded / crops, 4),
            "fixed_s": {"crops_per_batch": mean_s, "crops_per_s": round(mean_s * n_batches / fixed_s, 1), "ms_per_batch": round(fixed_s / n_batches * 1e3, 3)},
            "warm_vs_fixed": round((crops / warm_s) / (mean_s * n_batches / fixed_s), 4)}


def _dataset_name(cfg):
    from i2r_amd import caller
    ds = cfg.DATASET.DATASET.lower()
    return ds if ds in caller.FLIP_PAIRS else ("crowdpose" if cfg.MODEL.NUM_JOINTS == 14 else "coco")


def build_net(cname, precision, dev):
    cfg = config.load_config(cname)
    sd = synth.make_state_dict(arch.param_spec(cfg))
    net = eval("models." + cfg.MODEL.NAME + ".get_pose_net")(cfg, is_train=False)
    net.load_state_dict(sd, strict=True)
    return cfg, sd, net.to(dev).set_precision(precision)


def _time_steps(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        y = fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, y


def _brief(kv):
    """the figures of a _kernel_view that identify the kernel and its roofline fraction"""
    keep = ("kernel", "launches_per_step", "avg_launch_us", "launches_in_flight", "busy_ms_per_step", "share_of_step_kernel_time", "bound", "achieved",
            "peak", "unit", "frac", "frac_per_launch_in_situ")
    out = {k: kv[k] for k in keep if k in kv}
    if "standalone" in kv:
        out["standalone_avg_launch_us"], out["standalone_frac"] = kv["standalone"]["avg_launch_us"], kv["standalone"]["frac"]
    for view in ("mfma_view", "hbm_view"):
        if view in kv:
            out[view + "_frac"] = kv[view]["frac"]
    return out


def quick_workload(cname, dev, steps=30, warmup=5):
    """One of the other BASELINE workloads at its own batch shape and dtype, measured like the headline but short: `steps` timed forwards
    (inputs resident), the per-launch roofline pass (dominant kernel + the next two, attention blocks), parity of image 0 against the
    CPU oracle.  Everything it allocates is released on return."""
    wl = WORKLOADS[cname]
    precision = wl["precision"]
    cfg, sd, net = build_net(cname, precision, dev)
    W_, H_ = cfg.MODEL.IMAGE_SIZE
    length = list(wl["length"])
    x, m, _ = synth.make_inputs(length, H_, W_, seed=0)
    x, m = x.to(dev), m.to(dev)

    def fwd():
        y = net(x, m, length)
        return y["multi"] if isinstance(y, dict) else y
    dt, y = _time_steps(fwd, steps, warmup)
    assert torch.isfinite(y).all()
    eng = net.engine()
    r = roofline_report(fwd, eng.last_programs, precision, cname, eng)  # (the forward just timed and its program(s))
    gflop = sum(n * wl["gflop"](n) for n in length)
    out = {"workload": wl["label"], "dtype": DTYPE_NAME[precision], "crops_per_step": sum(length), "steps": steps, "warmup": warmup,
           "value": round(sum(length) * steps / dt, 1), "unit": "images/sec", "ms_per_step": round(dt / steps * 1e3, 3),
           "model_tflops_algorithmic": round(gflop * steps / dt / 1e3, 2),
           "dominant_kernel": _brief(r), "next_kernels": [_brief(k) for k in r["kernels"][1:3]],
           "traffic": r.get("traffic"), "algorithmic_bytes": r.get("algorithmic_bytes"), "traffic_over_algorithmic": r.get("traffic_over_algorithmic"),
           "traffic_source": r.get("traffic_source"), "traffic_what": r.get("traffic_what"),
           "kernel_time_sum_ms_per_step": r["kernel_time_sum_ms_per_step"],
           "kernel_time_overlap": round(r["kernel_time_sum_ms_per_step"] / (dt / steps * 1e3), 3),
           "per_kernel_avg_launch_us": r["per_kernel_avg_launch_us"], "forward_ms_with_timing_events": r["forward_ms_with_timing_events"]}
    if "concurrent_programs" in r:
        out["concurrent_programs"] = r["concurrent_programs"]
    if any(getattr(P, "uses_lanes", False) for P in eng.last_programs):
        out["device_side_lane_sync"] = any(getattr(P, "device_sync", False) for P in eng.last_programs)
        out["device_side_waits_timed_out"] = any(P.sync_timed_out() for P in eng.last_programs)
        assert not out["device_side_waits_timed_out"], "a device-side lane wait timed out: the forward's results are not ordered"
    if "attention_blocks" in r:
        out["attention_blocks"] = {k: r["attention_blocks"][k] for k in ("kernels", "ms_per_step", "achieved", "peak", "frac", "standalone") if k in r["attention_blocks"]}
    out["parity"] = oracle_parity(cfg, sd, x, m, length, fwd(), precision)
    out["cpu_baseline"] = cpu_baseline_short(cfg, sd, H_, W_, length)
    return out


def other_workloads(net, cfg, dev):
    """BASELINE configs[2..4] + the two caller-side modes, each measured briefly in this process (the driver runs `bench.py --gpus 1` only):
    {name: {value, ms_per_step, dtype, dominant kernel + frac, parity}}.  `net` / `cfg` = the headline model (re-used for the ragged
    stream and the pipeline, which BASELINE quotes on the vanilla W48 model)."""
    from i2r_amd import caller
    out = {}
    for cname in ("tph_192_p6_b4", "hrt_192_p4_b4", "coco_hrt_288_p2_b4"):
        try:
            out[cname] = quick_workload(cname, dev)
        except Exception as e:  # (reported in the entry: the headline of the line must survive a side workload's failure)
            out[cname] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        torch.cuda.empty_cache()
    W_, H_ = cfg.MODEL.IMAGE_SIZE
    rs = ragged_stream(net, cfg, dev, H_, W_, 16)
    out["ragged_stream"] = {"what": "16 batches of 16 images with 1-6 persons each (S changes per batch, lib/core/function.py:124-140), w48 fp32",
                            "value": rs["warm"]["crops_per_s"], "unit": "images/sec", "ms_per_batch": rs["warm"]["ms_per_batch"],
                            "cold_value": rs["cold"]["crops_per_s"], "program_builds_cold": rs["cold"]["program_builds"],
                            "padded_crop_fraction": rs["padded_crop_fraction"], "fixed_s_value": rs["fixed_s"]["crops_per_s"],
                            "warm_vs_fixed": rs["warm_vs_fixed"]}
    length = list(WORKLOADS["w48_pure_en6"]["length"])
    pipe = make_pipeline(net, cfg, length, H_, W_, dev, seed=0)
    dt, (preds, maxv) = _time_steps(pipe, 10, 3)
    assert torch.isfinite(preds).all() and torch.isfinite(maxv).all()
    fwd_ms = _pipeline_forward_ms(net, cfg, length, H_, W_, dev)
    out["pipeline"] = {"what": "uint8 image -> affine crops + bbox masks -> flip-test forward (2 x 32 crops) -> key-point decode, w48 fp32 "
                               "(lib/core/function.py:124-200)", "value": round(sum(length) * 10 / dt, 1), "unit": "images/sec",
                       "ms_per_step": round(dt / 10 * 1e3, 3), "forward_flip_only_ms": round(fwd_ms, 3),
                       "share_outside_forward": round(1.0 - fwd_ms / (dt / 10 * 1e3), 4)}
    return out


def _pipeline_forward_ms(net, cfg, length, H_, W_, dev):
    """ms of the flip-test forward alone (inputs resident): what the pipeline step costs without crops / masks / decode"""
    from i2r_amd import caller
    x, m, _ = synth.make_inputs(length, H_, W_, seed=0)
    x, m = x.to(dev), m.to(dev)
    pairs = caller.FLIP_PAIRS[_dataset_name(cfg)]
    dt, _ = _time_steps(lambda: net.forward_flip(x, m, length, pairs), 10, 3)
    return dt / 10 * 1e3


def collective_overhead(dev, cname="hrt_192_p4_b4", rounds=5, steps=20):
    """What the multi-GPU step costs at N = 1, measured A/B in ONE process: BASELINE configs[3] (the workload BASELINE quotes on 8 GPUs,
    16 crops per GPU) timed `rounds` times alternately as the plain step and as the step every rank of an N > 1 run executes -- device
    decode + asynchronous all-gather of the key points through a ONE-rank RCCL process group, waited for one step later.  The group is
    created here, after the engine's lane streams exist (as main() orders it for the N > 1 ranks).  -> medians and their ratio; the
    fresh-process form of the same comparison is tools/collective_overhead.py -> profiles/round6_collective.json."""
    from i2r_amd import caller
    own = not dist.is_initialized()
    if own:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        wl = WORKLOADS[cname]
        cfg, sd, net = build_net(cname, wl["precision"], dev)
        W_, H_ = cfg.MODEL.IMAGE_SIZE
        length = list(wl["length"])
        x, m, _ = synth.make_inputs(length, H_, W_, seed=0)
        x, m = x.to(dev), m.to(dev)
        counts = [sum(length)]
        def plain():
            y = net(x, m, length)
            return y["multi"] if isinstance(y, dict) else y

        posts = {"keypoints": i2r_dist.PostStep(dev, counts, decode=lambda t: caller.decode(t, None, None, cfg.TEST.BLUR_KERNEL, transform_back=False)),
                 "heatmaps": i2r_dist.PostStep(dev, counts)}

        def make(payload):
            def step():
                y = plain()
                posts[payload](y)
                return y
            return step

        def run(fn):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            for ps in posts.values():  # (the last step's decode + gather belong to the timed region)
                ps.result()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / steps * 1e3
        kp, hm = make("keypoints"), make("heatmaps")
        for fn in (plain, kp, hm):
            run(fn)
        ms = {"plain": [], "keypoints": [], "heatmaps": []}
        for _ in range(rounds):
            ms["plain"].append(run(plain))
            ms["keypoints"].append(run(kp))
            ms["heatmaps"].append(run(hm))
        med = {k: sorted(v)[len(v) // 2] for k, v in ms.items()}
        return {"workload": wl["label"], "what": "ms per step at N = 1, %d alternating rounds of %d steps in one process: plain forward | forward + device decode + "
                                                 "async RCCL all-gather of the key points (one-rank group) | forward + all-gather of the heat maps; decode and "
                                                 "gather run on a side stream under the next forward (dist.PostStep)" % (rounds, steps),
                "ms_per_step": {k: [round(t, 4) for t in v] for k, v in ms.items()}, "median_ms": {k: round(v, 4) for k, v in med.items()},
                "overhead_keypoints": round(med["keypoints"] / med["plain"] - 1.0, 4), "overhead_heatmaps": round(med["heatmaps"] / med["plain"] - 1.0, 4)}
    finally:
        if own:
            dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------------------------------------
def self_launch(argv, n):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one process per GPU) under torch.distributed.run on the
    loopback interface and hand their exit code back.  The driver's own form (python -m torch.distributed.run ... bench.py --gpus N)
    sets WORLD_SIZE and never comes here."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL's intra-node transport on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="w48_pure_en6", choices=sorted(WORKLOADS),
                    help="workload (default = BASELINE configs[1]); the others are BASELINE configs[2..4] with their own batch shapes")
    ap.add_argument("--precision", default=None, choices=["fp32", "bf16", "fp16"],
                    help="MFMA operand type (default: the workload's BASELINE dtype -- fp32 / bf16 / bf16 / fp16)")
    ap.add_argument("--pipeline", action="store_true", help="time image -> crops -> flip-test forward -> key points instead of the bare forward")
    ap.add_argument("--ragged-stream", action="store_true",
                    help="time a stream of 64 batches of 16 images with 1-6 persons each (S changes per batch, as in validate()); with --pipeline: flip-test forwards")
    ap.add_argument("--gather", default="keypoints", choices=["keypoints", "heatmaps"],
                    help="N > 1: payload of the per-step all-gather whose timing is `value`: the key points decoded on the device [S,J,3] "
                         "(168 B per crop: what validate() keeps of a batch, lib/core/function.py:190-200) or the heat maps [S,J,h,w] "
                         "(172 KB per crop, the payload north_star names); the other payload is re-timed and reported as gather_alt")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): every rank runs a batch of the workload's shape.  strong: a FIXED ragged list of 64 images with 1-6 persons "
                         "each (rng seed 0) is cut into contiguous shards balanced by crop count (dist.shard_bounds); a step = one pass over the whole "
                         "list, each rank running its images in batches of <= 16 (TEST.BATCH_SIZE_PER_GPU) and one padded all-gather of its uneven crop count")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="default single-GPU run only: skip the short runs of BASELINE configs[2..4], the ragged stream and the pipeline that are "
                         "embedded in the line as other_workloads")
    ap.add_argument("--world1-collective", action="store_true",
                    help="--gpus 1 only: create a ONE-rank process group and run the per-step all-gather, the barriers and the max-over-ranks "
                         "reduction exactly as the N > 1 ranks do (the N = 1 end of the scaling line with the collective in place; "
                         "tests/test_dist_gpu.py checks it against the plain N = 1 line)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help=argparse.SUPPRESS)
    ap.add_argument("--selftest-stub", action="store_true",
                    help=argparse.SUPPRESS)  # tests/test_bench_launcher.py: the launcher / shard / gather / timing path on CPU with gloo; the model step is a constant tensor and the line says so
    return ap.parse_args(argv)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse_args(argv)
    refuse_tuning_env()

    if args.scaling == "strong" and (args.pipeline or args.ragged_stream):
        raise SystemExit("--scaling strong times the bare forward over the fixed image list (no --pipeline / --ragged-stream)")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(argv, args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "WORLD_SIZE=%d but --gpus %d" % (world, args.gpus)
    stub = args.selftest_stub
    if args.world1_collective and world != 1:
        raise SystemExit("--world1-collective is the N = 1 form of the multi-GPU step (use it with --gpus 1)")
    coll = world > 1 or args.world1_collective  # the step ends in the collective
    if not stub and torch.cuda.is_available():
        # the engine's lane / part-batch streams are created (and probed for overlap) BEFORE any RCCL communicator exists, so which hardware
        # queues they sit on does not depend on the streams RCCL creates; what the probe found goes into the line: `lanes`
        from i2r_amd import engine as _engine
        torch.cuda.set_device(local_rank)
        _engine.lane_streams(torch.device("cuda", local_rank), 3)
    if coll:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {}
        if world == 1 and "RANK" not in os.environ:  # (no launcher: a one-rank group on a free loopback port)
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            kw = dict(rank=0, world_size=1)
        if args.backend == "nccl" and torch.cuda.is_available():  # "nccl" IS RCCL on ROCm; bind the communicator to this rank's GPU
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), **kw)
        else:  # (gloo self-test; or no GPU at all: ProcessGroupNCCL then says so itself)
            dist.init_process_group(args.backend, **kw)
    if stub:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)

    wl = WORKLOADS[args.config]
    precision = args.precision or wl["precision"]
    cfg = config.load_config(args.config)
    W_, H_ = cfg.MODEL.IMAGE_SIZE
    J = cfg.MODEL.NUM_JOINTS
    sd = net = None
    if not stub:
        cfg, sd, net = build_net(args.config, precision, dev)

    strong = args.scaling == "strong"
    if strong:
        # strong scaling: the total work is fixed -- 64 images of 1-6 persons (the crop counts validate() sees on CrowdPose, MAX_PATCH 6) --
        # and is cut into contiguous image shards balanced by crop count; the shards are UNEVEN in crops, so the padded all-gather and
        # the max-over-ranks timing carry the imbalance that dist.shard_bounds leaves
        length_all = STRONG_LENGTH
        bounds = i2r_dist.shard_bounds(length_all, world)
    else:
        # weak scaling: every rank's batch has the workload's shape; this rank's contiguous shard of the image list
        per_gpu = list(wl["length"])
        length_all = per_gpu * world
        bounds = i2r_dist.shard_bounds(length_all, world) if world > 1 else [0, len(length_all)]
        if world > 1:  # identical shapes -> the balanced cuts are the per-GPU batches
            assert [bounds[r + 1] - bounds[r] for r in range(world)] == [len(per_gpu)] * world, bounds
    length = length_all[bounds[rank]:bounds[rank + 1]]
    counts = [sum(length_all[bounds[r]:bounds[r + 1]]) for r in range(world)]
    gflop_per_step = sum(n * wl["gflop"](n) for n in length_all)
    # this rank's forwards of one step: the whole shard (weak), or batches of <= 16 images (strong; TEST.BATCH_SIZE_PER_GPU of the yamls)
    chunks = [length[i:i + 16] for i in range(0, len(length), 16)] if strong else [length]
    batches = []
    x = m = None
    if not stub:
        for ci, ln in enumerate(chunks):
            if not ln:
                continue
            xb, mb, _ = synth.make_inputs(ln, H_, W_, seed=rank * 16 + ci)
            batches.append((xb.to(dev), mb.to(dev), ln))
        if batches:
            x, m = batches[0][0], batches[0][1]

    from i2r_amd import caller
    pending = [None]
    pipe = make_pipeline(net, cfg, length, H_, W_, dev, seed=rank) if (args.pipeline and not stub and not args.ragged_stream) else None
    stub_y = torch.full((sum(length), J, H_ // 4, W_ // 4), float(rank), dtype=torch.float32) if stub else None

    def make_step(gather):
        def step():
            """one forward over this rank's images; N > 1: the all-gather of step k is waited for after step k+1 has been issued"""
            h = None
            if stub:
                y = stub_y
                if coll:
                    h = (i2r_dist.gather_keypoints(y[:, :, :1, 0].expand(-1, -1, 2), y[:, :, :1, 0], counts, async_op=True)
                         if gather == "keypoints" else i2r_dist.gather_heatmaps_async(y, counts))
            elif pipe is not None:
                preds, maxv = pipe()
                y = torch.cat([preds, maxv], 2)
                if coll:
                    h = i2r_dist.gather_heatmaps_async(y, counts)
            else:
                ys = []
                for xb, mb, ln in batches:
                    yb = net(xb, mb, ln)
                    ys.append(yb["multi"] if isinstance(yb, dict) else yb)
                # (strong scaling: a rank's forwards of one step are gathered together -- ranks run different numbers of forwards)
                y = ys[0] if len(ys) == 1 else (torch.cat(ys, 0) if ys else torch.zeros(0, J, H_ // 4, W_ // 4, device=dev))
                if coll:  # decode (key points: [S, J, 3], 168 B/crop instead of 172 KB/crop) + all-gather on a side stream, under the next forward
                    post[gather](y)
                    return y
            if coll:
                if pending[0] is not None:
                    pending[0].wait()
                pending[0] = h
            return y
        return step

    post = {}
    if coll and not stub:
        post = {"keypoints": i2r_dist.PostStep(dev, counts, decode=lambda t: caller.decode(t, None, None, cfg.TEST.BLUR_KERNEL, transform_back=False)),
                "heatmaps": i2r_dist.PostStep(dev, counts)}

    def drain():
        if pending[0] is not None:
            pending[0].wait()
            pending[0] = None
        for ps in post.values():
            ps.result()

    def sync():
        if not stub:
            torch.cuda.synchronize()

    def timed(step, steps, warmup):
        """warmup untimed steps, then EXACTLY `steps` steps bracketed by barrier + synchronize on both sides; max over ranks"""
        for _ in range(warmup):
            y = step()
        drain()
        sync()
        if coll:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            y = step()
        drain()
        sync()
        if coll:
            dist.barrier()
        sync()
        dt = time.perf_counter() - t0
        if coll:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
        return dt, y

    dt, y = timed(make_step(args.gather), args.steps, args.warmup)
    assert torch.isfinite(y).all()
    alt = None
    if coll and pipe is None:
        other = "keypoints" if args.gather == "heatmaps" else "heatmaps"
        dt2, _ = timed(make_step(other), args.steps, min(args.warmup, 3))
        alt = {"payload": other, "ms_per_step": round(dt2 / args.steps * 1e3, 4), "value": round(sum(length_all) * args.steps / dt2, 2)}

    crops_per_step = sum(length_all)
    value = crops_per_step * args.steps / dt
    payload = "key points" if (args.pipeline or args.gather == "keypoints") else "heat maps"
    first = batches[0][2] if batches else length  # the batch the roofline / parity legs look at
    out = {
        "metric": "images/sec (%dx%d crops) I2R-Net inference" % (H_, W_) if args.config != "w48_pure_en6"
                  else "images/sec (256x192 crops) I2R-Net HRNet-W48 inference",
        "value": round(value, 2), "unit": "images/sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": DTYPE_NAME[precision], "data": "synthetic",
        "config": {"workload": wl["label"] + ("" if precision == wl["precision"] else " -- run with %s MFMA operands" % precision)
                               + (" -- PIPELINE: uint8 image -> crops + masks -> flip-test forward -> key points" if args.pipeline else "")
                               + (" -- STRONG SCALING: the batch shape is replaced by a fixed list of 64 images with 1-6 persons (%d crops), "
                                  "run in batches of <= 16 images" % sum(STRONG_LENGTH) if strong else ""),
                   "images_per_gpu": len(length), "persons_per_image": length if len(set(length)) > 1 else (length[0] if length else 0),
                   "crops_per_gpu_step": sum(length),
                   "parallelism": "dp%d (images sharded, one RCCL all-gather of the %s per step, issued on a side stream under the next forward)" % (world, payload)
                                  if coll else "single GPU",
                   "gflop_per_step_per_gpu": round(gflop_per_step / world, 2),
                   "programs_per_forward": (len(net.engine().last_programs) if (net is not None and getattr(net.engine(), "last_programs", None)) else 1)},
        # SURVEY 8d algorithmic FLOPs of the reference forward (direct convolutions) per second of wall time: a delivered-work figure,
        # NOT a fraction of any pipe (the Winograd launches execute 2.25x fewer multiply-adds); roofline.model_tflops_executed is
        "model_tflops_algorithmic": round(gflop_per_step * (2 if args.pipeline else 1) * args.steps / dt / 1e3, 2),
    }
    if strong:
        out["shards"] = {"images_total": len(length_all), "crops_total": sum(length_all), "image_bounds": bounds, "crops_per_rank": counts,
                         "forwards_per_rank_step": [-(-(bounds[r + 1] - bounds[r]) // 16) for r in range(world)],
                         "imbalance_max_over_mean": round(max(counts) * world / float(sum(counts)), 4),
                         "gather_rows_padded_to": max(counts)}
    if alt is not None:
        out["gather_alt"] = alt
    if stub:
        out["data"] = "SELFTEST STUB: no model ran (launcher / sharding / all-gather / timing path only)"
        out["value"], out["model_tflops_algorithmic"] = 0.0, 0.0
        out["gathered_ok"] = True
    if rank == 0 and not stub:
        eng = net.engine()
        if not args.no_roofline:
            def fwd1():
                if pipe is not None:
                    return pipe()
                for xb, mb, ln in batches:
                    net(xb, mb, ln)
            out["roofline"] = roofline_report(fwd1, eng.last_programs, precision, args.config, eng)  # (the forward(s) of the timed step and the last one's program(s))
            if not strong and not args.pipeline:  # executed matrix-pipe + element-wise FLOPs of one forward over the step's wall time
                out["roofline"]["model_tflops_executed"] = round(out["roofline"].pop("_executed_gflop_per_step") * args.steps / dt / 1e3, 2)
            out["roofline"].pop("_executed_gflop_per_step", None)
            out["roofline"]["kernel_time_overlap"] = round(out["roofline"]["kernel_time_sum_ms_per_step"] / (dt / args.steps * 1e3), 3)
        if not args.no_parity and not args.pipeline:
            y1 = net(x, m, first)
            y1 = y1["multi"] if isinstance(y1, dict) else y1
            out["parity"] = oracle_parity(cfg, sd, x, m, first, y1, precision)
        if args.ragged_stream:
            pairs = None
            if args.pipeline:
                pairs = caller.FLIP_PAIRS[_dataset_name(cfg)]
            out["ragged_stream"] = ragged_stream(net, cfg, dev, H_, W_, 64, flip_pairs=pairs)
        default_line = (world == 1 and args.config == "w48_pure_en6" and not args.pipeline and not args.ragged_stream and not strong
                        and args.precision in (None, "fp32"))
        if default_line and not args.no_other_workloads:
            # the driver only runs `bench.py --gpus 1`: BASELINE configs[2..4] at their own batch shapes / dtypes, the ragged stream and
            # the pipeline are measured in the same process right after the headline's timed region, 10-30 steps each
            t_other = time.perf_counter()
            out["other_workloads"] = other_workloads(net, cfg, dev)
            out["other_workloads"]["wall_s"] = round(time.perf_counter() - t_other, 1)
        from i2r_amd import engine as _engine
        out["lanes"] = _engine.lane_report(dev)
        # a device-side wait that saw nothing for 50 ms gives up and raises a flag instead of hanging the GPU: none may have (the step's results
        # would not be ordered); checked here, behind the timed region's synchronize
        out["lanes"]["device_side_waits_timed_out"] = any(P.sync_timed_out() for v in eng.programs.values() for P in v[:1])
        if default_line and not args.no_other_workloads and not coll:
            try:
                out["collective_overhead"] = collective_overhead(dev)
            except Exception as e:  # (an RCCL group that cannot be created here must not cost the driver its bench line)
                out["collective_overhead"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(cfg, sd, H_, W_, first)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if coll:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
