// Pair of 1x1 convolutions on NHWC rows as a register-chained token GEMM (fp32 matrix pipe), gfx950.
//
//   y = act_a(W_a x + b_a (+ res))        [n_pix, ca_out]   -- written out (it is the residual of the next block)
//   z = act_b(W_b y + b_b)                [n_pix, cb_out]   -- optional
//
// Replaces conv3 + bn3 + residual + ReLU of a Bottleneck TOGETHER WITH conv1 + bn1 + ReLU of the next one (reference
// lib/models/hrnet.py Bottleneck.forward / interformer_pureMulti.py:69-107): as two launches of the generic conv kernel the 256-channel
// map y is written by the first and read back by the second (3.1 MB per crop each way at 64x48), and both launches are bound by their
// epilogues, not by arithmetic.  Here a wave owns MT tiles of 16 pixels and computes everything TRANSPOSED, like the encoder kernels
// (i2r_encoder.hip): Y^T[f] = W_a[f] X^T with the weight fragment as MFMA A operand and the pixels' channel quadruples as B operand
// (straight 16-byte loads from the NHWC rows, no LDS); the D fragment (lane (pixel li, g): channels 16 f + 4 g + r) takes bias /
// residual / ReLU, is stored as one 16-byte piece of the pixel's row, and IS the B operand image of the second GEMM, which
// accumulates Z^T += W_b[:, f] Y^T[f] fragment by fragment -- y never comes back from memory.  No LDS, no barrier: the four waves of
// a workgroup are independent.  Weights are fragment-packed by the host (engine.pack_frag: one 64-lane 16-byte load = 1 KB
// contiguous), fetched one channel fragment ahead.
#include "i2r_common.h"
#include "i2r_conv.h"

namespace {

struct PairK {
    const float* x; const float* w_a; const float* b_a; const float* res; float* y;
    const float* w_b; const float* b_b; float* z;
    int n_pix, ca_frag, x_cs, y_cs, z_cs, n_tiles;
    float lo_a, lo_b;  // activation floor: 0 (ReLU) or -inf (none)
};

__device__ __forceinline__ f32x4 floor4(f32x4 v, float lo) { return (f32x4){fmaxf(v[0], lo), fmaxf(v[1], lo), fmaxf(v[2], lo), fmaxf(v[3], lo)}; }

// KA: 16-channel steps of the first conv's input (k_a / 16); NB: 16-channel fragments of z (0 = no second conv); MT: pixel tiles per
// wave; RES: a residual is added.  All global accesses are buffer instructions (32-bit lane offsets, the fragment index in the scalar
// offset; a lane past the last pixel sits at kOOB: reads zeros, stores dropped) -- the loop body is one basic block.
template <int KA, int NB, int MT, bool RES>
__global__ __launch_bounds__(256) void conv1x1_pair_k(const PairK p) {
    constexpr int NBB = NB > 0 ? NB : 1;
    const int lane = threadIdx.x & 63, li = lane & 15, g = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));  // waves are independent: wave wv owns pixel tiles wv * MT ..
    const int tile0 = wv * MT;
    if (tile0 >= p.n_tiles) return;
    const int FA = p.ca_frag;
    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(p.x, (unsigned)p.n_pix * p.x_cs * 4), rs_y = make_rsrc(p.y, (unsigned)p.n_pix * p.y_cs * 4),
                                 rs_r = make_rsrc(p.res, (unsigned)p.n_pix * p.y_cs * 4), rs_z = make_rsrc(p.z, (unsigned)p.n_pix * p.z_cs * 4),
                                 rs_wa = make_rsrc(p.w_a, (unsigned)FA * KA * 1024), rs_wb = make_rsrc(p.w_b, (unsigned)FA * NB * 1024),
                                 rs_ba = make_rsrc(p.b_a, (unsigned)FA * 64);
    unsigned xo[MT], yo[MT], zo[MT];  // byte offsets of this lane's 16-byte piece (channels 4 g ..) in its pixel's rows
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int q = (tile0 + mt) * 16 + li;
        const bool ok = q < p.n_pix;
        xo[mt] = ok ? (unsigned)(q * p.x_cs + 4 * g) * 4 : kOOB;
        yo[mt] = ok ? (unsigned)(q * p.y_cs + 4 * g) * 4 : kOOB;
        zo[mt] = ok ? (unsigned)(q * p.z_cs + 4 * g) * 4 : kOOB;
    }
    // B operands of the first GEMM: channels 16 c + 4 g .. + 3 of the lane's pixels
    f32x4 xb[MT][KA];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int c = 0; c < KA; ++c) xb[mt][c] = buf_ld16(rs_x, xo[mt], c * 64);
    f32x4 zacc[MT][NBB];
    if constexpr (NB > 0) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(p.b_b + 16 * nb + 4 * g);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) zacc[mt][nb] = b;
        }
    }
    // everything channel fragment f needs from memory, fetched one fragment ahead: W_a[f][c] (KA KB), W_b[nb][f] (NB KB), the bias
    // quadruple and the residual pieces of the MT pixels
    struct Frag { f32x4 a[KA], b[NBB], r[MT], bias; };
    auto fetch = [&](int f, Frag& t) {
#pragma unroll
        for (int c = 0; c < KA; ++c) t.a[c] = buf_ld16(rs_wa, lane * 16, (f * KA + c) * 1024);
        if constexpr (NB > 0) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) t.b[nb] = buf_ld16(rs_wb, lane * 16, (nb * FA + f) * 1024);
        }
        t.bias = buf_ld16(rs_ba, g * 16, f * 64);
        if constexpr (RES) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) t.r[mt] = buf_ld16(rs_r, yo[mt], f * 64);
        }
    };
    auto step = [&](int f, const Frag& t) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f32x4 acc = t.bias;
            if constexpr (RES) acc += t.r[mt];  // (bias and residual enter as the accumulator's initial value)
#pragma unroll
            for (int c = 0; c < KA; ++c)
#pragma unroll
                for (int s = 0; s < 4; ++s) acc = mfma16(t.a[c][s], xb[mt][c][s], acc);  // Y^T[16 f + 4 g + r][pixel li]
            acc = floor4(acc, p.lo_a);
            buf_st16(rs_y, yo[mt], f * 64, acc);
            if constexpr (NB > 0) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int s = 0; s < 4; ++s) zacc[mt][nb] = mfma16(t.b[nb][s], acc[s], zacc[mt][nb]);  // Z^T += W_b[nb][f] Y^T[f]
            }
        }
    };
    Frag t0, t1;
    // (sched_barrier: the machine scheduler otherwise sinks the look-ahead loads to just before their use to save registers)
    fetch(0, t0);
    for (int f = 0; f < FA; f += 2) {  // (FA is even: checked by the host)
        fetch(f + 1, t1);
        __builtin_amdgcn_sched_barrier(0);
        step(f, t0);
        __builtin_amdgcn_sched_barrier(0);
        fetch(f + 2 < FA ? f + 2 : f, t0);  // (the look-ahead past the end re-reads a fragment, unused)
        __builtin_amdgcn_sched_barrier(0);
        step(f + 1, t1);
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (NB > 0) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                buf_st16(rs_z, zo[mt], nb * 64, floor4(zacc[mt][nb], p.lo_b));
    }
}

typedef void (*pair_fn)(const PairK);
template <int KA, int NB, bool RES>
pair_fn pick_mt(int mt) {
    if (mt == 1) return conv1x1_pair_k<KA, NB, 1, RES>;
    if (mt == 2) return conv1x1_pair_k<KA, NB, 2, RES>;
    if (mt == 4) return conv1x1_pair_k<KA, NB, 4, RES>;
    return nullptr;
}
template <int KA, int NB>
pair_fn pick_res(int mt, bool res) { return res ? pick_mt<KA, NB, true>(mt) : pick_mt<KA, NB, false>(mt); }

}  // namespace

extern "C" int i2r_conv1x1_pair(const i2r_conv1x1_pair_args* a, void* stream) {
    I2R_CHECK_ARG(a && a->x && a->w_a && a->b_a && a->y, "i2r_conv1x1_pair: null pointer");
    I2R_CHECK_ARG(a->n_pix > 0 && (a->k_a == 64 || a->k_a == 128) && a->ca_out > 0 && a->ca_out % 32 == 0 && a->x_cs >= a->k_a && a->y_cs >= a->ca_out &&
                      a->x_cs % 4 == 0 && a->y_cs % 4 == 0,
                  "i2r_conv1x1_pair: n_pix=%d k_a=%d (64 | 128) ca_out=%d (whole 32-channel steps) x_cs=%d y_cs=%d", a->n_pix, a->k_a, a->ca_out, a->x_cs, a->y_cs);
    I2R_CHECK_ARG(a->cb_out == 0 || (a->cb_out == 64 && a->w_b && a->b_b && a->z && a->z_cs >= 64 && a->z_cs % 4 == 0),
                  "i2r_conv1x1_pair: the second conv has 64 output channels (cb_out=%d)", a->cb_out);
    I2R_CHECK_ARG((const float*)a->y != a->x && a->z != a->y && a->y != a->res, "i2r_conv1x1_pair: outputs alias inputs");
    const int64_t cs_max = a->y_cs > a->x_cs ? a->y_cs : a->x_cs;
    I2R_CHECK_ARG((int64_t)a->n_pix * cs_max * 4 < (int64_t)kOOB, "i2r_conv1x1_pair: tensors must stay below 2 GiB (32-bit buffer offsets)");
    const int mt = a->mt ? a->mt : 2;
    PairK k;
    k.x = a->x; k.w_a = a->w_a; k.b_a = a->b_a; k.res = a->res; k.y = a->y; k.w_b = a->w_b; k.b_b = a->b_b; k.z = a->z;
    k.n_pix = a->n_pix; k.ca_frag = a->ca_out / 16; k.x_cs = a->x_cs; k.y_cs = a->y_cs; k.z_cs = a->z_cs;
    k.lo_a = a->relu_a ? 0.f : -INFINITY; k.lo_b = a->relu_b ? 0.f : -INFINITY;
    k.n_tiles = (a->n_pix + 15) / 16;
    pair_fn fn = nullptr;
    if (a->k_a == 64) fn = a->cb_out ? pick_res<4, 4>(mt, a->res != nullptr) : pick_res<4, 0>(mt, a->res != nullptr);
    else fn = a->cb_out ? pick_res<8, 4>(mt, a->res != nullptr) : pick_res<8, 0>(mt, a->res != nullptr);
    I2R_CHECK_ARG(fn != nullptr, "i2r_conv1x1_pair: mt=%d (1, 2, 4)", mt);
    const int waves = (k.n_tiles + mt - 1) / mt;
    i2r_launch(fn, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, k);
    I2R_CHECK_LAUNCH("i2r_conv1x1_pair");
    return I2R_OK;
}
