// 1x1 convolution over a SMALL number of NHWC pixel rows with 16-bit MFMA operands, straight from global memory (gfx950).
//
//   out = act(W x + b [+ res1]) [+ res_post]          x: [n_pix, x_cs] fp32 or 16 bit, out / res: [n_pix, out_cs] fp32 or 16 bit
//
// The low-resolution branches of HRFormer-B (C = 312 @ 16x12, C = 624 @ 8x6: 3072 / 768 pixels at 16 crops) run their transformer blocks
// as single launches of LayerNorm, q|k|v projection, window attention, out projection, fc1, depth-wise conv and fc2 on the critical
// stream lane (reference lib/models/hrformer.py:1230-1240, MlpDWBN :1094-1119, InterlacedPoolAttention :1164-1180), and so do the 1x1
// convolutions of the fuse layers (:1629-1704).  On the implicit-GEMM kernel such a conv is a few hundred workgroups that stage the
// pixels' channels through LDS in 64-channel chunks, one barrier pair per chunk, for 8 MFMAs each: 15-36 us for 0.3-2.4 GFLOP, all of
// it latency.  Here the GEMM is computed TRANSPOSED (Y^T = W X^T, as in the encoder kernels): the weight fragment is the MFMA A
// operand (fragment-packed by the host: one 64-lane 8-byte load = 512 contiguous bytes), a pixel's 4 consecutive channels the B operand
// (one 8-byte load from its row, or 16 bytes of fp32 packed on the fly) -- no LDS in the K loop, no barrier.  A workgroup owns MT
// 16-pixel tiles x NF 16-channel output fragments and its four waves SPLIT K (cin / 16 steps, two steps fetched ahead): that is what
// puts enough waves on the chip for these shapes; the four partial sums meet once through LDS in a fixed order (wave 0 .. 3), and
// the wave that finishes a fragment applies bias / residual / activation and stores 8- or 16-byte pieces of the pixels' rows.
#include <type_traits>

#include "i2r_common.h"
#include "i2r_conv.h"

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 b16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int DT>
__device__ __forceinline__ f32x4 mfma16_lp(u32x2 a, u32x2 b, f32x4 c) {
    if constexpr (DT == 1)
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(h16x4, a), __builtin_bit_cast(h16x4, b), c, 0, 0, 0);
}
template <int DT>
__device__ __forceinline__ u32x2 pack4_lp(f32x4 v) {
    if constexpr (DT == 1) {
        const b16x4 b = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
        return __builtin_bit_cast(u32x2, b);
    } else {
        const h16x4 h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
        return __builtin_bit_cast(u32x2, h);
    }
}

struct Lp1K {
    const void* x; const void* w; const float* bias; const void* res1; const void* res_post; void* out;
    int n_pix, n_tiles, kc, x_cs, out_cs, n_groups, act, out16;
};

// IN16: x is stored in the operand type (else fp32, packed on load)
template <int DT, int NF, int MT, bool IN16>
__global__ __launch_bounds__(256) void conv1x1_lp_k(const Lp1K p) {
    __shared__ f32x4 part[4][MT * NF][64];  // partial sums handed to the finishing wave (a wave's own slots stay unused)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, g = lane >> 4;
    const int grp = blockIdx.x % p.n_groups, tg = blockIdx.x / p.n_groups;  // output-fragment group, pixel-tile group
    const int tile0 = tg * MT, f0 = grp * NF;
    const unsigned xe = IN16 ? 2u : 4u;
    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(p.x, (unsigned)p.n_pix * p.x_cs * xe);
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, (unsigned)p.n_groups * NF * p.kc * 512);
    unsigned xo[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int q = (tile0 + mt) * 16 + li;
        xo[mt] = q < p.n_pix ? (unsigned)(q * p.x_cs + 4 * g) * xe : kOOB;
    }
    // this wave's share of the K steps
    const int c_lo = (p.kc * wave) >> 2, c_hi = (p.kc * (wave + 1)) >> 2;
    typedef typename std::conditional<IN16, u32x2, f32x4>::type xraw;
    struct Set { xraw x[MT]; u32x2 w[NF]; };
    auto fetch = [&](int c, Set& s) {
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) s.w[nf] = __builtin_amdgcn_raw_buffer_load_b64(rs_w, lane * 8, ((f0 + nf) * p.kc + c) * 512, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if constexpr (IN16) s.x[mt] = __builtin_amdgcn_raw_buffer_load_b64(rs_x, xo[mt], c * 32, 0);
            else s.x[mt] = buf_ld16(rs_x, xo[mt], c * 64);
        }
    };
    f32x4 acc[MT][NF];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) acc[mt][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto step = [&](const Set& s) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            u32x2 xb;
            if constexpr (IN16) xb = s.x[mt];
            else xb = pack4_lp<DT>(s.x[mt]);
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) acc[mt][nf] = mfma16_lp<DT>(s.w[nf], xb, acc[mt][nf]);
        }
    };
    {
        Set s0, s1, s2;  // step c uses set (c - c_lo) % 3; two steps are in flight ahead of it (look-aheads past c_hi re-read the last step)
        const int last = c_hi - 1;
        fetch(min(c_lo, last), s0);
        fetch(min(c_lo + 1, last), s1);
        int c = c_lo;
        for (; c + 3 <= c_hi; c += 3) {
            fetch(min(c + 2, last), s2);
            __builtin_amdgcn_sched_barrier(0);
            step(s0);
            fetch(min(c + 3, last), s0);
            __builtin_amdgcn_sched_barrier(0);
            step(s1);
            fetch(min(c + 4, last), s1);
            __builtin_amdgcn_sched_barrier(0);
            step(s2);
        }
        if (c < c_hi) {
            step(s0);
            if (c + 1 < c_hi) step(s1);
        }
    }
    // ---- the four partial sums meet: fragment (mt, nf) is finished by wave (mt * NF + nf) % 4, the other three hand theirs over ----
    const int slot = lane;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
            const int idx = mt * NF + nf;
            if ((idx & 3) != wave) part[wave][idx][slot] = acc[mt][nf];  // (wave-uniform)
        }
    __syncthreads();
    const bool o16 = p.out16 != 0;
    const unsigned oe = o16 ? 2u : 4u;
    const unsigned out_bytes = (unsigned)p.n_pix * p.out_cs * oe;
    const __amdgpu_buffer_rsrc_t rs_out = make_rsrc(p.out, out_bytes), rs_r1 = make_rsrc(p.res1, out_bytes), rs_rp = make_rsrc(p.res_post, out_bytes);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int q = (tile0 + mt) * 16 + li;
        const unsigned orow = q < p.n_pix ? (unsigned)(q * p.out_cs + 4 * g) * oe : kOOB;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
            const int idx = mt * NF + nf;
            if ((idx & 3) != wave) continue;  // (wave-uniform)
            f32x4 v = *reinterpret_cast<const f32x4*>(p.bias + 16 * (f0 + nf) + 4 * g);
#pragma unroll
            for (int w = 0; w < 4; ++w) v += (w == wave) ? acc[mt][nf] : part[w][idx][slot];  // fixed order: bit-identical replays
            const int soff = 16 * (f0 + nf) * (int)oe;
            if (p.res1) v += buf_ld_act4<DT>(rs_r1, orow + soff, o16);
            if (p.act == 1) {
                v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
            } else if (p.act == 2) {  // exact-erf GELU, as the implicit-GEMM kernel's epilogue (hrformer.py:1197)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = 0.5f * v[e] * (1.f + erff(v[e] * 0.70710678118654752f));
            }
            if (p.res_post) v += buf_ld_act4<DT>(rs_rp, orow + soff, o16);
            buf_st_act4<DT>(rs_out, orow + soff, v, o16);
        }
    }
}

typedef void (*lp1_fn)(const Lp1K);
template <int DT, int NF, bool IN16>
lp1_fn pick_mt(int mt) {
    if (mt == 1) return conv1x1_lp_k<DT, NF, 1, IN16>;
    if (mt == 2) return conv1x1_lp_k<DT, NF, 2, IN16>;
    return nullptr;
}
template <int DT>
lp1_fn pick(int nf, int mt, bool in16) {
    if (nf == 3) return in16 ? pick_mt<DT, 3, true>(mt) : pick_mt<DT, 3, false>(mt);
    if (nf == 4) return in16 ? pick_mt<DT, 4, true>(mt) : pick_mt<DT, 4, false>(mt);
    if (nf == 5) return in16 ? pick_mt<DT, 5, true>(mt) : pick_mt<DT, 5, false>(mt);
    if (nf == 6) return in16 ? pick_mt<DT, 6, true>(mt) : pick_mt<DT, 6, false>(mt);
    return nullptr;
}

}  // namespace

extern "C" int i2r_conv1x1_lp(const i2r_conv1x1_lp_args* a, void* stream) {
    I2R_CHECK_ARG(a && a->x && a->w && a->bias && a->out, "i2r_conv1x1_lp: null pointer");
    I2R_CHECK_ARG(a->dtype == 1 || a->dtype == 2, "i2r_conv1x1_lp: dtype %d (1 bf16, 2 f16 operands; fp32 convs go through i2r_conv)", a->dtype);
    I2R_CHECK_ARG(a->n_pix > 0 && a->cin_pad >= 64 && a->cin_pad % 16 == 0 && a->cout_pad >= 16 && a->cout_pad % 16 == 0 && a->x_cs >= a->cin_pad &&
                      a->out_cs >= a->cout_pad && a->x_cs % 4 == 0 && a->out_cs % 4 == 0 && a->act >= 0 && a->act <= 2,
                  "i2r_conv1x1_lp: n_pix=%d cin_pad=%d cout_pad=%d x_cs=%d out_cs=%d act=%d", a->n_pix, a->cin_pad, a->cout_pad, a->x_cs, a->out_cs, a->act);
    I2R_CHECK_ARG(a->out != a->x, "i2r_conv1x1_lp: out aliases x");  // (res1 / res_post may alias out: a lane reads its piece, then writes it)
    const int64_t cs_max = a->out_cs > a->x_cs ? a->out_cs : a->x_cs;
    I2R_CHECK_ARG((int64_t)a->n_pix * cs_max * 4 < (int64_t)kOOB, "i2r_conv1x1_lp: tensors must stay below 2 GiB (32-bit buffer offsets)");
    const int n_frag = a->cout_pad / 16;
    const int nf = n_frag % 5 == 0 ? 5 : n_frag % 6 == 0 ? 6 : n_frag % 4 == 0 ? 4 : n_frag % 3 == 0 ? 3 : 0;
    I2R_CHECK_ARG(nf != 0, "i2r_conv1x1_lp: cout_pad / 16 = %d is no multiple of 3, 4, 5 or 6 (use i2r_conv)", n_frag);
    Lp1K k;
    k.x = a->x; k.w = a->w; k.bias = a->bias; k.res1 = a->res1; k.res_post = a->res_post; k.out = a->out;
    k.n_pix = a->n_pix; k.n_tiles = (a->n_pix + 15) / 16; k.kc = a->cin_pad / 16; k.x_cs = a->x_cs; k.out_cs = a->out_cs;
    k.n_groups = n_frag / nf; k.act = a->act; k.out16 = a->out_16;
    // two pixel tiles per workgroup (each weight fragment feeds two MFMAs) when that still leaves a workgroup per CU
    int mt = a->mt;
    if (mt == 0) mt = ((k.n_tiles + 1) / 2) * k.n_groups >= 256 ? 2 : 1;
    lp1_fn fn = a->dtype == 1 ? pick<1>(nf, mt, a->in_16 != 0) : pick<2>(nf, mt, a->in_16 != 0);
    I2R_CHECK_ARG(fn != nullptr, "i2r_conv1x1_lp: mt=%d (1, 2)", mt);
    const long long nblk = (long long)((k.n_tiles + mt - 1) / mt) * k.n_groups;
    I2R_CHECK_ARG(nblk < (1ll << 31), "i2r_conv1x1_lp: grid");
    hipLaunchKernelGGL(fn, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, k);
    I2R_CHECK_LAUNCH("i2r_conv1x1_lp");
    return I2R_OK;
}
