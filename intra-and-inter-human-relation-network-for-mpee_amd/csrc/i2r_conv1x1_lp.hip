// 1x1 convolution over a SMALL number of NHWC pixel rows with 16-bit MFMA operands, straight from global memory (gfx950).
//
//   out = act(W x + b [+ res1] [+ res2]) [+ res_post]          x: [n_pix, x_cs] fp32 or 16 bit, out / res: [n_pix, out_cs] fp32 or 16 bit
//
// The low-resolution branches of HRFormer-B (C = 312 @ 16x12, C = 624 @ 8x6: 3072 / 768 pixels at 16 crops) run their transformer blocks
// as single launches of LayerNorm, q|k|v projection, window attention, out projection, fc1, depth-wise conv and fc2 on the critical
// stream lane (reference lib/models/hrformer.py:1230-1240, MlpDWBN :1094-1119, InterlacedPoolAttention :1164-1180), and so do the 1x1
// convolutions of the fuse layers (:1629-1704).  On the implicit-GEMM kernel such a conv is a few hundred workgroups that stage the
// pixels' channels through LDS in 64-channel chunks, one barrier pair per chunk, for 8 MFMAs each: 15-36 us for 0.3-2.4 GFLOP, all of
// it latency.  Here the GEMM is computed TRANSPOSED (Y^T = W X^T, as in the encoder kernels): the weight fragment is the MFMA A
// operand (v_mfma_f32_16x16x32; fragment-packed by the host: one 64-lane 16-byte load = 1 KB contiguous), a pixel's 8 consecutive
// channels the B operand (one 16-byte load from its row, or 32 bytes of fp32 packed on the fly) -- no LDS in the K loop, no barrier.
// (The loop is bound by the texture addresser -- NF + MT load instructions per NF x MT MFMAs -- which is why the 32-deep MFMA with its
// 16-byte operands is used: the 16-deep one needs twice the load instructions per FLOP and measured 1.5x slower.)  A workgroup owns MT
// 16-pixel tiles x NF 16-channel output fragments and its four waves SPLIT K (cin / 32 steps): that is what
// puts enough waves on the chip for these shapes; the four partial sums meet once through LDS in a fixed order (wave 0 .. 3), and
// the wave that finishes a fragment applies bias / residual / activation and stores 8- or 16-byte pieces of the pixels' rows.
#include "i2r_common.h"
#include "i2r_conv.h"

namespace {

template <int DT>
__device__ __forceinline__ f32x4 pack8_lp(f32x4 lo, f32x4 hi) {  // 8 fp32 -> 8 16-bit values in order
    if constexpr (DT == 1) {
        const bf16x8 v = {(__bf16)lo[0], (__bf16)lo[1], (__bf16)lo[2], (__bf16)lo[3], (__bf16)hi[0], (__bf16)hi[1], (__bf16)hi[2], (__bf16)hi[3]};
        return __builtin_bit_cast(f32x4, v);
    } else {
        const f16x8 v = {(_Float16)lo[0], (_Float16)lo[1], (_Float16)lo[2], (_Float16)lo[3], (_Float16)hi[0], (_Float16)hi[1], (_Float16)hi[2], (_Float16)hi[3]};
        return __builtin_bit_cast(f32x4, v);
    }
}

struct Lp1K {
    const void* x; const void* w; const float* bias; const void* res1; const void* res2; const void* res_post; void* out;
    int n_pix, n_tiles, kc, k_tail, x_cs, out_cs, n_groups, act, out16;  // kc: 32-channel steps (the last one half-filled when k_tail)
};

// IN16: x is stored in the operand type (else fp32, packed on load)
template <int DT, int NF, int MT, bool IN16>
__global__ __launch_bounds__(256) void conv1x1_lp_k(const Lp1K p) {
    __shared__ f32x4 part[MT * NF][3][64];  // partial sums handed to the finishing wave: slot (wave - owner - 1) mod 4 of the fragment
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, g = lane >> 4;
    const int grp = blockIdx.x % p.n_groups, tg = blockIdx.x / p.n_groups;  // output-fragment group, pixel-tile group
    const int tile0 = tg * MT, f0 = grp * NF;
    const unsigned xe = IN16 ? 2u : 4u;
    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(p.x, (unsigned)p.n_pix * p.x_cs * xe);
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, (unsigned)p.n_groups * NF * p.kc * 1024);
    // byte offset of this lane's 8 channels (8 g ..) of a 32-channel step in its pixels' rows.  When the channel count is an odd multiple
    // of 16 the last step is half a step: the lanes of its upper half (g >= 2) read nothing (zeros; the weights are zero-padded too)
    unsigned xo[MT], xo_last[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int q = (tile0 + mt) * 16 + li;
        xo[mt] = q < p.n_pix ? (unsigned)(q * p.x_cs + 8 * g) * xe : kOOB;
        xo_last[mt] = (p.k_tail && g >= 2) ? kOOB : xo[mt];
    }
    // this wave's share of the K steps
    const int c_lo = (p.kc * wave) >> 2, c_hi = (p.kc * (wave + 1)) >> 2;
    struct Set { f32x4 x[MT][IN16 ? 1 : 2]; f32x4 w[NF]; };
    auto fetch = [&](int c, Set& s) {
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) s.w[nf] = buf_ld16(rs_w, lane * 16, ((f0 + nf) * p.kc + c) * 1024);
        const bool tail = c == p.kc - 1;  // (wave-uniform)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const unsigned o = tail ? xo_last[mt] : xo[mt];
            if constexpr (IN16) s.x[mt][0] = buf_ld16(rs_x, o, c * 64);
            else {
                s.x[mt][0] = buf_ld16(rs_x, o, c * 128);
                s.x[mt][1] = buf_ld16(rs_x, o, c * 128 + 16);
            }
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
            f32x4 xb;
            if constexpr (IN16) xb = s.x[mt][0];
            else xb = pack8_lp<DT>(s.x[mt][0], s.x[mt][1]);
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) acc[mt][nf] = mfma32_lp<DT>(s.w[nf], xb, acc[mt][nf]);
        }
    };
    // No software look-ahead: rings of 2, 3 and 6 operand sets were measured (configs 4 / 5, same box: 3.64 / 3.86 / 4.24 ms per forward
    // against 3.54 ms like this) -- these launches share the chip with the other stream lanes' kernels, and what counts is how many
    // waves fit next to them (registers), the latency of a step is covered by the neighbours.
    for (int c = c_lo; c < c_hi; ++c) {
        Set s;
        fetch(c, s);
        step(s);
    }
    // ---- the four partial sums meet: fragment (mt, nf) is finished by wave (mt * NF + nf) % 4, the other three hand theirs over ----
    const int slot = lane;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
            const int idx = mt * NF + nf;
            if ((idx & 3) != wave) part[idx][(wave - (idx & 3) + 3) & 3][slot] = acc[mt][nf];  // (wave-uniform)
        }
    __syncthreads();
    const bool o16 = p.out16 != 0;
    const unsigned oe = o16 ? 2u : 4u;
    const unsigned out_bytes = (unsigned)p.n_pix * p.out_cs * oe;
    const __amdgpu_buffer_rsrc_t rs_out = make_rsrc(p.out, out_bytes), rs_r1 = make_rsrc(p.res1, out_bytes), rs_r2 = make_rsrc(p.res2, out_bytes), rs_rp = make_rsrc(p.res_post, out_bytes);
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
            for (int w = 0; w < 4; ++w) v += (w == wave) ? acc[mt][nf] : part[idx][w == wave ? 0 : (w - wave + 3) & 3][slot];  // fixed order 0 .. 3: bit-identical replays
            const int soff = 16 * (f0 + nf) * (int)oe;
            if (p.res1) v += buf_ld_act4<DT>(rs_r1, orow + soff, o16);
            if (p.res2) v += buf_ld_act4<DT>(rs_r2, orow + soff, o16);  // ((conv + res1) + res2: the order of the implicit-GEMM epilogue)
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
    if (mt == 4) return conv1x1_lp_k<DT, NF, 4, IN16>;
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
    k.x = a->x; k.w = a->w; k.bias = a->bias; k.res1 = a->res1; k.res2 = a->res2; k.res_post = a->res_post; k.out = a->out;
    k.n_pix = a->n_pix; k.n_tiles = (a->n_pix + 15) / 16; k.kc = (a->cin_pad + 31) / 32; k.k_tail = (a->cin_pad & 16) != 0; k.x_cs = a->x_cs; k.out_cs = a->out_cs;
    k.n_groups = n_frag / nf; k.act = a->act; k.out16 = a->out_16;
    // two pixel tiles per workgroup (each weight fragment feeds two MFMAs) when that still leaves a workgroup per CU
    int mt = a->mt;
    if (mt == 0) mt = ((k.n_tiles + 1) / 2) * k.n_groups >= 256 ? 2 : 1;
    lp1_fn fn = a->dtype == 1 ? pick<1>(nf, mt, a->in_16 != 0) : pick<2>(nf, mt, a->in_16 != 0);
    I2R_CHECK_ARG(fn != nullptr, "i2r_conv1x1_lp: mt=%d (1, 2, 4)", mt);
    const long long nblk = (long long)((k.n_tiles + mt - 1) / mt) * k.n_groups;
    I2R_CHECK_ARG(nblk < (1ll << 31), "i2r_conv1x1_lp: grid");
    i2r_launch(fn, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, k);
    I2R_CHECK_LAUNCH("i2r_conv1x1_lp");
    return I2R_OK;
}
