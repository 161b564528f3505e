// HRFormer-B transformer block, attention half, FUSED for the 16-bit modes (BASELINE configs 4-5):
//     x1 = x + out_proj( window_attention( q|k|v_proj( LayerNorm1(x) ) ) )          (reference lib/models/hrformer.py:1230-1236,
//     InterlacedPoolAttention :1164-1180, PadBlock :937-966, LocalPermuteModule :969-1001, MHA_ :692-935)
// in ONE launch per block instead of LayerNorm + q|k|v conv + window attention + out-proj conv (and without the [n, h, w, 3C] q|k|v
// tensor ever reaching HBM).  On the two high-resolution branches (C = 78 / 156) those four kernels are a few microseconds of
// arithmetic each, i.e. launch- and latency-bound; a 7x7 window is an independent problem of 49 tokens, so one workgroup does it all.
//
// One workgroup = one window, 4 waves, wave w owns window tokens 16w .. 16w+15 (49 real tokens + 15 padding rows).  Every GEMM is
// computed TRANSPOSED on v_mfma_f32_16x16x16_{bf16,f16} (A = 16 rows x 16 k: lane (i = l&15, g = l>>4) supplies A[i][4g..4g+3];
// B = 16 k x 16 columns: lane supplies B[4g..4g+3][j = l&15]; D: lane holds D[4g + r][l&15], r < 4):  Y^T = W . X^T with the weight
// fragment as A and the token columns as B.  The D fragment (features 4g+r of token l&15) packed to 16 bit IS the B operand of the
// next GEMM, so LayerNorm -> q/k/v -> S^T = K Q^T -> softmax -> O^T = V^T P^T -> out-proj chain through registers; only K rows and
// V^T (needed by all four waves) go through LDS, double-buffered per head (one barrier per head).
//   * LayerNorm in fp32 (eps 1e-6); the window's zero padding is applied AFTER it (hrformer.py:947-956): tokens outside the map are
//     exact zeros, so their q/k/v equal the projection biases and they take part as ordinary keys; rows 49..63 are masked keys.
//   * head_dim 39 is padded to 48 = three 16-wide k-steps (zero weight rows / columns); q carries head_dim^-0.5 * log2(e) (folded
//     by the host), the softmax runs in base 2.
//   * fp32 residual and output (the HRFormer tower keeps its maps in fp32); operands are bf16 / f16, accumulation fp32.
#include "i2r_common.h"

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 b16x4 __attribute__((ext_vector_type(4)));

template <int DT>
__device__ __forceinline__ f32x4 mfma16(uint2 a, uint2 b, f32x4 c) {  // D = A(16x16) B(16x16) + C
    if constexpr (DT == 1)
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(h16x4, a), __builtin_bit_cast(h16x4, b), c, 0, 0, 0);
}
template <int DT>
__device__ __forceinline__ uint2 pack4(f32x4 v) {
    if constexpr (DT == 1) {
        const b16x4 b = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
        return __builtin_bit_cast(uint2, b);
    } else {
        const h16x4 h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
        return __builtin_bit_cast(uint2, h);
    }
}
template <int DT>
__device__ __forceinline__ unsigned short pack1(float v) {
    if constexpr (DT == 1) return __builtin_bit_cast(unsigned short, (__bf16)v);
    else return __builtin_bit_cast(unsigned short, (_Float16)v);
}
__device__ __forceinline__ float xsum4(float v) {  // over the 4 lanes that share l & 15
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}
__device__ __forceinline__ float xmax4(float v) {
    v = fmaxf(v, __shfl_xor(v, 16));
    v = fmaxf(v, __shfl_xor(v, 32));
    return v;
}

struct AttnK {
    const float* x; float* out;
    const float* ln_w; const float* ln_b;
    const uint2* wqkv; const float* bqkv;   // [head][q,k,v][3 dim blocks][CB][64 lanes] fragments; [head][3][48] biases
    const uint2* wo; const float* bo;       // [CB out blocks][head][3 dim blocks][64 lanes]; [cs]
    int n_img, h, w, c, nwy, nwx, pad_top, pad_left;
    float eps;
};

constexpr int KS = 52, VS = 68;  // LDS row strides (16-bit elements): K rows [key][48 dims], V^T rows [dim][64 keys]

// (waves per SIMD: C = 78 runs 4 workgroups per CU -- a bound of 5 spills 60 VGPRs, 32 -> 49 us; C = 156 with the full register budget of 2: 48 -> 40 us)
template <int DT, int CB, int HEADS>
__global__ __launch_bounds__(256, CB == 5 ? 4 : 2) void hrt_attn_block_k(const AttnK p) {
    constexpr int cs = CB * 16;
    __shared__ __attribute__((aligned(16))) unsigned short Kb[2][64 * KS];
    __shared__ __attribute__((aligned(16))) unsigned short Vt[2][48 * VS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
    int bid = blockIdx.x;
    const int wx = bid % p.nwx; bid /= p.nwx;
    const int wy = bid % p.nwy;
    const int img = bid / p.nwy;
    // this lane's token (column li of the wave's 16-token fragment)
    const int t = wave * 16 + li;
    const int ty = t / 7, tx = t - ty * 7;
    const int y = wy * 7 + ty - p.pad_top, x = wx * 7 + tx - p.pad_left;
    const bool inmap = t < 49 && y >= 0 && y < p.h && x >= 0 && x < p.w;
    const size_t row = (((size_t)img * p.h + (inmap ? y : 0)) * p.w + (inmap ? x : 0)) * cs;

    // ---- LayerNorm 1 of the token (features 16c + 4g + r live in xr[c][r]); B operand of the projections ----
    f32x4 xr[CB];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CB; ++c) {
        xr[c] = inmap ? *reinterpret_cast<const f32x4*>(p.x + row + 16 * c + 4 * g) : (f32x4){0.f, 0.f, 0.f, 0.f};
        s += (xr[c][0] + xr[c][1]) + (xr[c][2] + xr[c][3]);  // pad channels are exact zeros
    }
    const float mean = xsum4(s) / (float)p.c;
    float q2 = 0.f;
#pragma unroll
    for (int c = 0; c < CB; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float d = (16 * c + 4 * g + r < p.c) ? xr[c][r] - mean : 0.f;
            q2 += d * d;
        }
    const float rstd = rsqrtf(xsum4(q2) / (float)p.c + p.eps);
    uint2 xn[CB];
#pragma unroll
    for (int c = 0; c < CB; ++c) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(p.ln_w + 16 * c + 4 * g), bv = *reinterpret_cast<const f32x4*>(p.ln_b + 16 * c + 4 * g);
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = inmap ? (xr[c][r] - mean) * rstd * wv[r] + bv[r] : 0.f;  // (padded ln_w = ln_b = 0 -> 0)
        xn[c] = pack4<DT>(v);
    }

    // Weight fragments stream from L2 (every workgroup reads the same ones); a dependent load -> MFMA chain per 16-feature block
    // costs one L2 latency each (measured: 30 us per window), so the fragments of one PART (q, k or v of a head: 3 blocks x CB
    // fragments) are fetched as a batch, one part ahead of the MFMAs that consume them, in two alternating register sets.
    uint2 wf[2][3][CB];
    auto fetch_part = [&](uint2(&dst)[3][CB], int hh, int part) {
        const uint2* src = p.wqkv + ((size_t)(hh * 3 + part) * 3) * CB * 64 + lane;
#pragma unroll
        for (int db = 0; db < 3; ++db)
#pragma unroll
            for (int c = 0; c < CB; ++c) dst[db][c] = src[(db * CB + c) * 64];
    };
    // one 16-feature block of a projection for this wave's tokens:  bias + sum_c W[blk][c] . xn[c]
    auto project = [&](const uint2(&wfr)[CB], const float* bias) -> f32x4 {
        f32x4 acc = *reinterpret_cast<const f32x4*>(bias + 4 * g);  // feature 4g + r of the block
#pragma unroll
        for (int c = 0; c < CB; ++c) acc = mfma16<DT>(wfr[c], xn[c], acc);
        return acc;
    };

    uint2 oB[HEADS][3];  // attention output O^T of every head, packed as the out-proj B operand
    fetch_part(wf[0], 0, 0);
#pragma unroll
    for (int hh = 0; hh < HEADS; ++hh) {
        unsigned short* const kb = Kb[hh & 1];
        unsigned short* const vt = Vt[hh & 1];
        const float* const bh = p.bqkv + hh * 9 * 16;
        const int s0 = (3 * hh) & 1;  // register set holding this head's q fragments (compile-time after unrolling)
        uint2 qB[3];
        fetch_part(wf[s0 ^ 1], hh, 1);
#pragma unroll
        for (int db = 0; db < 3; ++db) qB[db] = pack4<DT>(project(wf[s0][db], bh + (0 * 3 + db) * 16));
        fetch_part(wf[s0], hh, 2);
#pragma unroll
        for (int db = 0; db < 3; ++db) {
            const f32x4 kk = project(wf[s0 ^ 1][db], bh + (1 * 3 + db) * 16);
            *reinterpret_cast<uint2*>(kb + t * KS + 16 * db + 4 * g) = pack4<DT>(kk);  // K[key t][dims 16db + 4g ..]
        }
        if (hh + 1 < HEADS) fetch_part(wf[s0 ^ 1], hh + 1, 0);
#pragma unroll
        for (int db = 0; db < 3; ++db) {
            const f32x4 vv = project(wf[s0][db], bh + (2 * 3 + db) * 16);
#pragma unroll
            for (int r = 0; r < 4; ++r) vt[(16 * db + 4 * g + r) * VS + t] = pack1<DT>(vv[r]);  // V^T[dim][key t]
        }
        __syncthreads();
        // ---- S^T[key][query] for the 64 keys, softmax over the 49 real ones (base 2) ----
        f32x4 st[4];
        float mx = -__builtin_inff();
#pragma unroll
        for (int kf = 0; kf < 4; ++kf) {
            st[kf] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int db = 0; db < 3; ++db)
                st[kf] = mfma16<DT>(*reinterpret_cast<const uint2*>(kb + (16 * kf + li) * KS + 16 * db + 4 * g), qB[db], st[kf]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (16 * kf + 4 * g + r >= 49) st[kf][r] = -__builtin_inff();
                mx = fmaxf(mx, st[kf][r]);
            }
        }
        mx = xmax4(mx);
        float sum = 0.f;
        uint2 pB[4];
#pragma unroll
        for (int kf = 0; kf < 4; ++kf) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                st[kf][r] = __builtin_amdgcn_exp2f(st[kf][r] - mx);
                sum += st[kf][r];
            }
            pB[kf] = pack4<DT>(st[kf]);
        }
        const float inv = 1.f / xsum4(sum);
        // ---- O^T[dim][query] = V^T P^T ----
#pragma unroll
        for (int db = 0; db < 3; ++db) {
            f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
                o = mfma16<DT>(*reinterpret_cast<const uint2*>(vt + (16 * db + li) * VS + 16 * kf + 4 * g), pB[kf], o);
            oB[hh][db] = pack4<DT>(o * inv);
        }
        // (no second barrier: the next head writes the other K / V^T buffer, and the one after that is separated by the next barrier)
    }

    // ---- out-proj + bias + residual; only tokens inside the map are written; fragments of the next output block prefetched ----
    uint2 wo[2][HEADS * 3];
    auto fetch_out = [&](uint2(&dst)[HEADS * 3], int ob) {
        const uint2* src = p.wo + (size_t)ob * HEADS * 3 * 64 + lane;
#pragma unroll
        for (int i = 0; i < HEADS * 3; ++i) dst[i] = src[i * 64];
    };
    fetch_out(wo[0], 0);
#pragma unroll
    for (int ob = 0; ob < CB; ++ob) {
        if (ob + 1 < CB) fetch_out(wo[(ob + 1) & 1], ob + 1);
        f32x4 acc = *reinterpret_cast<const f32x4*>(p.bo + 16 * ob + 4 * g);
#pragma unroll
        for (int hh = 0; hh < HEADS; ++hh)
#pragma unroll
            for (int db = 0; db < 3; ++db) acc = mfma16<DT>(wo[ob & 1][hh * 3 + db], oB[hh][db], acc);
        if (inmap) *reinterpret_cast<f32x4*>(p.out + row + 16 * ob + 4 * g) = acc + xr[ob];
    }
}


// =====================================================================================================================
// MLP half of the block, fused:  x2 = x1 + GELU(BN3(fc2( GELU(BN2(DW3x3( GELU(BN1(fc1( LayerNorm2(x1) ))) ))) )))
// (reference hrformer.py:1237 + MlpDWBN.forward :1094-1119; the BatchNorms are folded into the conv weights by the host)
// in ONE launch per block instead of LayerNorm + fc1 conv + depth-wise conv + fc2 conv, and without the 4C-wide hidden tensor
// (the largest map of the block) ever reaching HBM.
//
// One workgroup = one 8x8 tile of output pixels (4 waves, one 16-pixel fragment each) + its halo of 1: 10 x 10 = 100 pixels padded
// to 7 fragments for the fc1 phase.  LayerNorm 2 runs once (each wave two fragments), the packed pixel columns go through LDS to all
// four waves, which keep them in registers.  The hidden dimension (padded to 64 CB = 4 CB blocks of 16) is walked in chunks of 4 blocks:
//   A  fc1: wave w computes hidden block h0 + w of ALL 7 halo fragments (even split; one set of weight fragments per wave and chunk):
//      transposed GEMM as in the attention kernel (A = weight fragment, B = pixel columns), + bias, GELU, fp32 to LDS as
//      H[halo pixel][chunk channel]; pixels outside the image give exact zeros (the depth-wise conv pads the HIDDEN tensor with zeros):
//      their columns and their bias are zeroed and GELU(0) = 0;
//   B  each lane owns one output pixel and 4 hidden channels per block: 9 taps x (16-byte LDS read of H, 16-byte LDS read of the tap
//      weights, two packed FMAs) -> + bias -> GELU -> packed: that IS the B operand (k = hidden channel) of fc2, accumulated over the
//      chunks into CB output fragments per wave.  The depth-wise weights of the whole hidden dimension sit in LDS (loaded once).
// Epilogue: + bias, GELU, + x1 (fp32 residual), fp32 store.  x / out are the fp32 residual stream [n, h, w, cs]; operands bf16 / f16.
// The kernel is VALU-bound (three GELUs per hidden element against 2 x 16 MFMA k-steps): see gelu4 below.
struct MlpK {
    const float* x; float* out;
    const float* ln_w; const float* ln_b;
    const uint2* w1; const float* b1;     // fc1 (+BN1): fragments [hidden block][CB][64 lanes]; bias [hidden_pad]
    const float* wdw; const float* bdw;   // depth-wise 3x3 (+BN2): [9][hidden_pad] tap-major; bias [hidden_pad]
    const uint2* w2; const float* b2;     // fc2 (+BN3): fragments [CB out blocks][hidden blocks][64 lanes]; bias [cs]
    int n_img, h, w, c, tiles_y, tiles_x;
    float eps;
};

// exact-erf GELU (nn.GELU, hrformer.py:1197) with ONE transcendental:  GELU(x) = x/2 + |x| (1/2 - Phi(-|x|)), and the normal tail
// Phi(-a) = erfc(a / sqrt 2) / 2 = 2^-Q(a) with Q a degree-5 polynomial, Q(0) = 1 (-log2 of the tail is smooth and grows like a^2;
// coefficients: minimax fit of the GELU error, tools/fit_gelu.py).  Q is evaluated by even and odd parts in w = x^2 -- packed FMAs on
// pairs, |x| used once -- with w clamped at 16^2 (the tail is below 2^-390 from there on, and huge inputs cannot overflow it).
// |error| < 1.2e-6 on the GELU value in fp32 arithmetic: three orders below one bf16 / f16 rounding, at ~9 issue slots + one exp2
// instead of the rational erf's rcp + exp2 + ~17.  (The fp32 parity kernels keep libm's erff.)
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct GeluC {  // the coefficients as opaque SGPR pairs: literals would make the compiler pick scalar fmaak / fmamk over the packed FMAs
    f32x2 c1, c2, c3, c4, c5, one, half;
    __device__ __forceinline__ GeluC() {
        auto splat = [](float v) { f32x2 r = {v, v}; asm("" : "+s"(r)); return r; };
        c1 = splat(1.15100077e+00f); c2 = splat(4.59594261e-01f); c3 = splat(5.21491148e-02f); c4 = splat(-7.20010049e-03f);
        c5 = splat(4.88351593e-04f); one = splat(1.f); half = splat(0.5f);
    }
};
__device__ __forceinline__ f32x2 gelu2(f32x2 x, const GeluC& k) {
    f32x2 w = x * x;
    w[0] = fminf(w[0], 256.f);
    w[1] = fminf(w[1], 256.f);
    const f32x2 ev = (w * k.c4 + k.c2) * w + k.one;
    const f32x2 od = (w * k.c5 + k.c3) * w + k.c1;
    f32x2 t;
    t[0] = __builtin_amdgcn_exp2f(-__builtin_fmaf(__builtin_fabsf(x[0]), od[0], ev[0]));
    t[1] = __builtin_amdgcn_exp2f(-__builtin_fmaf(__builtin_fabsf(x[1]), od[1], ev[1]));
    t = k.half - t;
    const f32x2 h = x * k.half;
    return (f32x2){__builtin_fmaf(__builtin_fabsf(x[0]), t[0], h[0]), __builtin_fmaf(__builtin_fabsf(x[1]), t[1], h[1])};
}
__device__ __forceinline__ f32x4 gelu4(f32x4 x, const GeluC& k) {
    const f32x2 lo = gelu2(x.xy, k), hi = gelu2(x.zw, k);
    return (f32x4){lo[0], lo[1], hi[0], hi[1]};
}

template <int DT, int CB>
__global__ __launch_bounds__(256) void hrt_mlp_block_k(const MlpK p) {
    // 8x8 output pixels per workgroup: 10 x 10 = 100 halo pixels -> 7 fragments.  (8x16 tiles -- 1.4x instead of 1.75x halo recompute --
    // measured 1.9x SLOWER at 16 crops: 384 workgroups do not fill 256 CUs.)
    constexpr int cs = CB * 16, TH = 8, TW = 8, HW = TW + 2, NHP = (TH + 2) * HW, NF = (NHP + 15) / 16;
    constexpr int HBT = 4 * CB, HID = HBT * 16;  // hidden blocks / padded hidden width (the host pads 4C to a multiple of 64)
    constexpr int HC = 4, HS = HC * 16 + 4;      // hidden blocks per chunk = waves; LDS row stride in floats (+4: bank spread, 16-byte rows)
    constexpr int FPW = (NF + 3) / 4;            // halo fragments LayerNorm-ed per wave (2)
    static_assert(TH * TW == 64 && NF == 7, "one output fragment per wave");
    constexpr int H_BYTES = NF * 16 * HS * 4, X_BYTES = NF * CB * 64 * 8;
    // H (per chunk) and the packed LayerNorm-ed pixel columns (read once, before the first chunk) share one region
    __shared__ __attribute__((aligned(16))) unsigned char smem[H_BYTES > X_BYTES ? H_BYTES : X_BYTES];
    __shared__ __attribute__((aligned(16))) float Wd[10 * HID];  // depth-wise taps [9][HID] + bias [HID]
    __shared__ float hin_s[NF * 16];                             // 1 = halo pixel inside the image
    float* const Hs = reinterpret_cast<float*>(smem);
    uint2* const Xs = reinterpret_cast<uint2*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
    int bid = blockIdx.x;
    const int tx = bid % p.tiles_x; bid /= p.tiles_x;
    const int ty = bid % p.tiles_y;
    const int img = bid / p.tiles_y;
    const int y0 = ty * TH - 1, x0 = tx * TW - 1;  // image coordinate of halo pixel (0, 0)

    // depth-wise weights + bias of the whole hidden dimension -> LDS (consumed after the barriers below)
    for (int i = tid; i < 10 * HID / 4; i += 256) {
        const f32x4 v = i < 9 * HID / 4 ? reinterpret_cast<const f32x4*>(p.wdw)[i] : reinterpret_cast<const f32x4*>(p.bdw)[i - 9 * HID / 4];
        reinterpret_cast<f32x4*>(Wd)[i] = v;
    }
    // ---- LayerNorm 2 of halo fragments wave, wave + 4: packed B operands -> Xs[fragment][c][lane] ----
#pragma unroll
    for (int f = 0; f < FPW; ++f) {
        const int frag = wave + 4 * f;
        if (frag >= NF) break;  // (wave-uniform)
        const int hp = frag * 16 + li;  // halo pixel index (rows of HW)
        const int py = hp / HW, px = hp - py * HW;
        const int y = y0 + py, x = x0 + px;
        const bool in = hp < NHP && y >= 0 && y < p.h && x >= 0 && x < p.w;
        const float* row = p.x + (((size_t)img * p.h + (in ? y : 0)) * p.w + (in ? x : 0)) * cs;
        f32x4 xr[CB];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            xr[c] = *reinterpret_cast<const f32x4*>(row + 16 * c + 4 * g);
            s += (xr[c][0] + xr[c][1]) + (xr[c][2] + xr[c][3]);
        }
        const float mean = xsum4(s) / (float)p.c;
        float q2 = 0.f;
#pragma unroll
        for (int c = 0; c < CB; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d = (16 * c + 4 * g + r < p.c) ? xr[c][r] - mean : 0.f;
                q2 += d * d;
            }
        const float rstd = in ? rsqrtf(xsum4(q2) / (float)p.c + p.eps) : 0.f;  // outside the image: zero columns
        const float sh = in ? 1.f : 0.f;
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(p.ln_w + 16 * c + 4 * g), bv = *reinterpret_cast<const f32x4*>(p.ln_b + 16 * c + 4 * g);
            const f32x4 v = (xr[c] - mean) * rstd * wv + bv * sh;
            Xs[(frag * CB + c) * 64 + lane] = pack4<DT>(v);
        }
        if (g == 0) hin_s[hp] = sh;
    }
    __syncthreads();
    uint2 xn[NF][CB];
    float hinf[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#pragma unroll
        for (int c = 0; c < CB; ++c) xn[f][c] = Xs[(f * CB + c) * 64 + lane];
        hinf[f] = hin_s[f * 16 + li];
    }

    // this lane's OUTPUT pixel (fragment `wave` of the 8x8 tile) and its 3x3 neighbourhood in the halo grid
    const int op = wave * 16 + li;
    const int oy = op / TW, ox = op - oy * TW;
    const int gy = ty * TH + oy, gx = tx * TW + ox;
    const bool oin = gy < p.h && gx < p.w;
    const float* const hrd = Hs + (oy * HW + ox) * HS + 4 * g;  // top-left tap, this lane's 4 channels of chunk block 0
    const float* const wrd = Wd + 4 * g;
    float* const hwr = Hs + li * HS + wave * 16 + 4 * g;         // fc1 result of halo fragment 0, this wave's block of the chunk

    const GeluC gk;
    f32x4 acc[CB];
#pragma unroll
    for (int ob = 0; ob < CB; ++ob) acc[ob] = (f32x4){0.f, 0.f, 0.f, 0.f};
    uint2 w1f[CB], w2f[2][CB];
    f32x4 b1v;
    auto fetch1 = [&](int hb) {
#pragma unroll
        for (int c = 0; c < CB; ++c) w1f[c] = p.w1[(hb * CB + c) * 64 + lane];
        b1v = *reinterpret_cast<const f32x4*>(p.b1 + hb * 16 + 4 * g);
    };
    auto fetch2 = [&](uint2(&dst)[CB], int hb) {
#pragma unroll
        for (int ob = 0; ob < CB; ++ob) dst[ob] = p.w2[(ob * HBT + hb) * 64 + lane];
    };
    fetch1(wave);
    fetch2(w2f[0], 0);
    __syncthreads();  // everyone holds its copy of the pixel columns: the region becomes H

    for (int h0 = 0; h0 < HBT; h0 += HC) {
        // ---- A: fc1, hidden block h0 + wave, all halo fragments -> GELU -> LDS ----
        if (h0 != 0) __syncthreads();  // everyone is done reading the previous chunk's H
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            f32x4 a = b1v * hinf[f];
#pragma unroll
            for (int c = 0; c < CB; ++c) a = mfma16<DT>(w1f[c], xn[f][c], a);
            *reinterpret_cast<f32x4*>(hwr + f * 16 * HS) = gelu4(a, gk);
        }
        if (h0 + HC < HBT) fetch1(h0 + HC + wave);  // next chunk's fragments: in flight under phase B
        __syncthreads();
        // ---- B: depth-wise 3x3 + GELU on this lane's pixel x 4 channels per hidden block, then the fc2 partial products ----
#pragma unroll
        for (int hb = 0; hb < HC; ++hb) {
            if (hb + 1 < HC) fetch2(w2f[(hb + 1) & 1], h0 + hb + 1);
            else if (h0 + HC < HBT) fetch2(w2f[0], h0 + HC);
            const float* const wch = wrd + (h0 + hb) * 16;
            f32x4 d = *reinterpret_cast<const f32x4*>(wch + 9 * HID);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const f32x4 hv = *reinterpret_cast<const f32x4*>(hrd + (ky * HW + kx) * HS + hb * 16);
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(wch + (ky * 3 + kx) * HID);
                    d = hv * wv + d;
                }
            const uint2 dB = pack4<DT>(gelu4(d, gk));
#pragma unroll
            for (int ob = 0; ob < CB; ++ob) acc[ob] = mfma16<DT>(w2f[hb & 1][ob], dB, acc[ob]);
        }
    }
    // ---- epilogue: + bias, GELU, + residual ----
    if (!oin) return;
    const size_t row = (((size_t)img * p.h + gy) * p.w + gx) * cs;
#pragma unroll
    for (int ob = 0; ob < CB; ++ob) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.b2 + 16 * ob + 4 * g);
        const f32x4 xres = *reinterpret_cast<const f32x4*>(p.x + row + 16 * ob + 4 * g);
        f32x4 o = gelu4(acc[ob] + b, gk) + xres;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (16 * ob + 4 * g + r < p.c) ? o[r] : 0.f;
        *reinterpret_cast<f32x4*>(p.out + row + 16 * ob + 4 * g) = o;
    }
}

}  // namespace

extern "C" int i2r_hrt_attn_block(const float* x, float* out, const float* ln_w, const float* ln_b, const void* wqkv, const float* bqkv,
                                  const void* wo, const float* bo, int32_t n_img, int32_t h, int32_t w, int32_t c, int32_t cs,
                                  int32_t heads, float eps, int32_t dtype, void* stream) {
    I2R_CHECK_ARG(x && out && ln_w && ln_b && wqkv && bqkv && wo && bo, "i2r_hrt_attn_block: null pointer");
    I2R_CHECK_ARG(dtype == 1 || dtype == 2, "i2r_hrt_attn_block: dtype %d (1 bf16, 2 f16; the fp32 path is i2r_layernorm + i2r_conv + i2r_window_attn)", dtype);
    I2R_CHECK_ARG(heads > 0 && c == heads * 39 && cs % 16 == 0 && c <= cs && ((cs == 80 && heads == 2) || (cs == 160 && heads == 4)),
                  "i2r_hrt_attn_block: c=%d cs=%d heads=%d (built for the two high-resolution HRFormer-B branches: 78 / 2, 156 / 4)", c, cs, heads);
    AttnK k;
    k.x = x; k.out = out; k.ln_w = ln_w; k.ln_b = ln_b; k.wqkv = (const uint2*)wqkv; k.bqkv = bqkv; k.wo = (const uint2*)wo; k.bo = bo;
    k.n_img = n_img; k.h = h; k.w = w; k.c = c; k.eps = eps;
    k.nwy = (h + 6) / 7; k.nwx = (w + 6) / 7;
    k.pad_top = (k.nwy * 7 - h) / 2; k.pad_left = (k.nwx * 7 - w) / 2;
    const long long nblk = (long long)n_img * k.nwy * k.nwx;
    I2R_CHECK_ARG(nblk > 0 && nblk < (1ll << 31), "i2r_hrt_attn_block: grid");
    const dim3 grid((unsigned)nblk), block(256);
    if (dtype == 1) {
        if (heads == 2) hipLaunchKernelGGL((hrt_attn_block_k<1, 5, 2>), grid, block, 0, (hipStream_t)stream, k);
        else hipLaunchKernelGGL((hrt_attn_block_k<1, 10, 4>), grid, block, 0, (hipStream_t)stream, k);
    } else {
        if (heads == 2) hipLaunchKernelGGL((hrt_attn_block_k<2, 5, 2>), grid, block, 0, (hipStream_t)stream, k);
        else hipLaunchKernelGGL((hrt_attn_block_k<2, 10, 4>), grid, block, 0, (hipStream_t)stream, k);
    }
    I2R_CHECK_LAUNCH("i2r_hrt_attn_block");
    return I2R_OK;
}

extern "C" int i2r_hrt_mlp_block(const float* x, float* out, const float* ln_w, const float* ln_b, const void* w1, const float* b1,
                                 const float* wdw, const float* bdw, const void* w2, const float* b2, int32_t n_img, int32_t h, int32_t w,
                                 int32_t c, int32_t cs, int32_t hidden_pad, float eps, int32_t dtype, void* stream) {
    I2R_CHECK_ARG(x && out && x != out && ln_w && ln_b && w1 && b1 && wdw && bdw && w2 && b2, "i2r_hrt_mlp_block: null pointer / out aliases x");
    I2R_CHECK_ARG(dtype == 1 || dtype == 2, "i2r_hrt_mlp_block: dtype %d (1 bf16, 2 f16; the fp32 path is i2r_layernorm + i2r_conv + i2r_dwconv3x3)", dtype);
    I2R_CHECK_ARG((cs == 80 || cs == 160) && c <= cs && c > cs - 16 && hidden_pad >= 4 * c && hidden_pad == 4 * cs,
                  "i2r_hrt_mlp_block: c=%d cs=%d hidden_pad=%d (built for the two high-resolution HRFormer-B branches; hidden padded to 4 cs)", c, cs, hidden_pad);
    MlpK k;
    k.x = x; k.out = out; k.ln_w = ln_w; k.ln_b = ln_b; k.w1 = (const uint2*)w1; k.b1 = b1; k.wdw = wdw; k.bdw = bdw; k.w2 = (const uint2*)w2; k.b2 = b2;
    k.n_img = n_img; k.h = h; k.w = w; k.c = c; k.eps = eps;
    k.tiles_y = (h + 7) / 8; k.tiles_x = (w + 7) / 8;
    const long long nblk = (long long)n_img * k.tiles_y * k.tiles_x;
    I2R_CHECK_ARG(nblk > 0 && nblk < (1ll << 31), "i2r_hrt_mlp_block: grid");
    const dim3 grid((unsigned)nblk), block(256);
    if (dtype == 1) {
        if (cs == 80) hipLaunchKernelGGL((hrt_mlp_block_k<1, 5>), grid, block, 0, (hipStream_t)stream, k);
        else hipLaunchKernelGGL((hrt_mlp_block_k<1, 10>), grid, block, 0, (hipStream_t)stream, k);
    } else {
        if (cs == 80) hipLaunchKernelGGL((hrt_mlp_block_k<2, 5>), grid, block, 0, (hipStream_t)stream, k);
        else hipLaunchKernelGGL((hrt_mlp_block_k<2, 10>), grid, block, 0, (hipStream_t)stream, k);
    }
    I2R_CHECK_LAUNCH("i2r_hrt_mlp_block");
    return I2R_OK;
}
