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
