// HRFormer-B transformer block, attention half, FUSED for the 16-bit modes (BASELINE configs 4-5):
//     x1 = x + out_proj( window_attention( q|k|v_proj( LayerNorm1(x) ) ) )          (reference lib/models/hrformer.py:1230-1236,
//     InterlacedPoolAttention :1164-1180, PadBlock :937-966, LocalPermuteModule :969-1001, MHA_ :692-935)
// in ONE launch per block instead of LayerNorm + q|k|v conv + window attention + out-proj conv (and without the [n, h, w, 3C] q|k|v
// tensor ever reaching HBM).  A 7x7 window is an independent problem of 49 tokens: one workgroup does it all.
//
// Round 4: every contraction runs on v_mfma_f32_16x16x32_{bf16,f16} (round 3 used the K = 16 form: twice the matrix instructions and
// twice the operand loads for the same sum), and the heads of a window are spread over HG groups of four waves.
//
// Workgroup = one window = 4 HG waves: wave (tg = wave & 3, hg = wave >> 2) owns window tokens 16 tg .. 16 tg + 15 (49 real tokens +
// 15 padding rows) and the heads hg HPW .. hg HPW + HPW - 1.  Every GEMM is computed TRANSPOSED, Y^T = W . X^T, weight fragment = MFMA A
// operand (16 rows x 32 k: lane (i = l & 15, g = l >> 4) supplies A[i][8g .. 8g+7]), token columns = B operand (lane supplies
// B[8g .. 8g+7][j = l & 15]); D: lane holds D[4g + r][l & 15].  Two D fragments packed to 16 bit form the B operand of the next GEMM
// with the k-slot order  8g + 4h + r  <->  row 16 (2s + h) + 4g + r  of k-step s -- a permutation of the 32 contraction indices that
// is applied to the other operand as well (the host permutes the out-proj columns, K rows / V^T rows are written to LDS in slot
// order), so LayerNorm -> q/k/v -> S^T = K Q^T -> softmax -> O^T = V^T P^T -> out-proj chain through registers.
// Only K rows and V^T (needed by the four token waves of a head group) pass through LDS: 128-byte rows whose 16-byte chunks are
// XOR-swizzled with (row >> 1) & 7, which makes the 64-lane ds_read_b128 of an A operand conflict-free without padding
// (checked against the lane groups of MI355X_MICROARCH.md); double-buffered per head, one barrier per head.
//   * LayerNorm in fp32 (eps 1e-6); the window's zero padding is applied AFTER it (hrformer.py:947-956): tokens outside the map are
//     exact zeros, so their q/k/v equal the projection biases and they take part as ordinary keys; rows 49..63 are masked keys.
//   * head_dim 39 is padded to 48 (zero weight rows / columns); q carries head_dim^-0.5 * log2(e) (folded by the host), the softmax
//     runs in base 2.
//   * weight fragments stream from L2 through a ring of RB units per wave -- unit = one k-step of the three 16-row dim blocks of a
//     q / k / v part -- fetched RB - 1 units ahead; MFMAs are issued k-step-major so that neighbours never share an accumulator.
//   * HG = 2 (C = 156): the two head groups meet in the out-proj -- each wave computes its heads' share of all output blocks, hands
//     the half it does not own over through LDS (the K / V^T area, dead by then) and finishes the other half: + bias + residual.
//   * fp32 residual and output (the HRFormer tower keeps its maps in fp32); operands are bf16 / f16, accumulation fp32.  The residual
//     x is re-read for the epilogue (issued before the out-proj MFMAs) instead of being carried through the kernel in registers.
#include "i2r_hrformer_attn.h"

namespace {

template <int DT>
__device__ __forceinline__ uint2 pack4(f32x4 v) {
    if constexpr (DT == 1) {
        typedef __bf16 b16x4 __attribute__((ext_vector_type(4)));
        const b16x4 b = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
        return __builtin_bit_cast(uint2, b);
    } else {
        typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
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

using AttnK = I2rAttnK;  // (i2r_hrformer_attn.h: shared with the head-per-wave kernel)

// tuning knobs of the A/B library variants (tools/ab/): ring depth per branch width, minimum waves per SIMD
#ifndef I2R_ATT_RB78
#define I2R_ATT_RB78 3
#endif
#ifndef I2R_ATT_RB156
#define I2R_ATT_RB156 2
#endif
#ifndef I2R_ATT_OCC78
#define I2R_ATT_OCC78 4
#endif
#ifndef I2R_ATT_OCC156
#define I2R_ATT_OCC156 4
#endif
#ifndef I2R_XCD_BAND
#define I2R_XCD_BAND 1   // A/B knob: 0 = windows in plain blockIdx order
#endif
// which kernel `variant` 0 selects per channel stride (measured): 2 = wave per head, 1 = wave per token tile
#ifndef I2R_ATTN_DEFAULT_VARIANT
#define I2R_ATTN_DEFAULT_VARIANT(cs) 2
#endif
constexpr int ROW = 64;  // 16-bit elements per LDS row (K: key x 64 dim slots; V^T: dim x 64 key slots) = 8 chunks of 16 bytes

template <int DT, int CB, int HEADS, int HG, int RB>
__global__ __launch_bounds__(256 * HG, HG == 1 ? I2R_ATT_OCC78 : I2R_ATT_OCC156) void hrt_attn_block_k(const AttnK p) {
    constexpr int cs = CB * 16, KS = (cs + 31) / 32, HPW = HEADS / HG, NU = 3 * HPW * KS, OS = 3 * HPW / 2, OSG = 3 * HEADS / 2;
    static_assert(HEADS % HG == 0 && HPW == 2, "two heads per wave: their six 16-dim blocks pair up into three out-proj k-steps");
    constexpr int KV_EL = (64 + 48) * ROW;                       // one head's K rows + V^T rows (16-bit elements)
    constexpr int RED_BYTES = HG == 1 ? 0 : 4 * HG * (CB / 2) * 64 * 16;
    constexpr int SMEM_BYTES = HG * 2 * KV_EL * 2 > RED_BYTES ? HG * 2 * KV_EL * 2 : RED_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
    const int tg = wave & 3, hg = wave >> 2;
    unsigned short* const kv = reinterpret_cast<unsigned short*>(smem) + hg * 2 * KV_EL;
    int bid = I2R_XCD_BAND ? xcd_band_item(blockIdx.x, p.total) : (int)blockIdx.x;  // (workgroup-uniform)
    if (bid < 0 || bid >= p.total) return;
    const int wx = bid % p.nwx; bid /= p.nwx;
    const int wy = bid % p.nwy;
    const int img = bid / p.nwy;
    // this lane's token (column li of the wave's 16-token fragment)
    const int t = tg * 16 + li;
    const int ty = t / 7, tx = t - ty * 7;
    const int y = wy * 7 + ty - p.pad_top, x = wx * 7 + tx - p.pad_left;
    const bool inmap = t < 49 && y >= 0 && y < p.h && x >= 0 && x < p.w;
    const size_t row = (((size_t)img * p.h + (inmap ? y : 0)) * p.w + (inmap ? x : 0)) * cs;
    const int sw = (li >> 1) & 7;  // swizzle of a row this lane READS as A operand (rows 16 b + li)

    // ---- the weight stream: unit u = (local head j, part q|k|v, k-step s) = the three 16-row dim blocks' fragments of that k-step, so
    //      that consecutive MFMAs go to three different accumulators (a dependent 16x16x32 chain would stall the wave at every link) ----
    f32x4 wf[RB][3];
    auto fetch = [&](int slot, int u) {
        const f32x4* src = p.wqkv + ((size_t)hg * NU + u) * 3 * 64 + lane;
#pragma unroll
        for (int db = 0; db < 3; ++db) wf[slot][db] = src[db * 64];
    };
    // ---- LayerNorm 1 of the token: features 32 s + 8 g .. + 7 of k-step s (the B operand of the projections).  The row loads go out
    //      FIRST and unconditionally (clamped addresses, selected afterwards: no exec-masked branches), the first weight units behind
    //      them: loads return in order, and the LayerNorm is what the wave needs first ----
    f32x4 xn[KS];
    {
        f32x4 xa[KS], xb[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int f0 = 32 * s + 8 * g < cs ? 32 * s + 8 * g : cs - 8;  // (cs = 80: lanes g >= 2 of the last k-step hold no channels)
            xa[s] = *reinterpret_cast<const f32x4*>(p.x + row + f0);
            xb[s] = *reinterpret_cast<const f32x4*>(p.x + row + f0 + 4);
        }
#pragma unroll
        for (int u = 0; u < RB - 1; ++u) fetch(u, u);
        float s1 = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const bool ok = inmap && (32 * s + 8 * g < cs);
            xa[s] = ok ? xa[s] : (f32x4){0.f, 0.f, 0.f, 0.f};
            xb[s] = ok ? xb[s] : (f32x4){0.f, 0.f, 0.f, 0.f};
            s1 += ((xa[s][0] + xa[s][1]) + (xa[s][2] + xa[s][3])) + ((xb[s][0] + xb[s][1]) + (xb[s][2] + xb[s][3]));  // pad channels are exact zeros
        }
        const float inv_c = __builtin_amdgcn_rcpf((float)p.c);
        const float mean = xsum4(s1) * inv_c;
        float q2 = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            // (the zero pad channels inside the row add mean^2 each: subtracted below instead of masking 8 elements per step; the
            //  channel-less lanes of a partial last step are switched off as a whole)
            const float keep = 32 * s + 8 * g < cs ? 1.f : 0.f;
            const f32x4 da = (xa[s] - mean) * keep, db = (xb[s] - mean) * keep;
            q2 += ((da[0] * da[0] + da[1] * da[1]) + (da[2] * da[2] + da[3] * da[3])) + ((db[0] * db[0] + db[1] * db[1]) + (db[2] * db[2] + db[3] * db[3]));
        }
        const float var = (xsum4(q2) - (float)(cs - p.c) * mean * mean) * inv_c;
        const float rstd = rsqrtf(fmaxf(var, 0.f) + p.eps);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int f0 = 32 * s + 8 * g < cs ? 32 * s + 8 * g : 0;  // (steps beyond cs: ok is false, any valid address)
            const f32x4 wa = *reinterpret_cast<const f32x4*>(p.ln_w + f0), wb = *reinterpret_cast<const f32x4*>(p.ln_w + f0 + 4);
            const f32x4 ba = *reinterpret_cast<const f32x4*>(p.ln_b + f0), bb = *reinterpret_cast<const f32x4*>(p.ln_b + f0 + 4);
            const float keep = (inmap && 32 * s + 8 * g < cs) ? 1.f : 0.f;  // tokens outside the map are exact zeros AFTER the LayerNorm
            // (padded ln_w = ln_b = 0 -> 0 in the pad channels)
            xn[s] = pack8<DT>(((xa[s] - mean) * rstd * wa + ba) * keep, ((xb[s] - mean) * rstd * wb + bb) * keep);
        }
    }

    f32x4 of[3 * HPW];  // attention output O^T of this wave's heads (fp32 until two blocks pair up into an out-proj k-step)
#pragma unroll
    for (int j = 0; j < HPW; ++j) {
        unsigned short* const kb = kv + (j & 1) * KV_EL;
        unsigned short* const vt = kb + 64 * ROW;
        f32x4 q[3];
#pragma unroll
        for (int part = 0; part < 3; ++part) {
            f32x4 acc[3];
            {   // bias of features 4g + r of the three dim blocks
                const float* bsrc = p.bqkv + ((hg * HPW + j) * 3 + part) * 48 + 4 * g;
#pragma unroll
                for (int db = 0; db < 3; ++db) acc[db] = *reinterpret_cast<const f32x4*>(bsrc + 16 * db);
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int u = (j * 3 + part) * KS + s;
                if (u + RB - 1 < NU) fetch((u + RB - 1) % RB, u + RB - 1);
#pragma unroll
                for (int db = 0; db < 3; ++db) acc[db] = mfma32_lp<DT>(wf[u % RB][db], xn[s], acc[db]);
            }
            if (part == 0) {
#pragma unroll
                for (int db = 0; db < 3; ++db) q[db] = acc[db];
            } else if (part == 1) {
                // K[key t][dims 16 db + 4g ..]: slot chunk (db >> 1) * 4 + g, half db & 1; block 2 fills its (empty) partner half with zeros
                const int c0 = (g ^ ((t >> 1) & 7)) * 8, c1 = ((4 + g) ^ ((t >> 1) & 7)) * 8;
                *reinterpret_cast<f32x4*>(kb + t * ROW + c0) = pack8<DT>(acc[0], acc[1]);
                *reinterpret_cast<f32x4*>(kb + t * ROW + c1) = pack8<DT>(acc[2], (f32x4){0.f, 0.f, 0.f, 0.f});
            } else {
                // V^T[dim 16 db + 4g + r][key t]: key slot (tg >> 1) * 32 + 8 (li >> 2) + 4 (tg & 1) + (li & 3)
#pragma unroll
                for (int db = 0; db < 3; ++db)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int d = 16 * db + 4 * g + r;
                        const int chunk = ((tg >> 1) * 4 + (li >> 2)) ^ ((2 * g + (r >> 1)) & 7);  // ((d >> 1) & 7 = 2g + (r >> 1))
                        vt[d * ROW + chunk * 8 + 4 * (tg & 1) + (li & 3)] = pack1<DT>(acc[db][r]);
                    }
            }
        }
        const f32x4 qB0 = pack8<DT>(q[0], q[1]), qB1 = pack8<DT>(q[2], (f32x4){0.f, 0.f, 0.f, 0.f});
        __syncthreads();
        // ---- S^T[key][query] for the 64 keys (k-step-major: four independent accumulators), softmax over the 49 real ones (base 2) ----
        f32x4 st[4];
#pragma unroll
        for (int kf = 0; kf < 4; ++kf)
            st[kf] = mfma32_lp<DT>(*reinterpret_cast<const f32x4*>(kb + (16 * kf + li) * ROW + ((0 + g) ^ sw) * 8), qB0, (f32x4){0.f, 0.f, 0.f, 0.f});
#pragma unroll
        for (int kf = 0; kf < 4; ++kf)
            st[kf] = mfma32_lp<DT>(*reinterpret_cast<const f32x4*>(kb + (16 * kf + li) * ROW + ((4 + g) ^ sw) * 8), qB1, st[kf]);
        float mx = -__builtin_inff();
#pragma unroll
        for (int kf = 0; kf < 4; ++kf)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (16 * kf + 4 * g + r >= 49) st[kf][r] = -__builtin_inff();
                mx = fmaxf(mx, st[kf][r]);
            }
        mx = xmax4(mx);
        float sum = 0.f;
#pragma unroll
        for (int kf = 0; kf < 4; ++kf)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                st[kf][r] = __builtin_amdgcn_exp2f(st[kf][r] - mx);
                sum += st[kf][r];
            }
        const f32x4 pB0 = pack8<DT>(st[0], st[1]), pB1 = pack8<DT>(st[2], st[3]);
        const float inv = 1.f / xsum4(sum);
        // ---- O^T[dim][query] = V^T P^T ----
        f32x4 o[3];
#pragma unroll
        for (int db = 0; db < 3; ++db)
            o[db] = mfma32_lp<DT>(*reinterpret_cast<const f32x4*>(vt + (16 * db + li) * ROW + ((0 + g) ^ sw) * 8), pB0, (f32x4){0.f, 0.f, 0.f, 0.f});
#pragma unroll
        for (int db = 0; db < 3; ++db)
            o[db] = mfma32_lp<DT>(*reinterpret_cast<const f32x4*>(vt + (16 * db + li) * ROW + ((4 + g) ^ sw) * 8), pB1, o[db]);
#pragma unroll
        for (int db = 0; db < 3; ++db) of[j * 3 + db] = o[db] * inv;
        // (no second barrier: the next head writes the other K / V^T buffer, and the one after that is separated by the next barrier)
    }

    // ---- out-proj over this wave's heads (k-steps hg OS .. hg OS + OS - 1 of the packed matrix) ----
    f32x4 oB[OS];
#pragma unroll
    for (int s = 0; s < OS; ++s) oB[s] = pack8<DT>(of[2 * s], of[2 * s + 1]);
    constexpr int OWN = CB / HG;  // output blocks this wave finishes: hg OWN .. hg OWN + OWN - 1 (HG = 1: all of them)
    static_assert(CB % HG == 0, "even split of the output blocks");
    f32x4 xres[OWN];
#pragma unroll
    for (int i = 0; i < OWN; ++i)  // the residual, fetched before the MFMAs that need it last
        xres[i] = inmap ? *reinterpret_cast<const f32x4*>(p.x + row + 16 * (hg * OWN + i) + 4 * g) : (f32x4){0.f, 0.f, 0.f, 0.f};
    // a group of NB consecutive output blocks, k-step-major (NB independent accumulators), fragments of a k-step fetched one step ahead
    auto outgroup = [&](int ob0, auto& acc) {
        constexpr int NB = sizeof(acc) / sizeof(acc[0]);
        f32x4 wv[2][NB];
        auto fetch_o = [&](int slot, int s) {
#pragma unroll
            for (int i = 0; i < NB; ++i) wv[slot][i] = p.wo[((size_t)(ob0 + i) * OSG + hg * OS + s) * 64 + lane];
        };
        fetch_o(0, 0);
#pragma unroll
        for (int s = 0; s < OS; ++s) {
            if (s + 1 < OS) fetch_o((s + 1) & 1, s + 1);
#pragma unroll
            for (int i = 0; i < NB; ++i) acc[i] = mfma32_lp<DT>(wv[s & 1][i], oB[s], acc[i]);
        }
    };
    if constexpr (HG == 1) {
        f32x4 acc[CB];
#pragma unroll
        for (int ob = 0; ob < CB; ++ob) acc[ob] = *reinterpret_cast<const f32x4*>(p.bo + 16 * ob + 4 * g);
        outgroup(0, acc);
#pragma unroll
        for (int ob = 0; ob < CB; ++ob)
            if (inmap) *reinterpret_cast<f32x4*>(p.out + row + 16 * ob + 4 * g) = acc[ob] + xres[ob];
    } else {
        f32x4* const red = reinterpret_cast<f32x4*>(smem);  // [tg][hg of the OWNER][OWN blocks][64 lanes]
        const int other = hg ^ 1;
        f32x4 mine[OWN], theirs[OWN];
#pragma unroll
        for (int i = 0; i < OWN; ++i) {
            theirs[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            mine[i] = *reinterpret_cast<const f32x4*>(p.bo + 16 * (hg * OWN + i) + 4 * g);
        }
        outgroup(other * OWN, theirs);  // the partner's blocks first: it waits for them
        __syncthreads();  // every wave has finished reading K / V^T of the last head: the area becomes the exchange buffer
#pragma unroll
        for (int i = 0; i < OWN; ++i) red[((tg * HG + other) * OWN + i) * 64 + lane] = theirs[i];
        outgroup(hg * OWN, mine);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < OWN; ++i) {
            const f32x4 o = (mine[i] + red[((tg * HG + hg) * OWN + i) * 64 + lane]) + xres[i];
            if (inmap) *reinterpret_cast<f32x4*>(p.out + row + 16 * (hg * OWN + i) + 4 * g) = o;
        }
    }
}

template <int DT>
int launch(const AttnK& k, int heads, long long nblk, hipStream_t stream) {
    const dim3 grid((unsigned)((nblk + 7) / 8 * 8));
    if (heads == 2) i2r_launch((hrt_attn_block_k<DT, 5, 2, 1, I2R_ATT_RB78>), grid, dim3(256), 0, stream, k);
    else i2r_launch((hrt_attn_block_k<DT, 10, 4, 2, I2R_ATT_RB156>), grid, dim3(512), 0, stream, k);
    return 0;
}

}  // namespace

extern "C" int i2r_hrt_attn_block(const float* x, float* out, const float* ln_w, const float* ln_b, const void* wqkv, const float* bqkv,
                                  const void* wo, const float* bo, int32_t n_img, int32_t h, int32_t w, int32_t c, int32_t cs,
                                  int32_t heads, float eps, int32_t dtype, int32_t variant, void* stream) {
    I2R_CHECK_ARG(x && out && x != out && ln_w && ln_b && wqkv && bqkv && wo && bo, "i2r_hrt_attn_block: null pointer / out aliases x");
    I2R_CHECK_ARG(dtype == 1 || dtype == 2, "i2r_hrt_attn_block: dtype %d (1 bf16, 2 f16; the fp32 path is i2r_layernorm + i2r_conv + i2r_window_attn)", dtype);
    const bool narrow = (cs == 80 && heads == 2) || (cs == 160 && heads == 4), wide = (cs == 320 && heads == 8) || (cs == 624 && heads == 16);
    I2R_CHECK_ARG(heads > 0 && c == heads * 39 && cs % 16 == 0 && c <= cs && (narrow || wide),
                  "i2r_hrt_attn_block: c=%d cs=%d heads=%d (built for the HRFormer-B branches: 78 / 2, 156 / 4, 312 / 8, 624 / 16)", c, cs, heads);
    I2R_CHECK_ARG(variant >= 0 && variant <= 2 && (variant != 1 || narrow),
                  "i2r_hrt_attn_block: variant %d (0 default, 1 wave per token tile: 78 / 156 only, 2 wave per head)", variant);
    I2R_CHECK_ARG((long long)n_img * h * w * cs < (1ll << 31), "i2r_hrt_attn_block: tensor too large");
    AttnK k;
    k.x = x; k.out = out; k.ln_w = ln_w; k.ln_b = ln_b; k.wqkv = (const f32x4*)wqkv; k.bqkv = bqkv; k.wo = (const f32x4*)wo; k.bo = bo;
    k.n_img = n_img; k.h = h; k.w = w; k.c = c; k.eps = eps;
    k.nwy = (h + 6) / 7; k.nwx = (w + 6) / 7;
    k.pad_top = (k.nwy * 7 - h) / 2; k.pad_left = (k.nwx * 7 - w) / 2;
    const long long nblk = (long long)n_img * k.nwy * k.nwx;
    I2R_CHECK_ARG(nblk > 0 && nblk < (1ll << 30), "i2r_hrt_attn_block: grid");
    k.total = (int)nblk;
    if (variant == 0) variant = I2R_ATTN_DEFAULT_VARIANT(cs);
    if (variant == 2 || wide) {
        const bool ok = i2r_attn_head_launch(k, dtype, cs, heads, nblk, (hipStream_t)stream);
        I2R_CHECK_ARG(ok, "i2r_hrt_attn_block: no head-per-wave kernel for cs=%d heads=%d", cs, heads);
    } else if (dtype == 1) {
        launch<1>(k, heads, nblk, (hipStream_t)stream);
    } else {
        launch<2>(k, heads, nblk, (hipStream_t)stream);
    }
    I2R_CHECK_LAUNCH("i2r_hrt_attn_block");
    return I2R_OK;
}
