// HRFormer-B transformer block, MLP half, FUSED for the 16-bit modes -- the form for the WIDE branches (round 5):
//     x2 = x1 + GELU(BN3(fc2( GELU(BN2(DW3x3( GELU(BN1(fc1( LayerNorm2(x1) ))) ))) )))
// (reference lib/models/hrformer.py:1237 + MlpDWBN.forward :1094-1119; BatchNorms folded by the host; same C-ABI entry point, operand
// images and arithmetic as i2r_hrformer_mlp.hip).
//
// i2r_hrformer_mlp.hip lets every wave accumulate fc2 over ITS hidden blocks for all output blocks: 3 x CB accumulator fragments per
// wave (CB = 10: 120 registers, one wave per SIMD at 406 registers; CB = 20 does not fit at all) and a reduction at the end.  Here fc2
// is split by OUTPUT block instead:
//   * workgroup = one 8 x 6 pixel tile + halo (10 x 8 = 80 pixels = five 16-pixel fragments), W = 10 waves;
//   * LayerNorm 2 once per workgroup, rows read quad-coalesced, written to LDS as the packed B-operand image all waves read (the
//     existing kernel keeps the 5 x KS fragments in registers: 100 / 200 of them);
//   * the hidden dimension goes by in ROUNDS of W pairs of 16-channel blocks: wave w runs fc1 (both blocks of its pair share every
//     LDS read of the pixel columns: 10 matrix instructions per weight-fragment pair), GELU, the depth-wise 3x3 through its private
//     LDS tile, GELU -- exactly the per-wave pipeline of the existing kernel -- and puts the pair's packed 32-deep B operand
//     (3 output-pixel fragments) into LDS; after a barrier every wave accumulates ITS CB / W output blocks over the round's W pairs
//     (a weight fragment feeds three matrix instructions): 3 x CB / W accumulator fragments, no reduction, + bias, GELU, + residual,
//     stored quad-coalesced.
// LDS (C = 312): 50 KB pixel columns + 30 KB B operands of a round + 70 KB hidden tiles + 6 KB depth-wise weights.
#include "i2r_hrformer_mlp.h"

namespace {

#ifndef I2R_XCD_BAND
#define I2R_XCD_BAND 1
#endif
constexpr int TY = 8, TX = 6;                    // output sub-tile (rows x columns)
constexpr int HY = TY + 2, HX = TX + 2;          // halo grid 10 x 8 = 80 pixels = 5 fragments (two halo rows each)
constexpr int NF = HY * HX / 16, NPF = TY * TX / 16;
constexpr int H_ROW = 40, H_QUAD = 448;          // LDS strides of a hidden tile in floats (i2r_hrformer_mlp.hip, tools/lds_layout.py)
static_assert(NF == 5 && NPF == 3, "sub-tile geometry");

template <int DT, int CB, int W>
__global__ __launch_bounds__(64 * W, (W + 3) / 4) void hrt_mlp_wide_k(const I2rMlpK p) {
    constexpr int cs = CB * 16, KS = (cs + 31) / 32, HBT = 4 * CB, HID = HBT * 16, NOB = CB / W, R = HBT / (2 * W);
    constexpr int KP = W / NF, NP = cs / 4, KQ = 2 * KS / KP;
    static_assert(CB % W == 0 && HBT % (2 * W) == 0 && W % NF == 0 && (2 * KS) % KP == 0, "waves / output blocks / hidden pairs / LayerNorm slices");
    __shared__ __attribute__((aligned(16))) f32x4 Xs[NF * KS * 64];   // LayerNorm-ed halo pixels: [fragment][k-step][lane] packed B operands
    __shared__ __attribute__((aligned(16))) f32x4 Ds[W * NPF * 64];   // a round's fc2 B operands: [pair of the round][pixel fragment][lane]
    __shared__ __attribute__((aligned(16))) float Hs[W * 4 * H_QUAD];  // per wave: hidden tile of one 16-channel block
    __shared__ __attribute__((aligned(16))) float Wd[W * 160];         // per wave: depth-wise weights + bias of that block [10][16]
    __shared__ float stat[2][KP * NF * 16];
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = I2R_XCD_BAND ? xcd_band_item(blockIdx.x, p.total) : (int)blockIdx.x;  // (workgroup-uniform)
    if (bid < 0 || bid >= p.total) return;
    const int sx = bid % p.tiles_x; bid /= p.tiles_x;
    const int sy = bid % p.tiles_y;
    const int img = bid / p.tiles_y;
    const int y0 = sy * TY - 1, x0 = sx * TX - 1;               // image coordinate of halo pixel (0, 0)
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // Weight fragments are fetched a whole BLOCK ahead (all KS fc1 fragments of the next hidden block fly under the GELU / depth-wise
    // phase of the current one; fc2's three pairs ahead): with one fragment of look-ahead a wave paid an L2 round trip per k-step.
    // The first block's go out before anything else: they fly under the LayerNorm's row loads.
    f32x4 w1f[KS];
    auto fetch1 = [&](int hb) {
#pragma unroll
        for (int s = 0; s < KS; ++s) w1f[s] = p.w1[((size_t)hb * KS + s) * 64 + lane];
    };
    fetch1(2 * wave);

    // ---- LayerNorm 2 of halo fragment wave % NF, channel slice wave / NF -> Xs (rows read quad-coalesced: lane n = pixel n >> 2 of the
    //      fragment, 16-byte pieces (n & 3) + 4 k; piece pi is half pi & 1 of lane (g = (pi & 7) >> 1, li = pixel) of k-step pi >> 3) ----
    {
        const int f = wave % NF, kh = wave / NF;
        const int tl = lane >> 2, q = lane & 3;
        const int y = y0 + 2 * f + (tl >> 3), x = x0 + (tl & 7);
        const bool in = y >= 0 && y < p.h && x >= 0 && x < p.w;
        const float* row = p.x + (((size_t)img * p.h + (in ? y : 0)) * p.w + (in ? x : 0)) * cs;
        const float inv_c = __builtin_amdgcn_rcpf((float)p.c);
        const float npad = (float)(cs - p.c);
        f32x4 xv[KQ];
#pragma unroll
        for (int k = 0; k < KQ; ++k) {
            const int pi = q + 4 * (kh * KQ + k);
            xv[k] = *reinterpret_cast<const f32x4*>(row + 4 * (pi < NP ? pi : 0));
        }
        float s1 = 0.f;
#pragma unroll
        for (int k = 0; k < KQ; ++k) {
            const bool ok = in && q + 4 * (kh * KQ + k) < NP;
            xv[k] = ok ? xv[k] : zero4;
            s1 += (xv[k][0] + xv[k][1]) + (xv[k][2] + xv[k][3]);
        }
        s1 += __shfl_xor(s1, 1);
        s1 += __shfl_xor(s1, 2);
        if constexpr (KP > 1) {
            if (q == 0) stat[0][(kh * NF + f) * 16 + tl] = s1;
            __syncthreads();
            s1 = 0.f;
#pragma unroll
            for (int kk = 0; kk < KP; ++kk) s1 += stat[0][(kk * NF + f) * 16 + tl];
        }
        const float mean = s1 * inv_c;
        float q2 = 0.f;
#pragma unroll
        for (int k = 0; k < KQ; ++k) {
            const float keep = q + 4 * (kh * KQ + k) < NP ? 1.f : 0.f;
            const f32x4 d = (xv[k] - mean) * keep;
            q2 += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
        q2 += __shfl_xor(q2, 1);
        q2 += __shfl_xor(q2, 2);
        if constexpr (KP > 1) {
            if (q == 0) stat[1][(kh * NF + f) * 16 + tl] = q2;
            __syncthreads();
            q2 = 0.f;
#pragma unroll
            for (int kk = 0; kk < KP; ++kk) q2 += stat[1][(kk * NF + f) * 16 + tl];
        }
        const float var = (q2 - npad * mean * mean) * inv_c;
        const float rstd = rsqrtf(fmaxf(var, 0.f) + p.eps);
#pragma unroll
        for (int k = 0; k < KQ; ++k) {
            const int pi = q + 4 * (kh * KQ + k);
            const bool has = pi < NP;
            const f32x4 wa = *reinterpret_cast<const f32x4*>(p.ln_w + 4 * (has ? pi : 0)), ba = *reinterpret_cast<const f32x4*>(p.ln_b + 4 * (has ? pi : 0));
            const float keep = (in && has) ? 1.f : 0.f;  // outside the image: zero columns (the depth-wise conv zero-pads the HIDDEN map)
            reinterpret_cast<uint2*>(Xs + (f * KS + (pi >> 3)) * 64 + ((pi & 7) >> 1) * 16 + tl)[pi & 1] = pack4<DT>(((xv[k] - mean) * rstd * wa + ba) * keep);
        }
    }
    float hinf[NF];  // 1 = halo pixel (16 f + li) lies inside the image (fc1's bias must not reach the zero padding of the hidden map)
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int y = y0 + 2 * f + (li >> 3), x = x0 + (li & 7);
        hinf[f] = (y >= 0 && y < p.h && x >= 0 && x < p.w) ? 1.f : 0.f;
    }
    __syncthreads();

    float* const Hw = Hs + wave * 4 * H_QUAD;
    float* const Wdw = Wd + wave * 160;
    // this lane's three output pixels: row oy, columns 3 xb .. 3 xb + 2 of the sub-tile
    const int oy = li & 7, xb = li >> 3;
    float* const hwr = Hw + g * H_QUAD + (li >> 3) * H_ROW + (li & 7) * 4;   // fc1 result of halo fragment 0 (fragment f: + 2 f rows)
    const float* const hrd = Hw + g * H_QUAD + oy * H_ROW + (3 * xb) * 4;    // top-left tap of the strip
    const float* const wrd = Wdw + 4 * g;
    const GeluC gk;

    auto store_h = [&](const f32x4 (&a)[NF]) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            *reinterpret_cast<f32x4*>(hwr + 2 * f * H_ROW) = gelu4<4>(a[f], gk);
            // (a fence per fragment: left alone, the scheduler runs all twenty GELUs side by side and spills sixty registers around them --
            //  a ten-wave workgroup has 168)
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // depth-wise 3x3 + bias + GELU of the block in Hw / Wdw (three output pixels x 4 channels per lane) -> packed halves of the fc2 B operand
    auto dwconv = [&](uint2 (&dp)[NPF]) {
        f32x4 d[NPF];
        {
            const f32x4 bias = *reinterpret_cast<const f32x4*>(wrd + 9 * 16);
#pragma unroll
            for (int pf = 0; pf < NPF; ++pf) d[pf] = bias;
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            f32x4 t[5], wv[3];
#pragma unroll
            for (int j = 0; j < 5; ++j) t[j] = *reinterpret_cast<const f32x4*>(hrd + ky * H_ROW + j * 4);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) wv[kx] = *reinterpret_cast<const f32x4*>(wrd + (ky * 3 + kx) * 16);
#pragma unroll
            for (int pf = 0; pf < NPF; ++pf)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) d[pf] = t[pf + kx] * wv[kx] + d[pf];
            __builtin_amdgcn_sched_barrier(0);  // (one tap row at a time: hoisted to the top, the 25 LDS reads of the three rows are spilled at once)
        }
#pragma unroll
        for (int pf = 0; pf < NPF; ++pf) {
            dp[pf] = pack4<DT>(gelu4<4>(d[pf], gk));
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // depth-wise weights + bias of hidden block hb: lane i < 40 fetches (tap = i >> 2 | bias, channel quad i & 3)
    auto fetch_dw = [&](int hb) {
        const int i = lane < 40 ? lane : 0, tap = i >> 2, qq = i & 3;
        return tap < 9 ? *reinterpret_cast<const f32x4*>(p.wdw + tap * HID + hb * 16 + 4 * qq)
                       : *reinterpret_cast<const f32x4*>(p.bdw + hb * 16 + 4 * qq);
    };

    f32x4 acc[NOB][NPF];
#pragma unroll
    for (int o = 0; o < NOB; ++o)
#pragma unroll
        for (int pf = 0; pf < NPF; ++pf) acc[o][pf] = zero4;

#pragma unroll 1
    for (int r = 0; r < R; ++r) {
        const int gp = wave + W * r;  // this wave's pair of hidden blocks 2 gp, 2 gp + 1 in this round
        const f32x4 dwv0 = fetch_dw(2 * gp), dwv1 = fetch_dw(2 * gp + 1);
        // ---- per block: fc1 on the five halo fragments, GELU -> hidden tile -> depth-wise 3x3 -> GELU (wave-private LDS: a wave's LDS
        //      accesses execute in order) ----
        uint2 d0[NPF], d1[NPF];
        auto block = [&](const int hb, const int hb_next, const f32x4 dwv, uint2 (&dp)[NPF]) {
            f32x4 a[NF];
            {
                const f32x4 b = *reinterpret_cast<const f32x4*>(p.b1 + hb * 16 + 4 * g);
#pragma unroll
                for (int f = 0; f < NF; ++f) a[f] = b * hinf[f];
            }
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int f = 0; f < NF; ++f) a[f] = mfma32_lp<DT>(w1f[s], Xs[(f * KS + s) * 64 + lane], a[f]);
            fetch1(hb_next);
            __builtin_amdgcn_wave_barrier();  // (the previous block's taps have all been read)
            if (lane < 40) *reinterpret_cast<f32x4*>(Wdw + 4 * lane) = dwv;
            store_h(a);
            __builtin_amdgcn_wave_barrier();
            dwconv(dp);
            // (pin the results HERE: the compiler otherwise sinks the tap arithmetic of the first block below the second block's fc1,
            //  down to the first use of dp, and carries the 25 raw LDS reads there through scratch)
#pragma unroll
            for (int pf = 0; pf < NPF; ++pf) asm volatile("" : "+v"(dp[pf].x), "+v"(dp[pf].y));
        };
        block(2 * gp, 2 * gp + 1, dwv0, d0);
        block(2 * gp + 1, r + 1 < R ? 2 * (gp + W) : 2 * gp + 1, dwv1, d1);  // (last round: a harmless re-fetch instead of a branch)
        // ---- the pair's 32-deep B operand -> LDS (slot 8g + 4h + r <- hidden channel 16 (2 pair + h) + 4g + r) ----
        const f32x4* w2s = p.w2 + (size_t)W * r * 64 + lane;  // fragment (ob, pair): (ob * (HBT / 2) + pair) * 64
        constexpr int RD = 3;  // fc2 fragments in flight: pairs k .. k + RD - 1
        f32x4 ring[RD][NOB];
#pragma unroll
        for (int k = 0; k < RD; ++k)
#pragma unroll
            for (int o = 0; o < NOB; ++o) ring[k][o] = w2s[((size_t)(wave + o * W) * (HBT / 2) + k) * 64];
        if (r > 0) __syncthreads();  // every wave has finished fc2 of the previous round: Ds is free
#pragma unroll
        for (int pf = 0; pf < NPF; ++pf) Ds[(wave * NPF + pf) * 64 + lane] = join8(d0[pf], d1[pf]);
        __syncthreads();
        // ---- fc2: this wave's output blocks wave + o W over the round's W pairs ----
#pragma unroll
        for (int k = 0; k < W; ++k) {
#pragma unroll
            for (int pf = 0; pf < NPF; ++pf) {
                const f32x4 bf = Ds[(k * NPF + pf) * 64 + lane];
#pragma unroll
                for (int o = 0; o < NOB; ++o) acc[o][pf] = mfma32_lp<DT>(ring[k % RD][o], bf, acc[o][pf]);
            }
            if (k + RD < W) {
#pragma unroll
                for (int o = 0; o < NOB; ++o) ring[k % RD][o] = w2s[((size_t)(wave + o * W) * (HBT / 2) + k + RD) * 64];
            }
            __builtin_amdgcn_sched_barrier(0);  // (keep the ring a ring: unfenced, all fragment loads of the round move to the top)
        }
    }

    // ---- + bias, GELU, + residual; D fragments (lane (li, g): channels 4g + r of pixel li) re-numbered so that lane n holds piece n & 3 of
    //      pixel n >> 2: residual loads and stores are quad-coalesced ----
    const int tl = lane >> 2, pq = lane & 3, srcl = tl + 16 * pq;
    const int gy = sy * TY + (tl & 7), xbn = tl >> 3;
#pragma unroll
    for (int pf = 0; pf < NPF; ++pf) {
        const int gx = sx * TX + 3 * xbn + pf;
        const bool oin = gy < p.h && gx < p.w;
        const size_t prow = (((size_t)img * p.h + (gy < p.h ? gy : 0)) * p.w + (gx < p.w ? gx : 0)) * cs + 4 * pq;
#pragma unroll
        for (int o = 0; o < NOB; ++o) {
            const int ob = wave + o * W;
            const f32x4 xres = *reinterpret_cast<const f32x4*>(p.x + prow + 16 * ob);
            const f32x4 b = *reinterpret_cast<const f32x4*>(p.b2 + 16 * ob + 4 * g);
            const f32x4 v = gelu4<5>(acc[o][pf] + b, gk);  // (pad channels: zero W2 rows and bias -> GELU(0) = 0 exactly, + the residual's zero)
            f32x4 vn;
#pragma unroll
            for (int e = 0; e < 4; ++e) vn[e] = __shfl(v[e], srcl);  // (ds_bpermute_b32)
            if (oin) *reinterpret_cast<f32x4*>(p.out + prow + 16 * ob) = vn + xres;
        }
    }
}

template <int DT>
bool launch(const I2rMlpK& k, int cs, long long nblk, hipStream_t stream) {
    const dim3 grid((unsigned)((nblk + 7) / 8 * 8));
#ifndef I2R_MLPW_W312
#define I2R_MLPW_W312 10
#endif
#ifndef I2R_MLPW_W156
#define I2R_MLPW_W156 10
#endif
    if (cs == 320) i2r_launch((hrt_mlp_wide_k<DT, 20, I2R_MLPW_W312>), grid, dim3(64 * I2R_MLPW_W312), 0, stream, k);
    else if (cs == 160) i2r_launch((hrt_mlp_wide_k<DT, 10, I2R_MLPW_W156>), grid, dim3(64 * I2R_MLPW_W156), 0, stream, k);
    else if (cs == 80) i2r_launch((hrt_mlp_wide_k<DT, 5, 5>), grid, dim3(320), 0, stream, k);
    else return false;
    return true;
}

}  // namespace

bool i2r_mlp_wide_launch(const I2rMlpK& k, int dtype, int cs, long long nblk, hipStream_t stream) {
    return dtype == 1 ? launch<1>(k, cs, nblk, stream) : launch<2>(k, cs, nblk, stream);
}
