// HRFormer-B transformer block, attention half, FUSED for the 16-bit modes -- the HEAD-PER-WAVE form (round 5):
//     x1 = x + out_proj( window_attention( q|k|v_proj( LayerNorm1(x) ) ) )          (reference lib/models/hrformer.py:1230-1236,
//     InterlacedPoolAttention :1164-1180, PadBlock :937-966, LocalPermuteModule :969-1001, MHA_ :692-935)
// Same C-ABI entry point, operand images and arithmetic as i2r_hrformer_lp.hip (one wave per 16-token tile of the window); what
// changes is the decomposition, and with it the traffic: there a weight fragment (1 KB through the CU's 64 B/clk vector-memory path)
// feeds ONE matrix instruction, so the four SIMDs ask for four times what that path delivers and the kernel waits on its weight
// stream (DESIGN.md, round-4 ablation: 8 us of arithmetic in a 25 us launch).
//
// Workgroup = one 7x7 window (49 tokens + 15 padding rows = four 16-token tiles); wave = ONE HEAD (HPW heads one after the other for
// the 16-head branch) over ALL four tiles:
//   * a q / k / v weight fragment is fetched once per wave and feeds FOUR matrix instructions (one per token tile);
//   * the whole attention of a head stays in the wave's registers -- no K / V^T staging through LDS, no barrier per head:
//       Q^T, K^T = W . X^T   (A = weight fragment, B = LayerNorm-ed token columns): D = [dim rows][token column];
//       two 16-dim D fragments packed to 16 bit are at once the A operand "K rows" (lane = key) and the B operand "Q^T" (lane = query)
//       of  S^T = K Q^T  with the k-slot order 8g + 4h + r <-> dim 16 (2s + h) + 4g + r  (the permutation of i2r_hrformer_lp.hip);
//       V = X . Wv^T  with the operands SWAPPED (A = token columns, B = weight fragment: the same two register images):
//       D = [token rows][dim column], and two token tiles packed side by side are the A operand "V^T rows" of  O^T = V^T P^T
//       in exactly the key-slot order in which the packed softmax output P^T arrives as B operand;
//   * LayerNorm 1 is computed once per window (the waves split tiles / channel slices, statistics through LDS) and written to LDS as
//     the packed operand image every head reads;
//   * O^T of all heads meets in LDS (the LayerNorm area, dead by then) for the out-proj: wave w finishes token-tile pair w & 1 of the
//     output blocks (w >> 1) + i NW/2 -- a weight fragment feeds two matrix instructions, + bias + residual (re-read), fp32 store.
// Tokens outside the map are exact zeros after the LayerNorm (their q / k / v are the projection biases, hrformer.py:947-956), rows
// 49..63 are masked keys; head_dim 39 padded to 48; q carries head_dim^-0.5 * log2(e), softmax in base 2; fp32 accumulation.
#include <type_traits>

#include "i2r_hrformer_attn.h"

namespace {

#ifndef I2R_XCD_BAND
#define I2R_XCD_BAND 1
#endif

template <int DT, int CB, int HEADS, int HPW, int OCC>
__global__ __launch_bounds__(64 * (HEADS / HPW), OCC) void hrt_attn_head_k(const I2rAttnK p) {
    constexpr int NW = HEADS / HPW, cs = CB * 16, KS = (cs + 31) / 32, OSG = 3 * HEADS / 2;
    constexpr int KP = NW >= 4 ? NW / 4 : 1;      // channel slices of the LayerNorm (waves w, w + 4, ... share token tile w & 3)
    constexpr int TPW = NW >= 4 ? 1 : 4 / NW;     // token tiles a wave normalises
    constexpr int KSL = KS / KP;
    static_assert(KS % KP == 0 && HEADS % HPW == 0 && (NW & 1) == 0, "LayerNorm slices / waves");
    constexpr int XS = 4 * KS * 64, OBN = 4 * OSG * 64;
    __shared__ __attribute__((aligned(16))) f32x4 smem[XS > OBN ? XS : OBN];  // LayerNorm-ed tokens [tile][k-step][lane]; later O^T [tile][k-step][lane]
    __shared__ float stat[2][KP > 1 ? KP * 64 : 1];
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = I2R_XCD_BAND ? xcd_band_item(blockIdx.x, p.total) : (int)blockIdx.x;  // (workgroup-uniform)
    if (bid < 0 || bid >= p.total) return;
    const int wx = bid % p.nwx; bid /= p.nwx;
    const int wy = bid % p.nwy;
    const int img = bid / p.nwy;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // the first weight fragments of this wave's (first) head go out before anything else: they fly under the LayerNorm's row loads
    constexpr int RDQ = 2;  // ring depth of the q / k / v weight stream (narrow branches, see below)
    f32x4 wr[RDQ + 1][3];
    auto ufetch = [&](const int head, const int n) {  // unit n of a head: part 2 - n / KS (v, k, q), k-step n % KS
        const f32x4* a = p.wqkv + ((size_t)(head * 3 + (2 - n / KS)) * KS + n % KS) * 192 + lane;
#pragma unroll
        for (int db = 0; db < 3; ++db) wr[n % (RDQ + 1)][db] = a[db * 64];
    };
    if constexpr (KS <= 10) {
#pragma unroll
        for (int n = 0; n < RDQ; ++n) ufetch(wave * HPW, n);
    }

    // ---- LayerNorm 1 -> smem as packed operand image: lane (li, g) of (tile t, k-step s) holds features 32 s + 8 g .. + 7 of token 16 t + li.
    //      The rows are READ quad-coalesced: lane n takes token n >> 2 of the tile and the 16-byte pieces (n & 3) + 4 k of its row, so the
    //      four lanes of a quad cover 64 contiguous bytes (a 64-lane access whose quads straddle rows takes four times the addresser
    //      cycles, DESIGN.md "quad rule"); statistics: in-lane, then over the quad; piece pi lands in half pi & 1 of lane
    //      (g = (pi & 7) >> 1, li = token) of k-step pi >> 3. ----
    {
        constexpr int NP = cs / 4;               // 16-byte pieces of a row
        constexpr int KQ = 2 * KS / KP;          // pieces per lane and slice (all 8 KS pieces of the operand image: beyond the row, zeros)
        static_assert((2 * KS) % KP == 0, "pieces per LayerNorm slice");
        const float inv_c = __builtin_amdgcn_rcpf((float)p.c);
        const float npad = (float)(cs - p.c);  // zero pad channels inside the row: each adds mean^2 to the sum of squares
        const int tl = lane >> 2, q = lane & 3;
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int t = NW >= 4 ? (wave & 3) : wave + i * NW;
            const int kh = NW >= 4 ? (wave >> 2) : 0;
            const int tt = 16 * t + tl;
            const int ty = tt / 7, tx = tt - ty * 7;
            const int y = wy * 7 + ty - p.pad_top, x = wx * 7 + tx - p.pad_left;
            const bool inmap = tt < 49 && y >= 0 && y < p.h && x >= 0 && x < p.w;
            const float* row = p.x + (((size_t)img * p.h + (inmap ? y : 0)) * p.w + (inmap ? x : 0)) * cs;
            f32x4 xv[KQ];
#pragma unroll
            for (int k = 0; k < KQ; ++k) {  // (unconditional loads from clamped addresses, selected afterwards)
                const int pi = q + 4 * (kh * KQ + k);
                xv[k] = *reinterpret_cast<const f32x4*>(row + 4 * (pi < NP ? pi : 0));
            }
            float s1 = 0.f;
#pragma unroll
            for (int k = 0; k < KQ; ++k) {
                const bool ok = inmap && q + 4 * (kh * KQ + k) < NP;
                xv[k] = ok ? xv[k] : zero4;
                s1 += (xv[k][0] + xv[k][1]) + (xv[k][2] + xv[k][3]);
            }
            s1 += __shfl_xor(s1, 1);
            s1 += __shfl_xor(s1, 2);
            if constexpr (KP > 1) {  // the slices' partial sums meet in a fixed order
                if (q == 0) stat[0][(kh * 4 + t) * 16 + tl] = s1;
                __syncthreads();
                s1 = 0.f;
#pragma unroll
                for (int kk = 0; kk < KP; ++kk) s1 += stat[0][(kk * 4 + t) * 16 + tl];
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
                if (q == 0) stat[1][(kh * 4 + t) * 16 + tl] = q2;
                __syncthreads();
                q2 = 0.f;
#pragma unroll
                for (int kk = 0; kk < KP; ++kk) q2 += stat[1][(kk * 4 + t) * 16 + tl];
            }
            const float var = (q2 - npad * mean * mean) * inv_c;
            const float rstd = rsqrtf(fmaxf(var, 0.f) + p.eps);
#pragma unroll
            for (int k = 0; k < KQ; ++k) {
                const int pi = q + 4 * (kh * KQ + k);
                const bool has = pi < NP;
                const f32x4 wa = *reinterpret_cast<const f32x4*>(p.ln_w + 4 * (has ? pi : 0)), ba = *reinterpret_cast<const f32x4*>(p.ln_b + 4 * (has ? pi : 0));
                const float keep = (inmap && has) ? 1.f : 0.f;  // tokens outside the map are exact zeros AFTER the LayerNorm; so is the image beyond the row
                reinterpret_cast<uint2*>(smem + (t * KS + (pi >> 3)) * 64 + ((pi & 7) >> 1) * 16 + tl)[pi & 1] = i2r_pack4<DT>(((xv[k] - mean) * rstd * wa + ba) * keep);
            }
        }
    }
    __syncthreads();

    // ---- q / k / v of one head over the four token tiles: 3 dim blocks x 4 tiles accumulators, weight fragments one k-step ahead ----
    auto project = [&](const int head, const int part, auto swap_c, f32x4 (&acc)[4][3]) {
        constexpr bool SWAP = decltype(swap_c)::value;
        const float* bsrc = p.bqkv + (head * 3 + part) * 48;
#pragma unroll
        for (int db = 0; db < 3; ++db) {
            f32x4 b4;
            if constexpr (SWAP) {  // D = [token rows][dim column li]
                const float b = bsrc[16 * db + li];
                b4 = (f32x4){b, b, b, b};
            } else {               // D = [dim rows 4g + r][token column]
                b4 = *reinterpret_cast<const f32x4*>(bsrc + 16 * db + 4 * g);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t][db] = b4;
        }
        const f32x4* wsrc = p.wqkv + (size_t)(head * 3 + part) * KS * 192 + lane;
        f32x4 wn[3];
#pragma unroll
        for (int db = 0; db < 3; ++db) wn[db] = wsrc[db * 64];
        // (a ROLLED loop: fully unrolled, the scheduler hoists the fragment loads of all k-steps to the top and spills)
#pragma unroll 1
        for (int s = 0; s < KS; ++s) {
            f32x4 wf[3];
#pragma unroll
            for (int db = 0; db < 3; ++db) wf[db] = wn[db];
            const int sn = s + 1 < KS ? s + 1 : s;  // (last step: a harmless re-fetch instead of a branch)
#pragma unroll
            for (int db = 0; db < 3; ++db) wn[db] = wsrc[sn * 192 + db * 64];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f32x4 xt = smem[(t * KS + s) * 64 + lane];
#pragma unroll
                for (int db = 0; db < 3; ++db)
                    acc[t][db] = SWAP ? mfma32_lp<DT>(xt, wf[db], acc[t][db]) : mfma32_lp<DT>(wf[db], xt, acc[t][db]);
            }
        }
    };

    uint2 opk[HPW][4][3];  // O^T of this wave's heads, packed: [head][query tile][dim block] = dims 16 db + 4g + r of query li
#pragma unroll
    for (int j = 0; j < HPW; ++j) {
        const int head = wave * HPW + j;
        f32x4 qpk[4][2], kpk[4][2], vpk[3][2];
        if constexpr (KS <= 10) {
            // Narrow branches (C = 78 / 156: 3 / 5 k-steps per projection): the 3 KS (part, k-step) units of a head form ONE fully unrolled
            // sequence whose weight fragments come through a ring fetched RD units ahead ACROSS the part boundaries -- with one k-step
            // of look-ahead inside a rolled loop per part (below) every step waited for an L2 round trip that 12 matrix instructions
            // do not cover.  A fence per unit keeps the ring a ring (unfenced, the scheduler hoists all fetches to the top and spills).
            constexpr int NU = 3 * KS, RD = RDQ;
            if (j > 0) {
#pragma unroll
                for (int n = 0; n < RD; ++n) ufetch(head, n);
            }
#pragma unroll
            for (int pi = 0; pi < 3; ++pi) {
                const int part = 2 - pi;
                f32x4 acc[4][3];
                const float* bsrc = p.bqkv + (head * 3 + part) * 48;
#pragma unroll
                for (int db = 0; db < 3; ++db) {
                    f32x4 b4;
                    if (part == 2) {  // V: D = [token rows][dim column li]
                        const float b = bsrc[16 * db + li];
                        b4 = (f32x4){b, b, b, b};
                    } else {          // Q^T, K^T: D = [dim rows 4g + r][token column]
                        b4 = *reinterpret_cast<const f32x4*>(bsrc + 16 * db + 4 * g);
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t][db] = b4;
                }
#pragma unroll
                for (int sk = 0; sk < KS; ++sk) {
                    const int n = pi * KS + sk;
                    if (n + RD < NU) ufetch(head, n + RD);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const f32x4 xt = smem[(t * KS + sk) * 64 + lane];
#pragma unroll
                        for (int db = 0; db < 3; ++db)
                            acc[t][db] = part == 2 ? mfma32_lp<DT>(xt, wr[n % (RD + 1)][db], acc[t][db]) : mfma32_lp<DT>(wr[n % (RD + 1)][db], xt, acc[t][db]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (part == 2) {
#pragma unroll
                    for (int db = 0; db < 3; ++db) { vpk[db][0] = pack8<DT>(acc[0][db], acc[1][db]); vpk[db][1] = pack8<DT>(acc[2][db], acc[3][db]); }
                } else if (part == 1) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) { kpk[t][0] = pack8<DT>(acc[t][0], acc[t][1]); kpk[t][1] = pack8<DT>(acc[t][2], zero4); }
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) { qpk[t][0] = pack8<DT>(acc[t][0], acc[t][1]); qpk[t][1] = pack8<DT>(acc[t][2], zero4); }
                }
            }
        } else {
        // (v first, q last: the accumulators of the widest phase then sit beside the fewest packed operands)
        {
            f32x4 acc[4][3];
            project(head, 2, std::true_type{}, acc);
#pragma unroll
            for (int db = 0; db < 3; ++db) { vpk[db][0] = pack8<DT>(acc[0][db], acc[1][db]); vpk[db][1] = pack8<DT>(acc[2][db], acc[3][db]); }
        }
        {
            f32x4 acc[4][3];
            project(head, 1, std::false_type{}, acc);
#pragma unroll
            for (int t = 0; t < 4; ++t) { kpk[t][0] = pack8<DT>(acc[t][0], acc[t][1]); kpk[t][1] = pack8<DT>(acc[t][2], zero4); }
        }
        {
            f32x4 acc[4][3];
            project(head, 0, std::false_type{}, acc);
#pragma unroll
            for (int t = 0; t < 4; ++t) { qpk[t][0] = pack8<DT>(acc[t][0], acc[t][1]); qpk[t][1] = pack8<DT>(acc[t][2], zero4); }
        }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            // S^T[key][query] of query tile t against the 64 keys, softmax over the 49 real ones (base 2)
            f32x4 st[4];
#pragma unroll
            for (int kf = 0; kf < 4; ++kf) st[kf] = mfma32_lp<DT>(kpk[kf][0], qpk[t][0], zero4);
#pragma unroll
            for (int kf = 0; kf < 4; ++kf) st[kf] = mfma32_lp<DT>(kpk[kf][1], qpk[t][1], st[kf]);
            float mx = -__builtin_inff();
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (16 * kf + 4 * g + r >= 49) st[kf][r] = -__builtin_inff();
                    mx = fmaxf(mx, st[kf][r]);
                }
            mx = i2r_xmax4(mx);
            float sum = 0.f;
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    st[kf][r] = __builtin_amdgcn_exp2f(st[kf][r] - mx);
                    sum += st[kf][r];
                }
            const f32x4 pB0 = pack8<DT>(st[0], st[1]), pB1 = pack8<DT>(st[2], st[3]);
            const float inv = 1.f / i2r_xsum4(sum);
            f32x4 o[3];
#pragma unroll
            for (int db = 0; db < 3; ++db) o[db] = mfma32_lp<DT>(vpk[db][0], pB0, zero4);
#pragma unroll
            for (int db = 0; db < 3; ++db) o[db] = mfma32_lp<DT>(vpk[db][1], pB1, o[db]);
#pragma unroll
            for (int db = 0; db < 3; ++db) opk[j][t][db] = i2r_pack4<DT>(o[db] * inv);
        }
    }

    // ---- O^T of all heads -> LDS as the out-proj's B operand: 16-dim block bi = 3 head + db is half bi & 1 of k-step bi >> 1 ----
    __syncthreads();  // every wave has read its last LayerNorm-ed column: the area becomes the exchange buffer
#pragma unroll
    for (int j = 0; j < HPW; ++j)
#pragma unroll
        for (int db = 0; db < 3; ++db) {
            const int bi = (wave * HPW + j) * 3 + db;
#pragma unroll
            for (int t = 0; t < 4; ++t)
                reinterpret_cast<uint2*>(smem + (t * OSG + (bi >> 1)) * 64 + lane)[bi & 1] = opk[j][t][db];
        }
    __syncthreads();

    // ---- out-proj + bias + residual: this wave's token-tile pair, output blocks (wave >> 1) + i NW / 2.  The D fragments (lane (li, g):
    //      features 4g + r of token li) are re-numbered with one ds_bpermute per register so that lane n holds piece n & 3 of token
    //      n >> 2: residual loads and stores are quad-coalesced like the LayerNorm's loads ----
    const int tp = wave & 1;
    const int tl = lane >> 2, pq = lane & 3;
    const int srcl = tl + 16 * pq;       // the lane that holds this lane's new data
    size_t rowT[2];
    bool inT[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const int tk = 16 * (2 * tp + tt) + tl;
        const int ty = tk / 7, tx = tk - ty * 7;
        const int y = wy * 7 + ty - p.pad_top, x = wx * 7 + tx - p.pad_left;
        inT[tt] = tk < 49 && y >= 0 && y < p.h && x >= 0 && x < p.w;
        rowT[tt] = (((size_t)img * p.h + (inT[tt] ? y : 0)) * p.w + (inT[tt] ? x : 0)) * cs + 4 * pq;
    }
    constexpr int UN = (CB + NW / 2 - 1) / (NW / 2);  // output blocks per wave (the last one may not exist)
    constexpr int UC = UN < 5 ? UN : 5;               // ... in chunks of at most five (accumulators + fragments + residuals in registers)
#pragma unroll
    for (int c0 = 0; c0 < UN; c0 += UC) {
        f32x4 acc[UC][2], xres[UC][2];
        int ob[UC];
#pragma unroll
        for (int i = 0; i < UC; ++i) {
            const int o_ = (wave >> 1) + (c0 + i) * (NW / 2);
            ob[i] = o_ < CB ? o_ : CB - 1;  // (a block that does not exist: recomputed, not stored)
            const f32x4 b = *reinterpret_cast<const f32x4*>(p.bo + 16 * ob[i] + 4 * g);
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                acc[i][tt] = b;
                xres[i][tt] = *reinterpret_cast<const f32x4*>(p.x + rowT[tt] + 16 * ob[i]);  // (clamped row: always readable)
            }
        }
        if constexpr (OSG <= 12) {  // narrow branches: all k-steps unrolled, fragments through a ring two steps ahead (as the projections)
            constexpr int RD = 2;
            f32x4 wr[RD + 1][UC];
            auto ofetch = [&](const int sn) {
#pragma unroll
                for (int i = 0; i < UC; ++i) wr[sn % (RD + 1)][i] = p.wo[((size_t)ob[i] * OSG + sn) * 64 + lane];
            };
#pragma unroll
            for (int sn = 0; sn < RD; ++sn) ofetch(sn);
#pragma unroll
            for (int sk = 0; sk < OSG; ++sk) {
                if (sk + RD < OSG) ofetch(sk + RD);
                __builtin_amdgcn_sched_barrier(0);
                const f32x4 b0 = smem[((2 * tp) * OSG + sk) * 64 + lane], b1 = smem[((2 * tp + 1) * OSG + sk) * 64 + lane];
#pragma unroll
                for (int i = 0; i < UC; ++i) {
                    acc[i][0] = mfma32_lp<DT>(wr[sk % (RD + 1)][i], b0, acc[i][0]);
                    acc[i][1] = mfma32_lp<DT>(wr[sk % (RD + 1)][i], b1, acc[i][1]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
        f32x4 wn[UC];
#pragma unroll
        for (int i = 0; i < UC; ++i) wn[i] = p.wo[(size_t)ob[i] * OSG * 64 + lane];
#pragma unroll 1
        for (int s = 0; s < OSG; ++s) {
            f32x4 wv[UC];
#pragma unroll
            for (int i = 0; i < UC; ++i) wv[i] = wn[i];
            const int sn = s + 1 < OSG ? s + 1 : s;
#pragma unroll
            for (int i = 0; i < UC; ++i) wn[i] = p.wo[((size_t)ob[i] * OSG + sn) * 64 + lane];
            const f32x4 b0 = smem[((2 * tp) * OSG + s) * 64 + lane], b1 = smem[((2 * tp + 1) * OSG + s) * 64 + lane];
#pragma unroll
            for (int i = 0; i < UC; ++i) {
                acc[i][0] = mfma32_lp<DT>(wv[i], b0, acc[i][0]);
                acc[i][1] = mfma32_lp<DT>(wv[i], b1, acc[i][1]);
            }
        }
        }
#pragma unroll
        for (int i = 0; i < UC; ++i) {
            const bool exists = (wave >> 1) + (c0 + i) * (NW / 2) < CB;  // (wave-uniform)
            if (!exists) continue;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = __shfl(acc[i][tt][r], srcl);  // (ds_bpermute_b32; NOT __builtin_bit_cast on a vector element: hipcc 7.2 folds the four into one)
                if (inT[tt]) *reinterpret_cast<f32x4*>(p.out + rowT[tt] + 16 * ob[i]) = v + xres[i][tt];
            }
        }
    }
}

template <int DT>
bool launch(const I2rAttnK& k, int cs, int heads, long long nblk, hipStream_t stream) {
    const dim3 grid((unsigned)((nblk + 7) / 8 * 8));
    if (cs == 80 && heads == 2) i2r_launch((hrt_attn_head_k<DT, 5, 2, 1, 3>), grid, dim3(128), 0, stream, k);
    else if (cs == 160 && heads == 4) i2r_launch((hrt_attn_head_k<DT, 10, 4, 1, 3>), grid, dim3(256), 0, stream, k);
    else if (cs == 320 && heads == 8) i2r_launch((hrt_attn_head_k<DT, 20, 8, 1, 2>), grid, dim3(512), 0, stream, k);
    else if (cs == 624 && heads == 16) i2r_launch((hrt_attn_head_k<DT, 39, 16, 2, 2>), grid, dim3(512), 0, stream, k);
    else return false;
    return true;
}

}  // namespace

bool i2r_attn_head_launch(const I2rAttnK& k, int dtype, int cs, int heads, long long nblk, hipStream_t stream) {
    return dtype == 1 ? launch<1>(k, cs, heads, nblk, stream) : launch<2>(k, cs, heads, nblk, stream);
}
