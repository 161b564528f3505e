// HRFormer-B transformer block, MLP half, FUSED for the 16-bit modes (BASELINE configs 4-5):
//     x2 = x1 + GELU(BN3(fc2( GELU(BN2(DW3x3( GELU(BN1(fc1( LayerNorm2(x1) ))) ))) )))
// (reference lib/models/hrformer.py:1237 + MlpDWBN.forward :1094-1119; the BatchNorms are folded into the conv weights by the host)
// in ONE launch per block, without the 4C-wide hidden tensor (the largest map of the block) ever reaching HBM.
//
// The kernel is bound by the VALU (three GELUs + nine depth-wise FMAs per hidden element against 2 x 16 MFMA k-steps), i.e. by issue
// slots and by how well a wave hides its own latencies -- not by the matrix pipe and not by HBM.  Structure (round 3):
//
//   * One workgroup = one 8 x 6 sub-tile of output pixels (three 16-pixel fragments) + its halo of 1: 10 x 8 = 80 pixels = exactly five
//     fragments for fc1 (1.67x recompute; every map of the HRFormer-B pyramid is a multiple of 8 x 6).
//   * The hidden dimension (4C padded to HBT = 4 CB blocks of 16 channels) is split over the NBG waves of the workgroup: wave w owns
//     blocks w, w + NBG, ...  A depth-wise conv never mixes channels, so for ITS blocks a wave does everything itself -- fc1 on the
//     five halo fragments (A = weight fragment, B = LayerNorm-ed pixel columns held in registers), + bias, GELU, fp32 to a wave-PRIVATE
//     LDS tile H[channel quad][halo row][halo column][4]; depth-wise 3x3 + bias + GELU for the three output pixels of each lane (a
//     horizontal strip: 15 LDS reads for 27 taps); packed, that is the B operand (k = hidden channel) of fc2, accumulated into
//     3 x CB fragments.  No workgroup barrier inside the loop: waves drift apart and fill each other's stalls.
//   * The loop is software-pipelined inside a wave: fc1's MFMAs of block b+1 are issued before the depth-wise phase of block b, the weight
//     fragments of block b+2 / b+1 are fetched right behind the MFMAs that free their registers.
//   * The waves' partial fc2 sums meet once, through LDS, in a fixed order (bit-identical replays): output block ob is finished by wave
//     ob % NBG: + bias, GELU, + x1 (fp32 residual), fp32 store.
// LDS layout of H (floats): quad g at g*448, halo row hy at hy*40, halo column hx at hx*4: the 16-lane groups of ds_read_b128 /
// the 8-lane groups of ds_write_b128 then hit 64 distinct banks for every tap (checked exhaustively, tools/lds_layout.py).
// x / out are the fp32 residual stream [n, h, w, cs]; operands bf16 / f16, accumulation fp32.
// Round 4: both GEMMs on v_mfma_f32_16x16x32 -- fc1 over 32-channel k-steps of the LayerNorm-ed pixel columns, fc2 over PAIRS of hidden
// blocks (the depth-wise results of blocks 2p and 2p + 1 packed side by side are one 32-deep B operand; the host stores W2's columns in
// that slot order, as for the attention kernel's out-proj) -- and the per-element masks of the prologue / epilogue are gone: pad
// channels are exact zeros on both sides (zero weights and biases, GELU(0) = 0), the LayerNorm variance subtracts their (cs - c)
// mean^2 instead of masking them.
#include "i2r_hrformer_mlp.h"

namespace {

#ifndef I2R_XCD_BAND
#define I2R_XCD_BAND 1   // A/B knob: 0 = tiles in plain blockIdx order
#endif
// fc1 of the next hidden block is issued AFTER the depth-wise phase of the current one.  Issuing it before (round 3's in-wave software
// pipeline: MFMAs under the vector ALU work) keeps five more accumulators live across that phase: at C = 78 that spills 77 registers
// (78.8 vs 36 us); at C = 156 it fits (467 registers, one wave per SIMD either way) and is 0.7 us faster stand-alone, but the whole
// forward measured 2-3 % SLOWER with it (3.35-3.40 vs 3.23-3.28 ms, configs 4 / 5 alike): late everywhere.
#ifndef I2R_MLP_FC1_LATE
#define I2R_MLP_FC1_LATE(CB) 1
#endif
// which kernel `variant` 0 selects for cs = 80 / 160 (measured): 1 = fc2 accumulated per wave, 2 = by output-block ownership
#ifndef I2R_MLP_DEFAULT_VARIANT
#define I2R_MLP_DEFAULT_VARIANT(cs) 1
#endif
constexpr int TY = 8, TX = 6;                    // output sub-tile (rows x columns)
constexpr int HY = TY + 2, HX = TX + 2;          // halo grid 10 x 8 = 80 pixels = 5 fragments (two halo rows each)
constexpr int NF = HY * HX / 16, NPF = TY * TX / 16;
constexpr int H_ROW = 40, H_QUAD = 448;          // LDS strides of H in floats (see the header)
static_assert(NF == 5 && NPF == 3, "sub-tile geometry");

template <int DT, int CB, int NBG>
__global__ __launch_bounds__(NBG * 64, NBG == 2 ? 2 : 1) void hrt_mlp_block_k(const MlpK p) {
    constexpr int cs = CB * 16, KS = (cs + 31) / 32, HBT = 4 * CB, HID = HBT * 16, NB = HBT / NBG, NP = NB / 2;  // NB hidden blocks = NP pairs per wave
    static_assert(HBT % (2 * NBG) == 0, "whole pairs of hidden blocks per wave");
    constexpr bool LATE = I2R_MLP_FC1_LATE(CB);
    constexpr int WD_WAVE = NB * 160;                           // floats: per block [10 = 9 taps + bias][16 channels]
    constexpr int H_WAVE = 4 * H_QUAD;                          // floats
    constexpr int X_FLOATS = NF * KS * 64 * 4;                  // packed pixel columns, 16 bytes per lane and (fragment, k-step)
    constexpr int R1 = X_FLOATS > NBG * H_WAVE ? X_FLOATS : NBG * H_WAVE;
    constexpr int RED_FLOATS = (NBG == 2 ? CB : (CB + 1) / 2) * (NBG - 1) * NPF * 64 * 4;  // partial fc2 sums handed to the finishing wave
    constexpr int SMEM = NBG * WD_WAVE + R1 > RED_FLOATS ? NBG * WD_WAVE + R1 : RED_FLOATS;
    __shared__ __attribute__((aligned(16))) float smem[SMEM];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, g = lane >> 4;  // (wave: scalar)
    float* const Wd = smem + wave * WD_WAVE;                    // this wave's depth-wise weights
    float* const Hs = smem + NBG * WD_WAVE + wave * H_WAVE;     // this wave's hidden tile
    f32x4* const Xs = reinterpret_cast<f32x4*>(smem + NBG * WD_WAVE);  // (prologue only; aliases the H tiles)
    int bid = I2R_XCD_BAND ? xcd_band_item(blockIdx.x, p.total) : (int)blockIdx.x;  // (workgroup-uniform)
    if (bid < 0 || bid >= p.total) return;
    const int sx = bid % p.tiles_x; bid /= p.tiles_x;
    const int sy = bid % p.tiles_y;
    const int img = bid / p.tiles_y;
    const int y0 = sy * TY - 1, x0 = sx * TX - 1;               // image coordinate of halo pixel (0, 0)
    // hidden block b (0 .. NB-1) of this wave: pair wave + NBG (b >> 1), half b & 1
    auto hblock = [&](int b) { return 2 * (wave + NBG * (b >> 1)) + (b & 1); };

    // ---- LayerNorm 2 of halo fragments wave, wave + NBG, ...: packed B operands -> Xs[fragment][k-step][lane].  Their row loads go out
    //      FIRST and unconditionally (clamped addresses, zeroed by `keep` afterwards), the depth-wise weights behind them ----
    float hinf[NF];  // 1 = halo pixel (16 f + li) lies inside the image
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int y = y0 + 2 * f + (li >> 3), x = x0 + (li & 7);
        hinf[f] = (y >= 0 && y < p.h && x >= 0 && x < p.w) ? 1.f : 0.f;
    }
    // Round 5: the rows are read QUAD-COALESCED -- lane n takes halo pixel n >> 2 of the fragment and the 16-byte pieces (n & 3) + 4 k of
    // its row, so a quad covers 64 contiguous bytes (one row per lane costs four times the addresser cycles, DESIGN.md "quad rule");
    // statistics in-lane + over the quad; piece pi lands in half pi & 1 of lane (g = (pi & 7) >> 1, li = pixel) of k-step pi >> 3.
    constexpr int NFW = (NF + NBG - 1) / NBG;  // halo fragments per wave
    constexpr int NPC = cs / 4, KQ = 2 * KS;     // 16-byte pieces of a row / pieces per lane (the whole operand image: beyond the row, zeros)
    const int tl = lane >> 2, q4 = lane & 3;
    f32x4 xv[NFW][KQ];
    bool pin[NFW];
#pragma unroll
    for (int i = 0; i < NFW; ++i) {
        const int f = wave + NBG * i;
        if (f >= NF) continue;  // (scalar)
        const int y = y0 + 2 * f + (tl >> 3), x = x0 + (tl & 7);
        pin[i] = y >= 0 && y < p.h && x >= 0 && x < p.w;
        const float* row = p.x + (((size_t)img * p.h + (pin[i] ? y : 0)) * p.w + (pin[i] ? x : 0)) * cs;
#pragma unroll
        for (int k = 0; k < KQ; ++k) {
            const int pi = q4 + 4 * k;
            xv[i][k] = *reinterpret_cast<const f32x4*>(row + 4 * (pi < NPC ? pi : 0));
        }
    }
    // ---- depth-wise weights + bias of this wave's blocks -> its LDS region: [block][tap | bias][16] ----
#pragma unroll
    for (int i0 = 0; i0 < NB * 40; i0 += 64) {
        const int i = i0 + lane;
        if (i < NB * 40) {
            const int blk = i / 40, r = i - blk * 40, tap = r >> 2, q = r & 3;
            const int hb = hblock(blk);
            const f32x4 v = tap < 9 ? *reinterpret_cast<const f32x4*>(p.wdw + tap * HID + hb * 16 + 4 * q)
                                    : *reinterpret_cast<const f32x4*>(p.bdw + hb * 16 + 4 * q);
            *reinterpret_cast<f32x4*>(Wd + blk * 160 + tap * 16 + 4 * q) = v;
        }
    }
    const float npad = (float)(cs - p.c);  // zero pad channels inside the loaded rows: each adds mean^2 to the sum of squares
    const float inv_c = __builtin_amdgcn_rcpf((float)p.c);
#pragma unroll
    for (int i = 0; i < NFW; ++i) {
        const int f = wave + NBG * i;
        if (f >= NF) continue;  // (scalar)
        float s1 = 0.f;
#pragma unroll
        for (int k = 0; k < KQ; ++k) {
            const bool ok = pin[i] && q4 + 4 * k < NPC;
            xv[i][k] = ok ? xv[i][k] : (f32x4){0.f, 0.f, 0.f, 0.f};
            s1 += (xv[i][k][0] + xv[i][k][1]) + (xv[i][k][2] + xv[i][k][3]);
        }
        s1 += __shfl_xor(s1, 1);
        s1 += __shfl_xor(s1, 2);
        const float mean = s1 * inv_c;
        float q2 = 0.f;
#pragma unroll
        for (int k = 0; k < KQ; ++k) {
            const float keep = q4 + 4 * k < NPC ? 1.f : 0.f;  // (compile-time 1 except beyond the row)
            const f32x4 d = (xv[i][k] - mean) * keep;
            q2 += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
        q2 += __shfl_xor(q2, 1);
        q2 += __shfl_xor(q2, 2);
        const float var = (q2 - npad * mean * mean) * inv_c;
        const float rstd = rsqrtf(fmaxf(var, 0.f) + p.eps);
#pragma unroll
        for (int k = 0; k < KQ; ++k) {
            const int pi = q4 + 4 * k;
            const bool has = pi < NPC;
            const f32x4 wa = *reinterpret_cast<const f32x4*>(p.ln_w + 4 * (has ? pi : 0)), ba = *reinterpret_cast<const f32x4*>(p.ln_b + 4 * (has ? pi : 0));
            const float keep = (pin[i] && has) ? 1.f : 0.f;  // outside the image: zero columns; beyond the row: zeros (pad channels: ln_w = ln_b = 0)
            reinterpret_cast<uint2*>(Xs + (f * KS + (pi >> 3)) * 64 + ((pi & 7) >> 1) * 16 + tl)[pi & 1] = pack4<DT>(((xv[i][k] - mean) * rstd * wa + ba) * keep);
        }
    }
    __syncthreads();
    f32x4 xn[NF][KS];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int s = 0; s < KS; ++s) xn[f][s] = Xs[(f * KS + s) * 64 + lane];
    __syncthreads();  // everyone holds the pixel columns in registers: the region becomes the H tiles

    // this lane's three output pixels: row oy, columns 3 xb .. 3 xb + 2 of the sub-tile
    const int oy = li & 7, xb = li >> 3;
    float* const hwr = Hs + g * H_QUAD + (li >> 3) * H_ROW + (li & 7) * 4;   // fc1 result of halo fragment 0 (fragment f: + 2 f rows)
    const float* const hrd = Hs + g * H_QUAD + oy * H_ROW + (3 * xb) * 4;    // top-left tap of the strip
    const float* const wrd = Wd + 4 * g;

    const GeluC gk;
    f32x4 acc[NPF][CB];
#pragma unroll
    for (int pf = 0; pf < NPF; ++pf)
#pragma unroll
        for (int ob = 0; ob < CB; ++ob) acc[pf][ob] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 w1f[KS], w2f[CB];
    f32x4 b1v;
    auto fetch1 = [&](int b) {
        const int hb = hblock(b);
#pragma unroll
        for (int s = 0; s < KS; ++s) w1f[s] = p.w1[(hb * KS + s) * 64 + lane];
        b1v = *reinterpret_cast<const f32x4*>(p.b1 + hb * 16 + 4 * g);
    };
    auto fetch2 = [&](int pr) {  // pair pr of this wave
        const int gp = wave + NBG * pr;
#pragma unroll
        for (int ob = 0; ob < CB; ++ob) w2f[ob] = p.w2[(ob * (HBT / 2) + gp) * 64 + lane];
    };
    f32x4 a[NF];
    auto fc1 = [&]() {  // hidden block in w1f / b1v, all five halo fragments (k-step-major: five independent accumulators)
#pragma unroll
        for (int f = 0; f < NF; ++f) a[f] = b1v * hinf[f];
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int f = 0; f < NF; ++f) a[f] = mfma32_lp<DT>(w1f[s], xn[f][s], a[f]);
    };
    auto store_h = [&]() {
#pragma unroll
        for (int f = 0; f < NF; ++f) *reinterpret_cast<f32x4*>(hwr + 2 * f * H_ROW) = gelu4<4>(a[f], gk);
    };

    // depth-wise 3x3 + bias + GELU of block b (three output pixels x 4 channels per lane) -> packed 16-bit halves of the fc2 B operand
    auto dwconv = [&](int b, uint2 (&dp)[NPF]) {
        const float* const wch = wrd + b * 160;
        f32x4 d[NPF];
        {
            const f32x4 bias = *reinterpret_cast<const f32x4*>(wch + 9 * 16);
#pragma unroll
            for (int pf = 0; pf < NPF; ++pf) d[pf] = bias;
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            f32x4 t[5], wv[3];
#pragma unroll
            for (int j = 0; j < 5; ++j) t[j] = *reinterpret_cast<const f32x4*>(hrd + ky * H_ROW + j * 4);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) wv[kx] = *reinterpret_cast<const f32x4*>(wch + (ky * 3 + kx) * 16);
#pragma unroll
            for (int pf = 0; pf < NPF; ++pf)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) d[pf] = t[pf + kx] * wv[kx] + d[pf];
        }
#pragma unroll
        for (int pf = 0; pf < NPF; ++pf) dp[pf] = pack4<DT>(gelu4<4>(d[pf], gk));
    };

    // software pipeline over the wave's blocks: fc1 of block b + 1 goes to the matrix pipe before the depth-wise phase of block b; a
    // PAIR of blocks feeds one 32-deep fc2 step (slot 8g + 4h + r <- hidden channel 16 (2 pair + h) + 4g + r)
    fetch1(0);
    fetch2(0);
    fc1();
    fetch1(1);
    store_h();
#pragma unroll 1
    for (int pr = 0; pr < NP; ++pr) {
        uint2 d0[NPF], d1[NPF];
        __builtin_amdgcn_wave_barrier();  // (H of block 2 pr is complete in program order; LDS executes a wave's accesses in order)
        if constexpr (!LATE) fc1();  // block 2 pr + 1 goes to the matrix pipe first
        dwconv(2 * pr, d0);
        if constexpr (LATE) fc1();
        fetch1(2 * pr + 2 < NB ? 2 * pr + 2 : 2 * pr + 1);  // (last round: a harmless re-fetch instead of a branch)
        store_h();                        // block 2 pr + 1's hidden tile (block 2 pr's taps have all been read)
        __builtin_amdgcn_wave_barrier();
        if constexpr (!LATE) { if (pr + 1 < NP) fc1(); }  // block 2 pr + 2 (wave-uniform branch)
        dwconv(2 * pr + 1, d1);
#pragma unroll
        for (int ob = 0; ob < CB; ++ob)
#pragma unroll
            for (int pf = 0; pf < NPF; ++pf) acc[pf][ob] = mfma32_lp<DT>(w2f[ob], join8(d0[pf], d1[pf]), acc[pf][ob]);
        fetch2(pr + 1 < NP ? pr + 1 : pr);
        if (pr + 1 < NP) {
            if constexpr (LATE) fc1();
            fetch1(2 * pr + 3);
            store_h();
        }
    }

    // ---- the waves' partial sums meet: output block ob is finished by wave ob % NBG.  Round 5: the finished D fragment (lane (li, g):
    //      channels 4g + r of pixel li) is re-numbered with one ds_bpermute per register so that lane n holds piece n & 3 of pixel n >> 2:
    //      residual loads and stores are quad-coalesced ----
    const int srcl = tl + 16 * q4;
    const int gy = sy * TY + (tl & 7);
    const bool rowin = gy < p.h;
    size_t prow[NPF];
    bool oin[NPF];
#pragma unroll
    for (int pf = 0; pf < NPF; ++pf) {
        const int gx = sx * TX + 3 * (tl >> 3) + pf;
        oin[pf] = rowin && gx < p.w;
        prow[pf] = (((size_t)img * p.h + (rowin ? gy : 0)) * p.w + (gx < p.w ? gx : 0)) * cs + 4 * q4;
    }
    // (partials of OBC output blocks at a time, so the exchange area fits the loop's LDS footprint)
    constexpr int OBC = NBG == 2 ? CB : (CB + 1) / 2;
    static_assert(OBC * (NBG - 1) * NPF * 64 * 4 <= SMEM, "exchange area");
    f32x4* const Red = reinterpret_cast<f32x4*>(smem);
#pragma unroll
    for (int ob0 = 0; ob0 < CB; ob0 += OBC) {
        __syncthreads();  // every wave is done with its H tile and weights / with the previous chunk: the buffer is the exchange area
#pragma unroll
        for (int ob = ob0; ob < ob0 + OBC && ob < CB; ++ob) {
            const int owner = ob % NBG;
            if (owner == wave) continue;  // (wave-uniform)
            const int slot = (wave - owner - 1 + NBG) % NBG;  // 0 .. NBG-2
#pragma unroll
            for (int pf = 0; pf < NPF; ++pf) Red[(((ob - ob0) * (NBG - 1) + slot) * NPF + pf) * 64 + lane] = acc[pf][ob];
        }
        __syncthreads();
#pragma unroll
        for (int ob = ob0; ob < ob0 + OBC && ob < CB; ++ob) {
            if (ob % NBG != wave) continue;
            const f32x4 b = *reinterpret_cast<const f32x4*>(p.b2 + 16 * ob + 4 * g);
#pragma unroll
            for (int pf = 0; pf < NPF; ++pf) {
                f32x4 v = acc[pf][ob];
#pragma unroll
                for (int s = 1; s < NBG; ++s)  // fixed order: waves owner + 1, owner + 2, ... (mod NBG)
                    v += Red[(((ob - ob0) * (NBG - 1) + (s - 1)) * NPF + pf) * 64 + lane];
                // (pad channels: zero W2 rows and bias -> GELU(0) = 0 exactly, + the residual's zero: no mask)
                v = gelu4<5>(v + b, gk);
                f32x4 vn;
#pragma unroll
                for (int e = 0; e < 4; ++e) vn[e] = __shfl(v[e], srcl);  // (ds_bpermute_b32; all lanes take part)
                if (!oin[pf]) continue;
                const f32x4 xres = *reinterpret_cast<const f32x4*>(p.x + prow[pf] + 16 * ob);
                *reinterpret_cast<f32x4*>(p.out + prow[pf] + 16 * ob) = vn + xres;
            }
        }
    }
}

}  // namespace

extern "C" int i2r_hrt_mlp_block(const float* x, float* out, const float* ln_w, const float* ln_b, const void* w1, const float* b1,
                                 const float* wdw, const float* bdw, const void* w2, const float* b2, int32_t n_img, int32_t h, int32_t w,
                                 int32_t c, int32_t cs, int32_t hidden_pad, float eps, int32_t dtype, int32_t variant, void* stream) {
    I2R_CHECK_ARG(x && out && x != out && ln_w && ln_b && w1 && b1 && wdw && bdw && w2 && b2, "i2r_hrt_mlp_block: null pointer / out aliases x");
    I2R_CHECK_ARG(dtype == 1 || dtype == 2, "i2r_hrt_mlp_block: dtype %d (1 bf16, 2 f16; the fp32 path is i2r_layernorm + i2r_conv + i2r_dwconv3x3)", dtype);
    I2R_CHECK_ARG((cs == 80 || cs == 160 || cs == 320) && c <= cs && c > cs - 16 && hidden_pad >= 4 * c && hidden_pad == 4 * cs,
                  "i2r_hrt_mlp_block: c=%d cs=%d hidden_pad=%d (built for the HRFormer-B branches 78 / 156 / 312; hidden padded to 4 cs)", c, cs, hidden_pad);
    if (variant == 0) variant = cs == 320 ? 2 : I2R_MLP_DEFAULT_VARIANT(cs);
    I2R_CHECK_ARG((variant == 1 && cs <= 160) || variant == 2,
                  "i2r_hrt_mlp_block: variant %d for cs=%d (1 = fc2 accumulated per wave: 78 / 156; 2 = fc2 by output-block ownership)", variant, cs);
    I2R_CHECK_ARG((long long)n_img * h * w * cs < (1ll << 31), "i2r_hrt_mlp_block: tensor too large");
    MlpK k;
    k.x = x; k.out = out; k.ln_w = ln_w; k.ln_b = ln_b; k.w1 = (const f32x4*)w1; k.b1 = b1; k.wdw = wdw; k.bdw = bdw; k.w2 = (const f32x4*)w2; k.b2 = b2;
    k.n_img = n_img; k.h = h; k.w = w; k.c = c; k.eps = eps;
    k.tiles_y = (h + TY - 1) / TY; k.tiles_x = (w + TX - 1) / TX;
    const long long nblk = (long long)n_img * k.tiles_y * k.tiles_x;
    I2R_CHECK_ARG(nblk > 0 && nblk < (1ll << 30), "i2r_hrt_mlp_block: grid");
    k.total = (int)nblk;
    const dim3 grid((unsigned)((nblk + 7) / 8 * 8));
    if (variant == 2) {
        const bool ok = i2r_mlp_wide_launch(k, dtype, cs, nblk, (hipStream_t)stream);
        I2R_CHECK_ARG(ok, "i2r_hrt_mlp_block: no output-block-ownership kernel for cs=%d", cs);
    } else if (dtype == 1) {
        if (cs == 80) i2r_launch((hrt_mlp_block_k<1, 5, 2>), grid, dim3(128), 0, (hipStream_t)stream, k);
        else i2r_launch((hrt_mlp_block_k<1, 10, 4>), grid, dim3(256), 0, (hipStream_t)stream, k);
    } else {
        if (cs == 80) i2r_launch((hrt_mlp_block_k<2, 5, 2>), grid, dim3(128), 0, (hipStream_t)stream, k);
        else i2r_launch((hrt_mlp_block_k<2, 10, 4>), grid, dim3(256), 0, (hipStream_t)stream, k);
    }
    I2R_CHECK_LAUNCH("i2r_hrt_mlp_block");
    return I2R_OK;
}
