// Fused DETR-style encoder layer for variable-length token groups (inter-human relation module and the
// TransPose-H intra-human encoder), fp32 on the matrix pipe.
//
// Formulation ("token-per-lane-column"): every GEMM is computed TRANSPOSED,  Y^T[f][t] = W[f][:] . X^T[:][t],
// with the weight matrix as the MFMA A operand and the activations as the B operand.  The 16x16x4 fp32 MFMA returns
// D with col = l&15 = token and rows 4*(l>>4)+r = feature -- which is exactly the B-operand register
// image the NEXT GEMM needs (token = l&15, k-group = l>>4, 4 consecutive features per float4).  So
// q-proj -> S^T = K Q^T -> softmax -> O^T = V^T P^T -> out-proj -> +res -> LN1 -> FFN1 -> ReLU -> FFN2 -> +res -> LN2
// chains register-to-register: no LDS transposes, no intermediate HBM round trips.  Softmax / LayerNorm reductions over
// features or keys are in-lane sums + two __shfl_xor (16, 32).
//
// Operand images in memory ("fragment-packed"): an A operand fragment (16 rows x 16 k) is what 64 lanes fetch with ONE
// 16-byte load each -- lane (li = l&15, g = l>>4) takes row li, k = 4g..4g+3.  Reading that from a row-major matrix makes
// every quad of consecutive lanes touch 4 different rows, and the texture addresser then needs 64 cycles per load
// instruction instead of 16 (tools/probe/load_pattern.hip: 9.6 vs 30 TB/s aggregate L2->register) -- the one-wave-per-
// fragment kernels were bound by exactly that.  So every A operand is stored in lane order:
//     packed[((rb * KC + c) * 64 + l) * 4 + r] = M[16*rb + (l&15)][16*c + 4*(l>>4) + r]        (KC = columns / 16)
// i.e. one load instruction = 1 KB contiguous.  Weights are packed like this by the host (engine.pack_frag); the K rows
// and V^T rows of the attention are WRITTEN like this by the kernels that produce them (per 16-key fragment of a group).
#include <stdlib.h>
#include <type_traits>

#include "i2r_common.h"

namespace {

struct EncK {
    const float* src;
    const float* pos;
    float* kbuf;
    float* vbuf;
    float* out;
    const int* grp_off;
    const float* w_in; const float* b_in;
    const float* w_out; const float* b_out;
    const float* ln1_w; const float* ln1_b;
    const float* w1; const float* b1;
    const float* w2; const float* b2;
    const float* ln2_w; const float* ln2_b;
    int n_tok, n_tok_pad, n_grp, d, cs, dff_pad, pos_period, n_qblk;
    float ln_eps, qscale;
    // 16-bit MFMA mode: weights as bf16/f16 [out][in] with the columns of every 32-block permuted to the MFMA operand order
    const void* w_in_lp; const void* w_out_lp; const void* w1_lp; const void* w2_lp;
    const float* vec_lp;  // 16-bit mode: the per-feature fp32 vectors of the layer, padded to csp = 96 and concatenated (LpVec below)
    // fused K/V projection of the NEXT layer (enc_layer4_k): its in_proj (k, v rows used), destination buffers; null = none
    const float* next_w_in; const float* next_b_in; float* next_kbuf; float* next_vbuf;
    // partial key split (enc_layer4_k<.., KS = 2>): hand-off scratch, per-tile counters, tiles per XCD and how many of them stay whole
    float* split_ws; int* split_cnt; int tiles_per_xcd, full_per_xcd;
    long long* stamp;  // tuning only (env I2R_ENC_STAMP = device address): 8 s_memtime stamps per wave
};

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

__device__ __forceinline__ float xsum(float v) {  // sum over the 4 lanes sharing l&15
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}
__device__ __forceinline__ float xmax(float v) {
    v = fmaxf(v, __shfl_xor(v, 16));
    v = fmaxf(v, __shfl_xor(v, 32));
    return v;
}

// ---- which (group, 16-token tile) is tile v?  64 groups per trip: lane-parallel prefix sum of the tile counts ----
struct Tile {
    int gs, ge;  // token range of the group
    int j;       // tile index inside the group
    int base;    // global index of the group's first tile (= fragment index of its first 16 keys in kbuf / vbuf)
};
__device__ __forceinline__ void load_groups(const EncK& p, int gi, int& s0, int& e0) {
    s0 = e0 = 0;
    if (gi < p.n_grp) {
        s0 = p.grp_off[gi];
        e0 = p.grp_off[gi + 1];
    }
}
// (s0, e0) = load_groups(lane), issued by the caller ahead of other loads so the table's latency overlaps them.
// Work items are runs of QT 16-token tiles of ONE group (QT = 1 or 2); Tile.j counts work items inside the group.
template <int QT>
__device__ __forceinline__ Tile locate_tile(const EncK& p, int v, int lane, int s0, int e0) {
    Tile t = {0, 0, 0, 0};
    int b = v, base16 = 0;
    for (int base = 0; base < p.n_grp; base += 64) {
        if (base > 0) load_groups(p, base + lane, s0, e0);
        const int nq16 = (e0 - s0 + 15) >> 4;
        const int nq = QT == 1 ? nq16 : (nq16 + QT - 1) / QT;
        int inc = nq, inc16 = nq16;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int u = __shfl_up(inc, d);
            if (lane >= d) inc += u;
            if (QT != 1) {
                const int u16 = __shfl_up(inc16, d);
                if (lane >= d) inc16 += u16;
            }
        }
        if (QT == 1) inc16 = inc;
        const int total = __shfl(inc, 63);
        if (b < total) {
            const int src = __ffsll((long long)__ballot(inc > b)) - 1;
            t.gs = __shfl(s0, src);
            t.ge = __shfl(e0, src);
            b -= __shfl(inc - nq, src);
            base16 += __shfl(inc16 - nq16, src);
            break;
        }
        b -= total;
        base16 += __shfl(inc16, 63);
    }
    t.gs = __builtin_amdgcn_readfirstlane(t.gs);
    t.ge = __builtin_amdgcn_readfirstlane(t.ge);
    t.j = __builtin_amdgcn_readfirstlane(b);
    t.base = __builtin_amdgcn_readfirstlane(base16);
    return t;
}

// store one 16-token fragment of K (as the A operand of S^T = K Q^T) and of V^T (A operand of O^T = V^T P^T), fragment-packed.
// k / v: D-layout fragments nt of the projections (lane (li, g): token li, features 16nt + 4g + r).
__device__ __forceinline__ void store_kv_frag(float* kbuf, float* vbuf, int frag, int DCn, int nt, f32x4 k, f32x4 v, bool has_k, bool has_v,
                                              int lane) {
    const int li = lane & 15, g = lane >> 4;
    const size_t base = ((size_t)frag * DCn + nt) * 64;
    if (has_k) *reinterpret_cast<f32x4*>(kbuf + (base + lane) * 4) = k;  // K[key li][16nt + 4g + r]: already the operand image
    if (has_v) {
        // V^T[dim 16nt + li'][key 4g' + r']: a 4x4 transpose inside the lane quads moves (key li, dims 4g + r) to
        // (dim 4g + (li&3), keys 4(li>>2) + r')  ->  operand lane li' = 4g + (li&3), g' = li >> 2
        const f32x4 tv = quad_transpose(v, li & 3);
        *reinterpret_cast<f32x4*>(vbuf + (base + 4 * g + (li & 3) + 16 * (li >> 2)) * 4) = tv;
    }
}

// one output fragment (16 features x 16 tokens) of  Y^T = W X^T (+bias):  acc += sum_c W[16nt+li][16c+4g..] * x[c];
// W fragment-packed (see the header): the KC loads of a fragment are 1 KB contiguous each
template <int KC>
__device__ __forceinline__ void load_wrow(f32x4 (&wr)[KC], const float* W, int nt, int lane) {
    const float* f = W + ((size_t)nt * KC * 64 + lane) * 4;
#pragma unroll
    for (int c = 0; c < KC; ++c) wr[c] = ld4(f + c * 256);
}
template <int KC>
__device__ __forceinline__ f32x4 frag_mm(const f32x4 (&wr)[KC], const f32x4 (&x)[KC], f32x4 acc) {
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = mfma16(wr[c][s], x[c][s], acc);
    return acc;
}
// fragment `idx` (wave-uniform, in an SGPR) of a register array: a scalar branch tree instead of DC selects per element
template <int DC>
__device__ __forceinline__ f32x4 pick_frag(const f32x4 (&v)[DC], int idx) {
    static_assert(DC == 5 || DC == 6, "DC");
    switch (idx) {
        case 0: return v[0];
        case 1: return v[1];
        case 2: return v[2];
        case 3: return v[3];
        case 4: return v[4];
        default: return v[DC - 1];
    }
}
// ---- K / V projection of one 16-token tile per workgroup (first layer of a stack; later layers get theirs from the fused tail).
//      4 waves: wave w computes the output fragments w, w+4, ... of [K | V] (2*DC fragments), weight rows fetched up front ----
template <int DC>
__global__ __launch_bounds__(256) void enc_kv_k(const EncK p) {
    constexpr int cs = DC * 16, SK = (2 * DC + 3) / 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
    // Re-arm the hand-off counters of the partial key split for the layers that follow in this stream (a kernel boundary orders this
    // store before them): enc_layer4_k leaves them zero itself, but a launch that faulted or was aborted half-way may not -- the
    // first kernel of every encoder stack therefore makes the invariant true again instead of trusting the previous forward.
    if (blockIdx.x == 0 && p.split_cnt) p.split_cnt[tid] = 0;
    int s0, e0;
    load_groups(p, lane, s0, e0);
    f32x4 wk[SK][DC], bk[SK];
#pragma unroll
    for (int s = 0; s < SK; ++s) {
        const int f = min(wave + 4 * s, 2 * DC - 1);
        load_wrow<DC>(wk[s], p.w_in + (size_t)cs * cs, f, lane);  // K rows = row blocks [DC, 2DC), V rows = [2DC, 3DC)
        bk[s] = ld4(p.b_in + cs + 16 * f + 4 * g);
    }
    const Tile t = locate_tile<1>(p, blockIdx.x, lane, s0, e0);
    const int tok = t.gs + 16 * t.j + li;
    const bool valid = tok < t.ge;
    const int row = valid ? tok : t.ge - 1;
    const int prow = p.pos_period > 0 ? row % p.pos_period : row;
    f32x4 xs[DC], xq[DC];
#pragma unroll
    for (int c = 0; c < DC; ++c) {
        xs[c] = ld4(p.src + (size_t)row * cs + 16 * c + 4 * g);
        xq[c] = xs[c];
        if (p.pos) xq[c] += ld4(p.pos + (size_t)prow * cs + 16 * c + 4 * g);
    }
#pragma unroll
    for (int s = 0; s < SK; ++s) {
        const int f = wave + 4 * s;
        if (f < 2 * DC) {  // (wave-uniform)
            const bool isv = f >= DC;
            f32x4 a = isv ? frag_mm<DC>(wk[s], xs, bk[s]) : frag_mm<DC>(wk[s], xq, bk[s]);
            if (isv && !valid) a = (f32x4){0.f, 0.f, 0.f, 0.f};  // keys past the group: masked scores, but V must stay finite
            store_kv_frag(p.kbuf, p.vbuf, t.base + t.j, DC, isv ? f - DC : f, a, a, !isv, isv, lane);
        }
    }
}

// LayerNorm over the real d features of each token column (features live in y[nt][r] x 4 lanes).  The pad features of y are exact
// zeros (zero weight rows, biases and residual), so the sum of squares over all cs features minus their (cs - d) mean^2 is the sum over
// the real ones -- no per-element mask -- and the two divisions by d are one reciprocal.  In the fp32 kernels every vector instruction
// competes with the MFMAs for the SIMD's fp32 ALUs (PMC: 1912 VALU next to 687 MFMAs per wave), so instructions saved here are pipe time.
template <int DC>
__device__ __forceinline__ void layer_norm(f32x4 (&y)[DC], const float* w, const float* b, int d, float eps, int g) {
    const float inv_d = __builtin_amdgcn_rcpf((float)d);
    float s = 0.f;
#pragma unroll
    for (int nt = 0; nt < DC; ++nt) s += (y[nt][0] + y[nt][1]) + (y[nt][2] + y[nt][3]);
    const float mean = xsum(s) * inv_d;
    float v = 0.f;
#pragma unroll
    for (int nt = 0; nt < DC; ++nt) {
        y[nt] -= mean;
        v += (y[nt][0] * y[nt][0] + y[nt][1] * y[nt][1]) + (y[nt][2] * y[nt][2] + y[nt][3] * y[nt][3]);
    }
    const float var = (xsum(v) - (float)(DC * 16 - d) * mean * mean) * inv_d;
    const float rstd = rsqrtf(fmaxf(var, 0.f) + eps);
#pragma unroll
    for (int nt = 0; nt < DC; ++nt) {
        const f32x4 wv = ld4(w + 16 * nt + 4 * g), bv = ld4(b + 16 * nt + 4 * g);
        y[nt] = y[nt] * (rstd * wv) + bv;  // (pad features: w = b = 0 -> exact zeros again)
    }
}

// =====================================================================================================================
// fp32 encoder layer, 4 waves per 16-query tile (enc_layer4_k) -- the default fp32 kernel
// ---------------------------------------------------------------------------------------------------------------------
// The vanilla inter-human encoder has only 6144 tokens = 384 query fragments for 1024 SIMDs, so one-wave-per-fragment is
// latency-bound.  Here a workgroup of 4 waves owns one fragment: the KEYS are split 4 ways (each wave runs its own online
// softmax over every 4th 16-key fragment, partial (m, l, O) merged through LDS), and the GEMMs of the layer tail are split
// by OUTPUT fragments (wave w computes fragments w, w+4, ...), activations exchanged through LDS (a few KB).  The K / V
// projection of the NEXT layer is fused into the tail (the layer output is already in registers), removing the separate
// enc_kv launch for all layers but the first.
//
// KS = 2 ("partial key split"): launches whose tile count lies between one and two per CU (the vanilla model's 384 tiles on 256 CUs)
// leave half the chip with two workgroups per CU and half with one.  Then exactly as many tiles as needed are handled by TWO
// workgroups, each over half of the group's keys, so that every CU gets two workgroups and 1.5 tiles of work: per XCD (64 slots)
// `full_per_xcd` whole tiles come first in the dispatch order (one per CU), the split tiles' halves follow (the second workgroup of
// each CU).  A half writes its partial softmax state (m, l, O) to a scratch slot with agent-scope (sc1) stores, completes them
// (s_waitcnt vmcnt(0)) and bumps the tile's counter; the half that finds the other already there merges both states IN A FIXED ORDER
// (bit-identical results whichever arrives last) and runs the layer tail, the first one exits.  Nobody waits: deadlock-free.
constexpr int kSplitSlot = 7 * 256;  // floats per (split tile, half): O (DC <= 6 fragments, lane order) + one f32x4 {m, l, -, -} per lane
__device__ __forceinline__ void st_agent(float* q, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(q), "v"(v) : "memory"); }
__device__ __forceinline__ f32x4 ld_agent(const float* q) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(q) : "memory");
    return v;
}
// the compiler does not count the loads above: wait for them with the results as in/out operands, so no use can move ahead of the wait
template <int N>
__device__ __forceinline__ void wait_agent_loads(f32x4 (&a)[N], f32x4& b) {
    static_assert(N == 5 || N == 6, "DC");
    if constexpr (N == 6)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(b)::"memory");
    else
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(b)::"memory");
}

template <int DC, int FC, int QT, int KS = 1>
__global__ __launch_bounds__(256, 2) void enc_layer4_k(const EncK p) {
    static_assert(KS == 1 || QT == 1, "the key split serves the small one-tile-per-workgroup launches only");
    constexpr int cs = DC * 16, dff = FC * 16;
    constexpr int SD = (DC + 3) / 4, SF = (FC + 3) / 4, SK = (2 * DC + 3) / 4;  // output-fragment slots per wave of each GEMM
    constexpr int XF = FC > DC ? FC : DC;
    // LDS (float4 units): exchange area X[XF][QT][64], partial-O area O[4][QT][DC][64] (re-used as the FFN2 exchange area
    // Y[DC][QT][64] once the merge is over), (m, l) area ML[4][QT][2][16]
    __shared__ f32x4 Xs[XF * QT * 64];
    __shared__ f32x4 Os[4 * QT * DC * 64];
    __shared__ float MLs[4 * QT * 2 * 16];
    f32x4* const Ys = Os;
    // the small per-feature vectors of the layer tail, staged once (read back as 16-byte broadcasts; no VMEM latency in the tail)
    constexpr int P_BOUT = 0, P_LN1W = cs, P_LN1B = 2 * cs, P_B1 = 3 * cs, P_B2 = 3 * cs + dff, P_LN2W = 4 * cs + dff,
                  P_LN2B = 5 * cs + dff, P_BKV = 6 * cs + dff, P_END = 8 * cs + dff;
    __shared__ __attribute__((aligned(16))) float Ps[P_END];
    __shared__ int s_arrived;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (scalar: the slot tests below are scalar branches)
    const int li = lane & 15, g = lane >> 4;
#ifdef I2R_TUNING
#define STAMP(i) do { if (p.stamp && lane == 0) p.stamp[((size_t)blockIdx.x * 4 + wave) * 8 + (i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#define STAMP0(i) do { if (p.stamp && lane == 0) p.stamp[((size_t)blockIdx.x * 4 + wave) * 8 + (i)] = 0; } while (0)
#else
#define STAMP(i) do { } while (0)
#define STAMP0(i) do { } while (0)
#endif
    STAMP(0);

    // Every GEMM of the layer hands wave w the output fragments w, w+4, ... ("slots"; a slot past the end is computed on a
    // clamped row and not written).  The weight rows of a slot are fetched one phase AHEAD of their use (they depend on
    // nothing but the wave id), so their L2 latency hides under the previous phase's exchange / barrier / LayerNorm.
    int gs0, ge0;
    load_groups(p, lane, gs0, ge0);  // (first: everything below waits in issue order)
    f32x4 wq[SD][DC], bq[SD];
#pragma unroll
    for (int s = 0; s < SD; ++s) {
        // (a slot past the last fragment loads a clamped row -- no branch around the loads -- but is not COMPUTED: time follows the MFMA count)
        load_wrow<DC>(wq[s], p.w_in, min(wave + 4 * s, DC - 1), lane);
        bq[s] = ld4(p.b_in + 16 * min(wave + 4 * s, DC - 1) + 4 * g);
    }
    f32x4 pstage = (f32x4){0.f, 0.f, 0.f, 0.f};
    {   // one 16-byte piece per thread of the concatenated tail vectors (P_END / 4 <= 256 pieces)
        static_assert(P_END <= 1024, "tail vector staging: one piece per thread");
        const int i = tid * 4;
        const float* sp = p.b_out;
        int o = i;
        if (i >= P_LN1W) sp = p.ln1_w, o = i - P_LN1W;
        if (i >= P_LN1B) sp = p.ln1_b, o = i - P_LN1B;
        if (i >= P_B1) sp = p.b1, o = i - P_B1;
        if (i >= P_B2) sp = p.b2, o = i - P_B2;
        if (i >= P_LN2W) sp = p.ln2_w, o = i - P_LN2W;
        if (i >= P_LN2B) sp = p.ln2_b, o = i - P_LN2B;
        if (i >= P_BKV) sp = p.next_b_in, o = cs + i - P_BKV;
        if (i < P_END && sp) pstage = ld4(sp + o);
    }

    // ---- which (group, query tiles)?  Workgroup b runs on XCD b % 8: give every XCD a contiguous run of tiles, so the K / V
    //      of one group are read through ONE L2 instead of all eight ----
    int v, half = 0, slot = 0;
    bool split = false;
    if constexpr (KS == 1) {
        const int per_xcd = gridDim.x >> 3;
        v = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    } else {
        const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3, F = p.full_per_xcd;
#ifndef I2R_SPLIT_WHOLE_FIRST
        // the halves come FIRST in the dispatch order: the SIMDs issue oldest-first, and the half that finishes its tile is the
        // longest chain of the launch (start-up + half the keys + hand-off + tail) -- it must not queue behind a whole tile's MFMAs
        const int H = 2 * (p.tiles_per_xcd - F), qh = q;
        split = q < H;
        v = xcd * p.tiles_per_xcd + (split ? F + (q >> 1) : q - H);
#else
        const int qh = q - F;
        split = q >= F;
        v = xcd * p.tiles_per_xcd + (split ? F + (qh >> 1) : q);
#endif
        half = split ? qh & 1 : 0;
        slot = xcd * (p.tiles_per_xcd - F) + (qh >> 1);
    }
    if (v >= p.n_qblk) return;
    const Tile t = locate_tile<QT>(p, v, lane, gs0, ge0);
    if (tid * 4 < P_END) *reinterpret_cast<f32x4*>(Ps + tid * 4) = pstage;  // (visible after the barrier that follows the q projection)
    const int gs = t.gs, ge = t.ge;
    const int nfrag = (ge - gs + 15) >> 4;
    // this workgroup's share of the group's 16-key fragments: [f_lo, f_hi) (a split tile: lower / upper half)
    const int nh = (KS == 2 && split) ? (nfrag + 1) >> 1 : nfrag;
    const int f_lo = half * nh, f_hi = min(nfrag, f_lo + nh);
    int qtok[QT], qrow[QT], prow[QT];
    bool qvalid[QT];
#pragma unroll
    for (int u = 0; u < QT; ++u) {
        qtok[u] = gs + (t.j * QT + u) * 16 + li;
        qvalid[u] = qtok[u] < ge;
        qrow[u] = qvalid[u] ? qtok[u] : ge - 1;
        prow[u] = p.pos_period > 0 ? qrow[u] % p.pos_period : qrow[u];
    }

    // A operands of the group's jj-th 16-key fragment (fragment-packed: 2 x DC contiguous 1 KB loads)
    auto fetch_kv = [&](int jj, f32x4(&ka)[DC], f32x4(&va)[DC]) {
        const size_t f = ((size_t)(t.base + jj) * DC * 64 + lane) * 4;
#pragma unroll
        for (int c = 0; c < DC; ++c) {
            ka[c] = ld4(p.kbuf + f + c * 256);
            va[c] = ld4(p.vbuf + f + c * 256);
        }
    };
    f32x4 ka0[DC], va0[DC], ka1[DC], va1[DC];
    f32x4 q[QT][DC];
    {
        f32x4 xq[QT][DC];
#pragma unroll
        for (int u = 0; u < QT; ++u)
#pragma unroll
            for (int c = 0; c < DC; ++c) {
                xq[u][c] = ld4(p.src + (size_t)qrow[u] * cs + 16 * c + 4 * g);
                if (p.pos) xq[u][c] += ld4(p.pos + (size_t)prow[u] * cs + 16 * c + 4 * g);
            }
        fetch_kv(min(f_lo + wave, nfrag - 1), ka0, va0);  // first key fragment of this wave: in flight under the q projection
        __builtin_amdgcn_sched_barrier(0);
        // ---- q projection (scaled by d^-1/2 * log2 e: the softmax below works in base 2), exchanged through LDS ----
#pragma unroll
        for (int s = 0; s < SD; ++s) {
            const int nt = wave + 4 * s;
            if (nt >= DC) continue;
#pragma unroll
            for (int u = 0; u < QT; ++u) Xs[(nt * QT + u) * 64 + lane] = frag_mm<DC>(wq[s], xq[u], bq[s]) * (p.qscale * 1.4426950408889634f);
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < QT; ++u)
#pragma unroll
        for (int c = 0; c < DC; ++c) q[u][c] = Xs[(c * QT + u) * 64 + lane];
    STAMP(1);

    // ---- attention over this wave's key fragments (wave, wave+4, ...), each prefetched a whole fragment ahead.
    //      Online softmax with a LAZY reference: the running reference m (per query, base-2 units) is only raised -- and O, l
    //      rescaled -- when some score exceeds it by more than 2^10; otherwise p = 2^(s-m) <= 1024 is accumulated as is.
    //      The normalisation O / l at the end is exact either way (same reference in numerator and denominator). ----
    f32x4 o[QT][DC];
    float m_run[QT], l_run[QT];  // l_run: this lane's keys only (summed over the 4 lanes of a query at the end)
#pragma unroll
    for (int u = 0; u < QT; ++u) {
        m_run[u] = -__builtin_inff();
        l_run[u] = 0.f;
#pragma unroll
        for (int nt = 0; nt < DC; ++nt) o[u][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    auto attend = [&](int k0, const f32x4(&ka)[DC], const f32x4(&va)[DC]) {
        f32x4 st[QT];
#pragma unroll
        for (int u = 0; u < QT; ++u) st[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < DC; ++c)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int u = 0; u < QT; ++u) st[u] = mfma16(ka[c][s], q[u][c][s], st[u]);  // S^T[key 4g+r][query li]
        if (k0 + 16 > ge) {  // (wave-uniform) ragged last fragment of the group
#pragma unroll
            for (int u = 0; u < QT; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (k0 + 4 * g + r >= ge) st[u][r] = -__builtin_inff();
        }
#pragma unroll
        for (int u = 0; u < QT; ++u) {
            const float mx = fmaxf(fmaxf(st[u][0], st[u][1]), fmaxf(st[u][2], st[u][3]));
            if (__any(mx > m_run[u] + 10.f)) {  // (wave-uniform, rare after the first fragment)
                const float m_new = fmaxf(m_run[u], xmax(mx));
                const float alpha = __builtin_amdgcn_exp2f(m_run[u] - m_new);  // first time: 2^-inf = 0
                l_run[u] *= alpha;
#pragma unroll
                for (int nt = 0; nt < DC; ++nt) o[u][nt] *= alpha;
                m_run[u] = m_new;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                st[u][r] = __builtin_amdgcn_exp2f(st[u][r] - m_run[u]);
                l_run[u] += st[u][r];
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int nt = 0; nt < DC; ++nt)
#pragma unroll
                for (int u = 0; u < QT; ++u) o[u][nt] = mfma16(va[nt][s], st[u][s], o[u][nt]);  // O^T[dim][query] += V^T[dim][key] P^T
    };
    {
        int jj = f_lo + wave;
        for (; jj + 4 < f_hi; jj += 8) {  // two fragments (jj, jj+4) per trip, each fetched under the other's arithmetic
            fetch_kv(jj + 4, ka1, va1);
            __builtin_amdgcn_sched_barrier(0);
            attend(gs + 16 * jj, ka0, va0);
            fetch_kv(jj + 8 < f_hi ? jj + 8 : jj, ka0, va0);  // (past the end: harmless re-read)
            __builtin_amdgcn_sched_barrier(0);
            attend(gs + 16 * jj + 64, ka1, va1);
        }
        if (jj < f_hi) attend(gs + 16 * jj, ka0, va0);
    }
    STAMP(2);
    // out-proj rows of this wave's slots + the residual pieces they need: in flight under the merge
    f32x4 wo[SD][DC], res[SD][QT];
    auto fetch_wo = [&]() {
#pragma unroll
        for (int s = 0; s < SD; ++s) {
            load_wrow<DC>(wo[s], p.w_out, min(wave + 4 * s, DC - 1), lane);
#pragma unroll
            for (int u = 0; u < QT; ++u) res[s][u] = ld4(p.src + (size_t)qrow[u] * cs + 16 * min(wave + 4 * s, DC - 1) + 4 * g);
        }
    };
    if constexpr (QT == 1) fetch_wo();
    __builtin_amdgcn_sched_barrier(0);
    // ---- merge the four partial softmax states ----
#pragma unroll
    for (int u = 0; u < QT; ++u) {
        const float l = xsum(l_run[u]);
        if (g == 0) {
            MLs[((wave * QT + u) * 2 + 0) * 16 + li] = m_run[u];
            MLs[((wave * QT + u) * 2 + 1) * 16 + li] = l;
        }
#pragma unroll
        for (int nt = 0; nt < DC; ++nt) Os[((wave * QT + u) * DC + nt) * 64 + lane] = o[u][nt];
    }
    __syncthreads();
    f32x4 oc[QT][DC];
    float m_wg = 0.f, l_wg = 0.f;  // (KS = 2) state of the workgroup's merged partial
#pragma unroll
    for (int u = 0; u < QT; ++u) {
        float mw[4], m = -__builtin_inff();
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            mw[w] = MLs[((w * QT + u) * 2) * 16 + li];
            m = fmaxf(m, mw[w]);
        }
        float l = 0.f, sc[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            sc[w] = __builtin_amdgcn_exp2f(mw[w] - m);  // a wave without keys has m = -inf, l = 0, O = 0 -> scale 0
            if constexpr (KS == 2) sc[w] = mw[w] == -__builtin_inff() ? 0.f : sc[w];  // (a half of a short group may leave ALL waves without keys)
            l += MLs[((w * QT + u) * 2 + 1) * 16 + li] * sc[w];
        }
        float inv = __builtin_amdgcn_rcpf(l);
        if constexpr (KS == 2) {
            if (split) inv = 1.f;  // keep the partial state un-normalised for the hand-off below
            m_wg = m;
            l_wg = l;
        }
#pragma unroll
        for (int nt = 0; nt < DC; ++nt) {
            f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < 4; ++w) a += Os[((w * QT + u) * DC + nt) * 64 + lane] * sc[w];
            oc[u][nt] = a * inv;
        }
    }
    if constexpr (KS == 2) {
        if (split) {  // (workgroup-uniform) ---- hand-off between the two halves of the tile ----
            float* const mine = p.split_ws + ((size_t)slot * 2 + half) * kSplitSlot + lane * 4;
            const float* const other = p.split_ws + ((size_t)slot * 2 + (half ^ 1)) * kSplitSlot + lane * 4;
            // Publish / consume in the write-through form of MI355X_MICROARCH.md (section "inter-workgroup visibility", valid forms):
            // 16-byte `sc1` payload stores (they leave the XCD's L2 for the coherence point) -> asm `s_waitcnt vmcnt(0)` (inline asm: the
            // compiler cannot drop it) -> agent-scope atomic on the counter; the consumer reads the payload with `sc1` loads, which
            // bypass its CU's L1.  No L2 write-back / L1 invalidate fence is needed on either side (a release / acquire pair would cost
            // ~3.5 us per tile, a quarter of the layer), and the second arriver never spins: whoever sees the counter at 1 finishes.
            // One Program = one stream: launches sharing a (split_ws, split_cnt) pair must be stream-ordered (include/i2r_hip.h).
            if (wave == 0) {
#pragma unroll
                for (int nt = 0; nt < DC; ++nt) st_agent(mine + nt * 256, oc[0][nt]);
                st_agent(mine + 6 * 256, (f32x4){m_wg, l_wg, 0.f, 0.f});
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the partial has reached the coherence point before the counter moves
                if (lane == 0) s_arrived = __hip_atomic_fetch_add(p.split_cnt + slot, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();  // (also: every wave is done reading MLs / Os of the in-workgroup merge)
            if (s_arrived == 0) {  // the partner is still at work: it will find this partial and finish the tile
                STAMP(3);
                STAMP0(7);
                return;
            }
            if (tid == 0) __hip_atomic_store(p.split_cnt + slot, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
            f32x4 ao[DC], ml;
#pragma unroll
            for (int nt = 0; nt < DC; ++nt) ao[nt] = ld_agent(other + nt * 256);
            ml = ld_agent(other + 6 * 256);
            wait_agent_loads<DC>(ao, ml);
            // combine in a fixed order (half 0, then half 1) whichever workgroup finishes: the scales are computed per ROLE (mine / the
            // other's) and the products are added lower half first -- a scalar branch on `half`, no per-element selects
            const float m2 = fmaxf(m_wg, ml[0]);
            const float s_me = m_wg == -__builtin_inff() ? 0.f : __builtin_amdgcn_exp2f(m_wg - m2);
            const float s_ot = ml[0] == -__builtin_inff() ? 0.f : __builtin_amdgcn_exp2f(ml[0] - m2);
            if (half == 0) {
                const float inv2 = __builtin_amdgcn_rcpf(l_wg * s_me + ml[1] * s_ot);
#pragma unroll
                for (int nt = 0; nt < DC; ++nt) oc[0][nt] = (oc[0][nt] * s_me + ao[nt] * s_ot) * inv2;
            } else {
                const float inv2 = __builtin_amdgcn_rcpf(ml[1] * s_ot + l_wg * s_me);
#pragma unroll
                for (int nt = 0; nt < DC; ++nt) oc[0][nt] = (ao[nt] * s_ot + oc[0][nt] * s_me) * inv2;
            }
        }
    }
    STAMP(3);
    // ---- out-proj + residual (this wave's slots) -> LDS -> LayerNorm 1 on the full row in every wave ----
    if constexpr (QT != 1) {
        fetch_wo();
        __builtin_amdgcn_sched_barrier(0);
    }
    f32x4 w1r[SF][DC];
    if constexpr (QT == 1) {
#pragma unroll
        for (int s = 0; s < SF; ++s) load_wrow<DC>(w1r[s], p.w1, min(wave + 4 * s, FC - 1), lane);  // FFN1 rows, one phase ahead
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int s = 0; s < SD; ++s) {
        const int nt = wave + 4 * s;
        if (nt >= DC) continue;
#pragma unroll
        for (int u = 0; u < QT; ++u) Xs[(nt * QT + u) * 64 + lane] = frag_mm<DC>(wo[s], oc[u], ld4(Ps + P_BOUT + 16 * nt + 4 * g)) + res[s][u];
    }
    if constexpr (QT != 1) {
#pragma unroll
        for (int s = 0; s < SF; ++s) load_wrow<DC>(w1r[s], p.w1, min(wave + 4 * s, FC - 1), lane);  // (two tiles in flight: after the out-proj has released its rows)
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    f32x4 x1[QT][DC];
#pragma unroll
    for (int u = 0; u < QT; ++u) {
#pragma unroll
        for (int c = 0; c < DC; ++c) x1[u][c] = Xs[(c * QT + u) * 64 + lane];
        layer_norm<DC>(x1[u], Ps + P_LN1W, Ps + P_LN1B, p.d, p.ln_eps, g);
    }
    STAMP(4);
    // ---- FFN ----
    f32x4 w2r[SD][FC];
    if constexpr (QT == 1) {  // FFN2 rows one phase ahead (with two tiles in flight the registers are not there: fetched after FFN1)
#pragma unroll
        for (int s = 0; s < SD; ++s) load_wrow<FC>(w2r[s], p.w2, min(wave + 4 * s, DC - 1), lane);
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();  // everyone has read Xs before FFN1 overwrites it
#pragma unroll
    for (int s = 0; s < SF; ++s) {
        const int ft = wave + 4 * s;
        if (ft >= FC) continue;
#pragma unroll
        for (int u = 0; u < QT; ++u) {
            f32x4 a = frag_mm<DC>(w1r[s], x1[u], ld4(Ps + P_B1 + 16 * ft + 4 * g));
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] = fmaxf(a[r], 0.f);
            Xs[(ft * QT + u) * 64 + lane] = a;
        }
    }
    if constexpr (QT != 1) {
#pragma unroll
        for (int s = 0; s < SD; ++s) load_wrow<FC>(w2r[s], p.w2, min(wave + 4 * s, DC - 1), lane);
    }
    f32x4 x1n[SD][QT];  // the residual of FFN2 is only needed for this wave's slots (wave is scalar: a branch per slot, no selects)
#pragma unroll
    for (int s = 0; s < SD; ++s)
#pragma unroll
        for (int u = 0; u < QT; ++u) x1n[s][u] = pick_frag<DC>(x1[u], wave + 4 * s);
    __syncthreads();
    STAMP(5);
#pragma unroll
    for (int u = 0; u < QT; ++u) {  // (one tile at a time: h is FC fragments)
        f32x4 h[FC];
#pragma unroll
        for (int c = 0; c < FC; ++c) h[c] = Xs[(c * QT + u) * 64 + lane];
#pragma unroll
        for (int s = 0; s < SD; ++s) {
            const int nt = wave + 4 * s;
            if (nt >= DC) continue;
            Ys[(nt * QT + u) * 64 + lane] = frag_mm<FC>(w2r[s], h, ld4(Ps + P_B2 + 16 * nt + 4 * g)) + x1n[s][u];
        }
    }
    // K / V rows of the next layer's in_proj for this wave's slots (fragments 0..DC-1: K rows, DC..2DC-1: V rows)
    f32x4 wk[SK][DC], yq[QT][DC];
    if (p.next_w_in) {
#pragma unroll
        for (int s = 0; s < SK; ++s) load_wrow<DC>(wk[s], p.next_w_in + (size_t)cs * cs, min(wave + 4 * s, 2 * DC - 1), lane);
        if (p.pos) {
#pragma unroll
            for (int u = 0; u < QT; ++u)
#pragma unroll
                for (int c = 0; c < DC; ++c) yq[u][c] = ld4(p.pos + (size_t)prow[u] * cs + 16 * c + 4 * g);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    f32x4 y[QT][DC];
#pragma unroll
    for (int u = 0; u < QT; ++u) {
#pragma unroll
        for (int c = 0; c < DC; ++c) y[u][c] = Ys[(c * QT + u) * 64 + lane];
        layer_norm<DC>(y[u], Ps + P_LN2W, Ps + P_LN2B, p.d, p.ln_eps, g);
    }
    STAMP(6);
#pragma unroll
    for (int u = 0; u < QT; ++u)
        if (qvalid[u]) {
#pragma unroll
            for (int s = 0; s < SD; ++s) {
                const int nt = wave + 4 * s;
                if (nt < DC) *reinterpret_cast<f32x4*>(p.out + (size_t)qtok[u] * cs + 16 * nt + 4 * g) = pick_frag<DC>(y[u], nt);
            }
        }
    // ---- K / V of the next layer from the layer output still in registers ----
    if (p.next_w_in) {
#pragma unroll
        for (int u = 0; u < QT; ++u)
#pragma unroll
            for (int c = 0; c < DC; ++c) yq[u][c] = p.pos ? yq[u][c] + y[u][c] : y[u][c];
#pragma unroll
        for (int s = 0; s < SK; ++s) {
            const int f = wave + 4 * s;
            if (f < 2 * DC) {  // (wave-uniform)
                const bool isv = f >= DC;
                const int nt = isv ? f - DC : f;
                const f32x4 bias = ld4(Ps + P_BKV + 16 * f + 4 * g);
#pragma unroll
                for (int u = 0; u < QT; ++u) {
                    if (t.j * QT + u >= nfrag) continue;  // (wave-uniform) the odd last tile of a group has no second half
                    f32x4 a = isv ? frag_mm<DC>(wk[s], y[u], bias) : frag_mm<DC>(wk[s], yq[u], bias);
                    if (isv && !qvalid[u]) a = (f32x4){0.f, 0.f, 0.f, 0.f};  // keys past the group: V must stay finite
                    store_kv_frag(p.next_kbuf, p.next_vbuf, t.base + t.j * QT + u, DC, nt, a, a, !isv, isv, lane);
                }
            }
        }
    }
    STAMP(7);
#undef STAMP
}

// =====================================================================================================================
// 16-bit MFMA variant (BASELINE config 3: bf16 compute, fp32 accumulate; f16 for config 5)
// ---------------------------------------------------------------------------------------------------------------------
// v_mfma_f32_16x16x32_{bf16,f16} contracts 32 k-values: lane (l&15, g = l>>4) supplies 8 of them.  The fp32 D layout of the
// previous GEMM gives a lane the features {16nt + 4g + r}; two neighbouring fragments (nt = 2c, 2c+1) packed together are
// 8 features of the 32-block c -- a PERMUTATION of the block (new position 8g + 4*half + r  <-  feature 32c + 16*half + 4g + r)
// that is applied consistently to both MFMA operands: the host permutes the weight columns the same way
// (engine.Packer.encoder_layer_lp), enc_kv_lp_k writes K rows and V^T key-blocks in that order.  So the register-to-register
// chaining of the fp32 kernel carries over with a pack8() between GEMMs.
__device__ __forceinline__ f32x4 ld16(const void* base, size_t elem_off) {  // 16 bytes = 8 16-bit elements at element offset
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const unsigned short*>(base) + elem_off);
}

// K (permuted rows, 16-bit) and V^T (permuted 32-key blocks, 16-bit), projected on the 16-bit pipe like every other GEMM of the layer
// Y^T[NF frags] = W16[NF*16, KC*32] X^T + b for QF token fragments at once (the weight rows are fetched once per fragment row)
template <int NF, int KC, int QF, int DT, typename Epi>
__device__ __forceinline__ void gemm_T_lp(const void* W, const float* bias, int ld, const f32x4 (&x)[QF][KC], int li, int g, Epi epi) {
    f32x4 w0[KC], w1[KC];
    auto loadw = [&](f32x4(&w)[KC], int nt) {
#pragma unroll
        for (int c = 0; c < KC; ++c) w[c] = ld16(W, (((size_t)nt * KC + c) * 64 + (li + 16 * g)) * 8);  // fragment-packed, 1 KB per load
    };
    auto mm = [&](const f32x4(&w)[KC], int nt) {
        const f32x4 b = ld4(bias + 16 * nt + 4 * g);
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) {
            f32x4 acc = b;
#pragma unroll
            for (int c = 0; c < KC; ++c) acc = mfma32_lp<DT>(w[c], x[qf][c], acc);
            epi(nt, qf, acc);
        }
    };
    loadw(w0, 0);
#pragma unroll
    for (int nt = 0; nt < NF; nt += 2) {
        if (nt + 1 < NF) loadw(w1, nt + 1);
        mm(w0, nt);
        if (nt + 1 < NF) {
            if (nt + 2 < NF) loadw(w0, nt + 2);
            mm(w1, nt + 1);
        }
    }
}

// Offsets (floats) into EncK::vec_lp: every per-feature vector of the layer padded to csp = DC * 16 (= 96 for d = 96 AND d = 78:
// the 16-bit MFMA contracts 32 features at a time, so a 78-wide model runs as 3 blocks with zero weights beyond feature 77)
template <int DC, int FC>
struct LpVec {
    static constexpr int csp = DC * 16, dff = FC * 16;
    static constexpr int BIN = 0, BOUT = 3 * csp, LN1W = 4 * csp, LN1B = 5 * csp, B1 = 6 * csp, B2 = 6 * csp + dff, LN2W = 7 * csp + dff,
                         LN2B = 8 * csp + dff, END = 9 * csp + dff;
};

// one 32-feature block c of a token row as the packed B operand: features 32c + 4g.. and 32c + 16 + 4g..; CSR = 16-feature blocks that
// exist in the HBM row (cs / 16: 6, or 5 for the 80-float rows of d = 78 -- block 5 is read as zeros)
template <int CSR, int DT>
__device__ __forceinline__ f32x4 load_xblk(const float* row, const float* prow, int c, int g) {
    f32x4 a0 = ld4(row + 32 * c + 4 * g), a1 = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (2 * c + 1 < CSR) a1 = ld4(row + 32 * c + 16 + 4 * g);
    if (prow) {
        a0 += ld4(prow + 32 * c + 4 * g);
        if (2 * c + 1 < CSR) a1 += ld4(prow + 32 * c + 16 + 4 * g);
    }
    return pack8<DT>(a0, a1);
}

// K / V projection, TB consecutive 32-token blocks of ONE group per workgroup (blocks are numbered group by group, like the 16-key
// fragments of the fp32 kernels: block b of group g sits at index sum_{g' < g} ceil(len_g' / 32) + b, so group offsets need no
// alignment).  TB = 2 (round 4): the 36 KB of K / V projection rows stream from L2 once per 64 tokens instead of once per 32 -- every
// wave of the launch re-reads them, which was more L2 traffic than the launch's HBM traffic -- and K is projected and stored before V
// so the two sets of accumulators never live together.
#ifndef I2R_KV_TB
#define I2R_KV_TB 2
#endif
template <int DC, int CSR, int DT, int TB>
__global__ __launch_bounds__(64) void enc_kv_lp_k(const EncK p) {
    constexpr int KC = DC / 2, cs = CSR * 16, csp = DC * 16, TF = 2 * TB;
    const int lane = threadIdx.x, li = lane & 15, g = lane >> 4;
    int b = blockIdx.x, gs = 0, ge = 0, base32 = 0;
    for (int grp = 0; grp < p.n_grp; ++grp) {
        gs = p.grp_off[grp];
        ge = p.grp_off[grp + 1];
        const int nb = (ge - gs + 31) >> 5, nw = (nb + TB - 1) / TB;
        if (b < nw) break;
        b -= nw;
        base32 += nb;
    }
    const int t0 = gs + b * 32 * TB;
    // B operands of the 16-token fragments: src + pos (keys) and src (values), packed to 16 bit like every GEMM input here
    f32x4 xq[TF][KC], xs[TF][KC];
    bool valid[TF];
#pragma unroll
    for (int tf = 0; tf < TF; ++tf) {
        const int tok = t0 + tf * 16 + li;
        valid[tf] = tok < ge;
        const int row = valid[tf] ? tok : ge - 1;  // (rows past the group: finite duplicates, masked as keys, zeroed as values)
        const int prow = p.pos_period > 0 ? row % p.pos_period : row;
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            xs[tf][c] = load_xblk<CSR, DT>(p.src + (size_t)row * cs, nullptr, c, g);
            xq[tf][c] = p.pos ? load_xblk<CSR, DT>(p.src + (size_t)row * cs, p.pos + (size_t)prow * cs, c, g) : xs[tf][c];
        }
    }
    // K rows of the in_proj: row blocks [DC, 2DC) of the fragment-packed 16-bit matrix, V rows: [2DC, 3DC)
    const unsigned short* w16 = reinterpret_cast<const unsigned short*>(p.w_in_lp);
    const float* bin = p.vec_lp;  // LpVec::BIN = 0
    // fragment-packed 16-bit operand images of a 32-token block, one 16-byte store per lane:
    //   K:   [(blk*2 + tf)*KC + c][lane][8]      lane (li, g): key 16tf + li, the 8 permuted features g*8.. of 32-block c
    //   V^T: [blk*DC + nt][lane'][8]              lane' (li' = feature in fragment nt, g' = key quad): 8 keys in the block's
    //        permuted order (position 8*((k%16)/4) + 4*(k/16) + k%4); a 4x4 quad transpose turns (key li, features 4g+r) into
    //        (feature 4g + (li&3), keys 4(li>>2) + r') so both 16-key halves pack into the destination lane's 16 bytes
    unsigned short* k16 = reinterpret_cast<unsigned short*>(p.kbuf);
    unsigned short* v16 = reinterpret_cast<unsigned short*>(p.vbuf);
    const size_t blk0 = (size_t)base32 + (size_t)b * TB;
    bool bok[TB];  // (wave-uniform: the group may end inside the workgroup's run of blocks -- a block that does not exist is not stored,
                   //  its index belongs to the next group)
#pragma unroll
    for (int h = 0; h < TB; ++h) bok[h] = t0 + 32 * h < ge;
    {
        f32x4 ak[TF][DC];
        gemm_T_lp<DC, KC, TF, DT>(w16 + (size_t)DC * KC * 512, bin + csp, csp, xq, li, g, [&](int nt, int tf, f32x4 a) { ak[tf][nt] = a; });
#pragma unroll
        for (int h = 0; h < TB; ++h) {
            if (!bok[h]) continue;
#pragma unroll
            for (int c = 0; c < KC; ++c)
#pragma unroll
                for (int tf = 0; tf < 2; ++tf)
                    *reinterpret_cast<f32x4*>(k16 + (((((blk0 + h) * 2 + tf) * KC + c) * 64 + lane) * 8)) = pack8<DT>(ak[2 * h + tf][2 * c], ak[2 * h + tf][2 * c + 1]);
        }
    }
    {
        f32x4 av[TF][DC];
        gemm_T_lp<DC, KC, TF, DT>(w16 + (size_t)2 * DC * KC * 512, bin + 2 * csp, csp, xs, li, g, [&](int nt, int tf, f32x4 a) {
            av[tf][nt] = valid[tf] ? a : (f32x4){0.f, 0.f, 0.f, 0.f};
        });
#pragma unroll
        for (int h = 0; h < TB; ++h) {
            if (!bok[h]) continue;
#pragma unroll
            for (int nt = 0; nt < DC; ++nt) {
                const f32x4 pv = pack8<DT>(quad_transpose(av[2 * h][nt], li & 3), quad_transpose(av[2 * h + 1][nt], li & 3));
                *reinterpret_cast<f32x4*>(v16 + ((((blk0 + h) * DC + nt) * 64 + 4 * g + (li & 3) + 16 * (li >> 2)) * 8)) = pv;
            }
        }
    }
}

template <int DC, int FC, int QF, int DT, int CSR>
__global__ __launch_bounds__(64) void enc_layer_lp_k(const EncK p) {
    constexpr int cs = CSR * 16, csp = DC * 16, dff = FC * 16, KC = DC / 2, FKC = FC / 2;
    typedef LpVec<DC, FC> V;
    const int lane = threadIdx.x, li = lane & 15, g = lane >> 4;
    constexpr int QT = QF * 16;
    // XCD x (= workgroup index % 8, MI355X_MICROARCH.md) takes a CONTIGUOUS eighth of the work items: the waves of a group then stream its
    // K / V through ONE 4 MB L2.  In launch order every XCD touched every group -- 57 crops x 1.2 MB of K / V per layer, all L2 misses.
    int b = xcd_band_item(blockIdx.x, p.n_qblk), gs = 0, ge = 0, base32 = 0;
    if (b < 0) return;
    for (int grp = 0; grp < p.n_grp; ++grp) {
        gs = p.grp_off[grp];
        ge = p.grp_off[grp + 1];
        const int nq = (ge - gs + QT - 1) / QT;
        if (b < nq) break;
        b -= nq;
        base32 += (ge - gs + 31) >> 5;
    }
    const int q0 = gs + b * QT;
    int qrow[QF];
#pragma unroll
    for (int qf = 0; qf < QF; ++qf) qrow[qf] = min(q0 + qf * 16 + li, ge - 1);
    const float* vec = p.vec_lp;

    // ---- q projection: B operand = packed (src + pos) ----
    f32x4 qB[QF][KC];
    {
        f32x4 xB[QF][KC];
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) {
            const int prow = p.pos_period > 0 ? qrow[qf] % p.pos_period : qrow[qf];
#pragma unroll
            for (int c = 0; c < KC; ++c)
                xB[qf][c] = load_xblk<CSR, DT>(p.src + (size_t)qrow[qf] * cs, p.pos ? p.pos + (size_t)prow * cs : nullptr, c, g);
        }
        f32x4 q32[QF][DC];
        gemm_T_lp<DC, KC, QF, DT>(p.w_in_lp, vec + V::BIN, csp, xB, li, g, [&](int nt, int qf, f32x4 a) { q32[qf][nt] = a * (p.qscale * 1.4426950408889634f); });
#pragma unroll
        for (int qf = 0; qf < QF; ++qf)
#pragma unroll
            for (int c = 0; c < KC; ++c) qB[qf][c] = pack8<DT>(q32[qf][2 * c], q32[qf][2 * c + 1]);
    }

    // ---- flash attention over the group's 32-key blocks; K rows / V^T blocks stream from L2 into registers, one block ahead ----
    // The loop is bound by the vector ALU, not by the matrix pipe (the 16-bit MFMA is 16x the fp32 rate; round 2: ~50 VALU instructions
    // per 12 MFMAs), so the softmax is arranged to need as few VALU instructions per score as possible:
    //  * the running reference m is folded into the S accumulator (acc starts at -m instead of 0): no subtraction per score;
    //  * the row sums l come out of the matrix pipe: one extra A fragment whose row 0 is all ones makes  ol = 1^T P  next to O = V^T P;
    f32x4 o[QF][DC], ol[QF], negm[QF];
#pragma unroll
    for (int qf = 0; qf < QF; ++qf) {
        negm[qf] = ol[qf] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < DC; ++nt) o[qf][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const f32x4 one4 = (f32x4){1.f, 1.f, 1.f, 1.f};
    const f32x4 onesA = li == 0 ? pack8<DT>(one4, one4) : (f32x4){0.f, 0.f, 0.f, 0.f};  // A fragment: row 0 = ones over the block's 32 keys
    // (plain fmaxf: an inline-asm v_max3 on MFMA results gets no hazard wait states from the compiler -- NaNs under -amdgpu-mfma-vgpr-form)
    auto max3 = [](float x, float y, float z) { return fmaxf(fmaxf(x, y), z); };
    auto fetch_kv = [&](int k0, f32x4(&ka)[2][KC], f32x4(&va)[DC]) {  // fragment-packed images of the group's block (k0 - gs) / 32
        const size_t blk = (size_t)(base32 + ((k0 - gs) >> 5));
#pragma unroll
        for (int kf = 0; kf < 2; ++kf)
#pragma unroll
            for (int c = 0; c < KC; ++c) ka[kf][c] = ld16(p.kbuf, (((blk * 2 + kf) * KC + c) * 64 + lane) * 8);
#pragma unroll
        for (int nt = 0; nt < DC; ++nt) va[nt] = ld16(p.vbuf, ((blk * DC + nt) * 64 + lane) * 8);
    };
    // online softmax in base 2 with a lazy reference: m (kept as -m in every lane of a query's column) is raised -- and O, l rescaled --
    // only when a score exceeds it by more than 2^10; the first block always sets it (scores far below 0 must not underflow).  The
    // common path has no cross-lane traffic and no O rescale.
    auto attend = [&](int k0, const f32x4(&ka)[2][KC], const f32x4(&va)[DC]) {
        const bool ragged = k0 + 32 > ge;  // (wave-uniform)
        const bool first = k0 == gs;
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) {
            f32x4 st[2];
#pragma unroll
            for (int kf = 0; kf < 2; ++kf) {
                f32x4 a = negm[qf];
#pragma unroll
                for (int c = 0; c < KC; ++c) a = mfma32_lp<DT>(ka[kf][c], qB[qf][c], a);  // S^T[key 16kf+4g+r][query li] - m
                if (ragged) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (k0 + 16 * kf + 4 * g + r >= ge) a[r] = -__builtin_inff();
                }
                st[kf] = a;
            }
            const float mx = max3(max3(max3(st[0][0], st[0][1], st[0][2]), st[0][3], st[1][0]), max3(st[1][1], st[1][2], st[1][3]), st[1][1]);
            if (first || __any(mx > 10.f)) {
                float d = xmax(mx);             // how far the block's maximum lies above the reference
                if (!first) d = fmaxf(d, 0.f);  // (the reference only rises afterwards)
                if (!first) {
                    const float alpha = __builtin_amdgcn_exp2f(-d);
                    ol[qf] *= alpha;
#pragma unroll
                    for (int nt = 0; nt < DC; ++nt) o[qf][nt] *= alpha;
                }
                negm[qf] -= d;
                st[0] -= d;
                st[1] -= d;
            }
#pragma unroll
            for (int kf = 0; kf < 2; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) st[kf][r] = __builtin_amdgcn_exp2f(st[kf][r]);
            const f32x4 pB = pack8<DT>(st[0], st[1]);  // keys {4g+r} U {16+4g+r} of the block: the order V^T blocks are stored in
#pragma unroll
            for (int nt = 0; nt < DC; ++nt) o[qf][nt] = mfma32_lp<DT>(va[nt], pB, o[qf][nt]);
            ol[qf] = mfma32_lp<DT>(onesA, pB, ol[qf]);  // row 0: l[query li] += sum of the block's (16-bit rounded) weights
        }
    };
    // The same for a block that is neither a group's first nor ragged -- every block of the main loop -- as ONE basic block: the scores of
    // all QF fragments first, ONE (rarely taken) branch for the reference update of all of them, then exp / pack / O = V^T P.  With a
    // branch per fragment (attend above) the scheduler cannot place a fragment's softmax beside the next fragment's MFMAs, and the wave is
    // alone on its SIMD: whatever it does not overlap itself is idle matrix-pipe time (same box: 1.57 -> 1.45 ms per 6-layer stack).
    auto attend_fast = [&](const f32x4(&ka)[2][KC], const f32x4(&va)[DC]) {
        f32x4 st[QF][2];
        float mx[QF];
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) {
#pragma unroll
            for (int kf = 0; kf < 2; ++kf) {
                f32x4 a = negm[qf];
#pragma unroll
                for (int c = 0; c < KC; ++c) a = mfma32_lp<DT>(ka[kf][c], qB[qf][c], a);
                st[qf][kf] = a;
            }
        }
        float mxa = -__builtin_inff();
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) {
            mx[qf] = max3(max3(max3(st[qf][0][0], st[qf][0][1], st[qf][0][2]), st[qf][0][3], st[qf][1][0]), max3(st[qf][1][1], st[qf][1][2], st[qf][1][3]), st[qf][1][1]);
            mxa = fmaxf(mxa, mx[qf]);
        }
        if (__any(mxa > 10.f)) {
#pragma unroll
            for (int qf = 0; qf < QF; ++qf) {
                if (!__any(mx[qf] > 10.f)) continue;  // (exactly the fragments attend would have touched)
                const float d = fmaxf(xmax(mx[qf]), 0.f);
                const float alpha = __builtin_amdgcn_exp2f(-d);
                ol[qf] *= alpha;
#pragma unroll
                for (int nt = 0; nt < DC; ++nt) o[qf][nt] *= alpha;
                negm[qf] -= d;
                st[qf][0] -= d;
                st[qf][1] -= d;
            }
        }
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) {
#pragma unroll
            for (int kf = 0; kf < 2; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) st[qf][kf][r] = __builtin_amdgcn_exp2f(st[qf][kf][r]);
            const f32x4 pB = pack8<DT>(st[qf][0], st[qf][1]);
#pragma unroll
            for (int nt = 0; nt < DC; ++nt) o[qf][nt] = mfma32_lp<DT>(va[nt], pB, o[qf][nt]);
            ol[qf] = mfma32_lp<DT>(onesA, pB, ol[qf]);
        }
    };
    {
        f32x4 ka0[2][KC], va0[DC], ka1[2][KC], va1[DC];
        fetch_kv(gs, ka0, va0);
        const int last = gs + (((ge - 1 - gs) >> 5) << 5);  // first key of the group's last block
        int k0 = gs;
        if (k0 + 64 <= ge) {  // the group's first block sets the reference: general path
            fetch_kv(k0 + 32, ka1, va1);
            attend(k0, ka0, va0);
            fetch_kv(min(k0 + 64, last), ka0, va0);
            attend_fast(ka1, va1);
            k0 += 64;
        }
        for (; k0 + 64 <= ge; k0 += 64) {
            fetch_kv(k0 + 32, ka1, va1);
            attend_fast(ka0, va0);
            fetch_kv(min(k0 + 64, last), ka0, va0);  // (look-ahead past the end re-reads the last block, unused)
            attend_fast(ka1, va1);
        }
        if (k0 < ge) {
            if (k0 + 32 < ge) fetch_kv(k0 + 32, ka1, va1);
            attend(k0, ka0, va0);
            if (k0 + 32 < ge) attend(k0 + 32, ka1, va1);
        }
    }

    // ---- out-proj + residual + LN1, FFN, + residual + LN2 (per query fragment; weights shared across fragments) ----
    f32x4 oB[QF][KC];
#pragma unroll
    for (int qf = 0; qf < QF; ++qf) {
        const float inv = 1.f / xsum(ol[qf][0]);  // l sits in row 0 of the ones tile (lane g = 0, register 0); rows 4g of g > 0 are zero
#pragma unroll
        for (int c = 0; c < KC; ++c) oB[qf][c] = pack8<DT>(o[qf][2 * c] * inv, o[qf][2 * c + 1] * inv);
    }
    f32x4 x1[QF][DC];
    gemm_T_lp<DC, KC, QF, DT>(p.w_out_lp, vec + V::BOUT, csp, oB, li, g, [&](int nt, int qf, f32x4 a) {
        x1[qf][nt] = nt < CSR ? ld4(p.src + (size_t)qrow[qf] * cs + 16 * nt + 4 * g) + a : a;  // (features >= cs: zero weights -> a = 0)
    });
    f32x4 x1B[QF][KC];
#pragma unroll
    for (int qf = 0; qf < QF; ++qf) {
        layer_norm<DC>(x1[qf], vec + V::LN1W, vec + V::LN1B, p.d, p.ln_eps, g);
#pragma unroll
        for (int c = 0; c < KC; ++c) x1B[qf][c] = pack8<DT>(x1[qf][2 * c], x1[qf][2 * c + 1]);
    }
    f32x4 hB[QF][FKC];
    {
        f32x4 hprev[QF];
        gemm_T_lp<FC, KC, QF, DT>(p.w1_lp, vec + V::B1, csp, x1B, li, g, [&](int ft, int qf, f32x4 a) {
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] = fmaxf(a[r], 0.f);
            if (ft & 1) hB[qf][ft >> 1] = pack8<DT>(hprev[qf], a); else hprev[qf] = a;
        });
    }
    gemm_T_lp<DC, FKC, QF, DT>(p.w2_lp, vec + V::B2, dff, hB, li, g, [&](int nt, int qf, f32x4 a) { x1[qf][nt] += a; });
#pragma unroll
    for (int qf = 0; qf < QF; ++qf) {
        layer_norm<DC>(x1[qf], vec + V::LN2W, vec + V::LN2B, p.d, p.ln_eps, g);
        const int qtok = q0 + qf * 16 + li;
        if (qtok < ge) {
#pragma unroll
            for (int nt = 0; nt < CSR; ++nt) *reinterpret_cast<f32x4*>(p.out + (size_t)qtok * cs + 16 * nt + 4 * g) = x1[qf][nt];
        }
    }
}

// =====================================================================================================================
// enc_layer_lp4_k (round 5): the same 16-bit layer with the K / V stream SHARED by the four waves of a workgroup.
// ---------------------------------------------------------------------------------------------------------------------
// enc_layer_lp_k above is one wave per 64 queries that streams every 32-key block (6 K + 6 V^T fragments = 12 KB) from L2 into its own
// registers: 12 KB through the CU's 64 B/clk vector-memory path per 52 matrix instructions -- with a wave on each of the four SIMDs that
// is 59 B/clk, the whole path -- and the double-buffered block (96 registers) is what pins the kernel at one wave per SIMD, where the
// softmax of a block and its matrix instructions serialise (0.29 of the 16-bit peak for two rounds).  Here
//   * a workgroup = 4 waves x QF = 3 query fragments = 192 consecutive queries of ONE group; every wave fetches 3 of the 12 fragments
//     of a block (one block ahead, through registers) and puts them into a two-slot LDS ring; one barrier per block; all four waves
//     read their A operands from LDS -- a fragment crosses the vector-memory path once per 156 matrix instructions;
//   * without the register-resident K / V blocks a wave needs < 256 registers: TWO workgroups per CU = two waves per SIMD, one
//     wave's softmax beside the other's matrix instructions;
//   * everything else (q projection, lazy-reference online softmax in base 2, row sums from the matrix pipe, out-proj / LN / FFN tail)
//     is the code of enc_layer_lp_k, per wave.
template <int DC, int FC, int DT, int CSR>
__global__ __launch_bounds__(256, 2) void enc_layer_lp4_k(const EncK p) {
    constexpr int QF = 3, cs = CSR * 16, csp = DC * 16, dff = FC * 16, KC = DC / 2, FKC = FC / 2, NFR = 2 * KC + DC;  // fragments of a 32-key block
    static_assert(NFR == 12, "three fragments per wave");
    typedef LpVec<DC, FC> V;
    constexpr int NS = 4, PD = 3;  // LDS ring slots (12 KB each) / prefetch distance in blocks
    static_assert(PD <= NS - 1 && PD >= 2, "ring");
    __shared__ __attribute__((aligned(16))) f32x4 kvs[NS][NFR * 64];
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int QW = QF * 16, QT = 4 * QW;  // queries per wave / per workgroup
    int b = xcd_band_item(blockIdx.x, p.n_qblk), gs = 0, ge = 0, base32 = 0;  // (XCD-banded like enc_layer_lp_k; workgroup-uniform)
    if (b < 0) return;
    for (int grp = 0; grp < p.n_grp; ++grp) {
        gs = p.grp_off[grp];
        ge = p.grp_off[grp + 1];
        const int nq = (ge - gs + QT - 1) / QT;
        if (b < nq) break;
        b -= nq;
        base32 += (ge - gs + 31) >> 5;
    }
    const int q0 = gs + b * QT + wave * QW;
    int qrow[QF];
#pragma unroll
    for (int qf = 0; qf < QF; ++qf) qrow[qf] = min(q0 + qf * 16 + li, ge - 1);
    const float* vec = p.vec_lp;

    // ---- this wave's share of a block: fragments 3 wave .. 3 wave + 2 of [K (kf, c) | V^T nt] go from global memory STRAIGHT into an LDS
    //      slot (global_load_lds_dwordx4: 16 bytes per lane to M0 + 16 lane, i.e. the 1 KB fragment image as it is; no staging
    //      registers).  Nothing but the issuing wave's vmcnt orders a later ds_read behind such a load: `landed()` + the barrier. ----
    auto gload = [&](int blk, int slot) {
        const size_t bb = (size_t)base32 + blk;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int f = wave * 3 + i;  // (wave-uniform: waves 0, 1 fetch K, waves 2, 3 V^T)
            const unsigned short* src = f < 2 * KC ? reinterpret_cast<const unsigned short*>(p.kbuf) + ((bb * 2 * KC + f) * 64 + lane) * 8
                                                   : reinterpret_cast<const unsigned short*>(p.vbuf) + ((bb * DC + (f - 2 * KC)) * 64 + lane) * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)&kvs[slot][f * 64], 16, 0, 0);
        }
    };
    auto landed = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
    const int nblk = (ge - gs + 31) >> 5;
#pragma unroll
    for (int jb = 0; jb < PD; ++jb)
        if (jb < nblk) gload(jb, jb);

    // ---- q projection: B operand = packed (src + pos) ----
    f32x4 qB[QF][KC];
    {
        f32x4 xB[QF][KC];
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) {
            const int prow = p.pos_period > 0 ? qrow[qf] % p.pos_period : qrow[qf];
#pragma unroll
            for (int c = 0; c < KC; ++c)
                xB[qf][c] = load_xblk<CSR, DT>(p.src + (size_t)qrow[qf] * cs, p.pos ? p.pos + (size_t)prow * cs : nullptr, c, g);
        }
        f32x4 q32[QF][DC];
        gemm_T_lp<DC, KC, QF, DT>(p.w_in_lp, vec + V::BIN, csp, xB, li, g, [&](int nt, int qf, f32x4 a) { q32[qf][nt] = a * (p.qscale * 1.4426950408889634f); });
#pragma unroll
        for (int qf = 0; qf < QF; ++qf)
#pragma unroll
            for (int c = 0; c < KC; ++c) qB[qf][c] = pack8<DT>(q32[qf][2 * c], q32[qf][2 * c + 1]);
    }
    landed();
    __syncthreads();

    f32x4 o[QF][DC], ol[QF], negm[QF];  // negm: -reference of the lane's query column in all four elements (the S accumulators start from it)
#pragma unroll
    for (int qf = 0; qf < QF; ++qf) {
        negm[qf] = ol[qf] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < DC; ++nt) o[qf][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const f32x4 one4 = (f32x4){1.f, 1.f, 1.f, 1.f};
    const f32x4 onesA = li == 0 ? pack8<DT>(one4, one4) : (f32x4){0.f, 0.f, 0.f, 0.f};  // A fragment: row 0 = ones over the block's 32 keys
    auto max3 = [](float x, float y, float z) { return fmaxf(fmaxf(x, y), z); };
    // one 32-key block out of LDS slot `sl`; general = the group's first block (sets the reference) or its ragged last one
    auto attend = [&](const int sl, const int k0, const bool general) {
        const f32x4* ks = kvs[sl];
        f32x4 st[QF][2];
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) st[qf][0] = st[qf][1] = negm[qf];
        // The twelve A fragments of the block come out of LDS through a short register pipeline: fragment i + LA is read before the
        // matrix instructions of fragment i are issued (fences keep that order).  Read -> wait -> use per fragment left ~1700 of a
        // block's 3300 wave cycles parked on lgkmcnt (PMC); reading all twelve up front does not fit two waves per SIMD.
        constexpr int LA = 2;
        f32x4 fr[LA + 1];
        auto frag_at = [&](int i) { return ks[i * 64 + lane]; };  // fragment order in a slot: K (c-major below: kf * KC + c), then V^T nt
        auto kidx = [&](int i) { return (i & 1) * KC + (i >> 1); };  // S phase visits (c, kf) = (i >> 1, i & 1)
#pragma unroll
        for (int i = 0; i < LA; ++i) fr[i] = frag_at(kidx(i));
#pragma unroll
        for (int i = 0; i < 2 * KC; ++i) {
            const int c = i >> 1, kf = i & 1;
            if (i + LA < 2 * KC) fr[(i + LA) % (LA + 1)] = frag_at(kidx(i + LA));
            else fr[(i + LA) % (LA + 1)] = frag_at(2 * KC + (i + LA - 2 * KC));  // (runs on into the first V^T fragments)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int qf = 0; qf < QF; ++qf) st[qf][kf] = mfma32_lp<DT>(fr[i % (LA + 1)], qB[qf][c], st[qf][kf]);  // S^T[key 16kf+4g+r][query li] - m
            __builtin_amdgcn_sched_barrier(0);
        }
        const bool first = general && k0 == gs;
        if (general && k0 + 32 > ge) {  // (wave-uniform)
#pragma unroll
            for (int qf = 0; qf < QF; ++qf)
#pragma unroll
                for (int kf = 0; kf < 2; ++kf)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (k0 + 16 * kf + 4 * g + r >= ge) st[qf][kf][r] = -__builtin_inff();
        }
        float mx[QF], mxa = -__builtin_inff();
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) {
            mx[qf] = max3(max3(max3(st[qf][0][0], st[qf][0][1], st[qf][0][2]), st[qf][0][3], st[qf][1][0]), max3(st[qf][1][1], st[qf][1][2], st[qf][1][3]), st[qf][1][1]);
            mxa = fmaxf(mxa, mx[qf]);
        }
        if (first || __any(mxa > 10.f)) {
#pragma unroll
            for (int qf = 0; qf < QF; ++qf) {
                if (!first && !__any(mx[qf] > 10.f)) continue;
                float d = xmax(mx[qf]);          // how far the block's maximum lies above the reference
                if (!first) {
                    d = fmaxf(d, 0.f);           // (the reference only rises afterwards)
                    const float alpha = __builtin_amdgcn_exp2f(-d);
                    ol[qf] *= alpha;
#pragma unroll
                    for (int nt = 0; nt < DC; ++nt) o[qf][nt] *= alpha;
                }
                negm[qf] -= d;
                st[qf][0] -= d;
                st[qf][1] -= d;
            }
        }
        f32x4 pB[QF];
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) {
#pragma unroll
            for (int kf = 0; kf < 2; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) st[qf][kf][r] = __builtin_amdgcn_exp2f(st[qf][kf][r]);
            pB[qf] = pack8<DT>(st[qf][0], st[qf][1]);  // keys {4g+r} U {16+4g+r} of the block: the order V^T blocks are stored in
        }
#pragma unroll
        for (int nt = 0; nt < DC; ++nt) {
            const int i = 2 * KC + nt;
            if (nt + LA < DC) fr[(i + LA) % (LA + 1)] = frag_at(i + LA);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int qf = 0; qf < QF; ++qf) o[qf][nt] = mfma32_lp<DT>(fr[i % (LA + 1)], pB[qf], o[qf][nt]);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) ol[qf] = mfma32_lp<DT>(onesA, pB[qf], ol[qf]);  // row 0: l[query li] += sum of the block's (16-bit rounded) weights
    };
    // ---- the group's blocks through a ring of NS LDS slots, fetched PD blocks ahead (one block ahead the landing of a fetch -- a
    //      microsecond under load -- was exposed behind every block: 52 % of the wave cycles waiting, PMC).  Block j is read from slot
    //      j % NS while block j + PD lands in slot (j + PD) % NS, whose last readers (block j + PD - NS <= j - 1) are all behind the
    //      previous barrier; ONE barrier per block.  The first and the last block (reference set-up / ragged keys) are peeled off so
    //      that the loop body is one straight path ----
    auto step = [&](const int j, auto general_c) {
        if (j + PD < nblk) gload(j + PD, (j + PD) % NS);
        attend(j % NS, gs + 32 * j, decltype(general_c)::value);
        // block j + 1 must have landed: the fetches of blocks j + 2 .. j + PD issued behind it may still be in flight (3 loads each)
        const int rem = nblk - 1 - j;
        if (rem >= PD) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (PD - 1)) : "memory");
        else if (rem == PD - 1 && PD >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PD >= 2 ? 3 * (PD - 2) : 0) : "memory");
        else landed();
        __syncthreads();
    };
    step(0, std::true_type{});
#pragma unroll 1
    for (int j = 1; j < nblk - 1; ++j) step(j, std::false_type{});
    if (nblk > 1) step(nblk - 1, std::true_type{});

    // ---- out-proj + residual + LN1, FFN, + residual + LN2, ONE query fragment at a time: all three side by side (as enc_layer_lp_k runs
    //      its four) need 316 registers; the weights (90 KB) are re-read per fragment, 1 % of what the key loop streams ----
#pragma unroll
    for (int qf = 0; qf < QF; ++qf) {
        f32x4 oB[1][KC];
        {
            const float inv = 1.f / xsum(ol[qf][0]);  // l sits in row 0 of the ones tile (lane g = 0, register 0); rows 4g of g > 0 are zero
#pragma unroll
            for (int c = 0; c < KC; ++c) oB[0][c] = pack8<DT>(o[qf][2 * c] * inv, o[qf][2 * c + 1] * inv);
        }
        f32x4 x1[DC];
        gemm_T_lp<DC, KC, 1, DT>(p.w_out_lp, vec + V::BOUT, csp, oB, li, g, [&](int nt, int, f32x4 a) {
            x1[nt] = nt < CSR ? ld4(p.src + (size_t)qrow[qf] * cs + 16 * nt + 4 * g) + a : a;  // (features >= cs: zero weights -> a = 0)
        });
        f32x4 x1B[1][KC];
        layer_norm<DC>(x1, vec + V::LN1W, vec + V::LN1B, p.d, p.ln_eps, g);
#pragma unroll
        for (int c = 0; c < KC; ++c) x1B[0][c] = pack8<DT>(x1[2 * c], x1[2 * c + 1]);
        f32x4 hB[1][FKC];
        {
            f32x4 hprev;
            gemm_T_lp<FC, KC, 1, DT>(p.w1_lp, vec + V::B1, csp, x1B, li, g, [&](int ft, int, f32x4 a) {
#pragma unroll
                for (int r = 0; r < 4; ++r) a[r] = fmaxf(a[r], 0.f);
                if (ft & 1) hB[0][ft >> 1] = pack8<DT>(hprev, a); else hprev = a;
            });
        }
        gemm_T_lp<DC, FKC, 1, DT>(p.w2_lp, vec + V::B2, dff, hB, li, g, [&](int nt, int, f32x4 a) { x1[nt] += a; });
        layer_norm<DC>(x1, vec + V::LN2W, vec + V::LN2B, p.d, p.ln_eps, g);
        const int qtok = q0 + qf * 16 + li;
        if (qtok < ge) {
#pragma unroll
            for (int nt = 0; nt < CSR; ++nt) *reinterpret_cast<f32x4*>(p.out + (size_t)qtok * cs + 16 * nt + 4 * g) = x1[nt];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

int fill(const i2r_encoder_desc* d, EncK& k) {
    I2R_CHECK_ARG(d && d->src && d->kbuf && d->vbuf && d->out && d->grp_off && d->w_in && d->b_in && d->w_out && d->b_out &&
                      d->ln1_w && d->ln1_b && d->w1 && d->b1 && d->w2 && d->b2 && d->ln2_w && d->ln2_b,
                  "i2r_encoder: null pointer");
    I2R_CHECK_ARG((d->cs == 96 || d->cs == 80) && d->dff_pad == 192, "i2r_encoder: cs=%d dff_pad=%d unsupported (96|80, 192)",
                  d->cs, d->dff_pad);
    I2R_CHECK_ARG(d->d > 0 && d->d <= d->cs && d->n_tok > 0 && d->n_grp > 0, "i2r_encoder: sizes");
    I2R_CHECK_ARG(d->out != d->src, "i2r_encoder: out aliases src");
    k.src = d->src; k.pos = d->pos; k.kbuf = d->kbuf; k.vbuf = d->vbuf; k.out = d->out; k.grp_off = d->grp_off;
    k.w_in = d->w_in; k.b_in = d->b_in; k.w_out = d->w_out; k.b_out = d->b_out; k.ln1_w = d->ln1_w; k.ln1_b = d->ln1_b;
    k.w1 = d->w1; k.b1 = d->b1; k.w2 = d->w2; k.b2 = d->b2; k.ln2_w = d->ln2_w; k.ln2_b = d->ln2_b;
    k.n_tok = d->n_tok; k.n_tok_pad = ((d->n_tok + 63) / 64) * 64 + 64; k.n_grp = d->n_grp; k.d = d->d; k.cs = d->cs;
    k.dff_pad = d->dff_pad; k.pos_period = d->pos_period; k.ln_eps = d->ln_eps; k.n_qblk = d->n_qtiles16;
    k.qscale = 1.0f / sqrtf((float)d->d);
    k.w_in_lp = d->w_in_lp; k.w_out_lp = d->w_out_lp; k.w1_lp = d->w1_lp; k.w2_lp = d->w2_lp; k.vec_lp = d->vec_lp;
    k.next_w_in = d->next_w_in; k.next_b_in = d->next_b_in; k.next_kbuf = d->next_kbuf; k.next_vbuf = d->next_vbuf;
    k.split_ws = d->split_ws; k.split_cnt = d->split_cnt; k.tiles_per_xcd = k.full_per_xcd = 0;
    k.stamp = nullptr;
#ifdef I2R_TUNING
    {   // tuning build only: I2R_ENC_STAMP=<device address of 32 B x waves>, I2R_ENC_STAMP_FUSED=1 stamps the layers with a fused K/V tail
        static const char* se = getenv("I2R_ENC_STAMP");
        static const bool fused = getenv("I2R_ENC_STAMP_FUSED") && atoi(getenv("I2R_ENC_STAMP_FUSED"));
        k.stamp = (se && fused == (d->next_w_in != nullptr)) ? (long long*)strtoull(se, nullptr, 0) : nullptr;
    }
#endif
    I2R_CHECK_ARG(!d->next_w_in || (d->next_b_in && d->next_kbuf && d->next_vbuf && d->next_kbuf != d->kbuf && d->next_vbuf != d->vbuf && d->dtype == 0),
                  "i2r_encoder: fused next-layer K/V needs its own buffers (fp32 mode only)");
    I2R_CHECK_ARG(d->dtype >= 0 && d->dtype <= 2, "i2r_encoder: dtype %d", d->dtype);
    if (d->dtype != 0)
        I2R_CHECK_ARG(d->w_in_lp && d->w_out_lp && d->w1_lp && d->w2_lp && d->vec_lp && d->n_qtiles16 > 0 && d->n_qtiles32 > 0 && d->n_qtiles64 > 0,
                      "i2r_encoder: the 16-bit MFMA mode needs the permuted 16-bit weights, vec_lp and the tile counts");
    return I2R_OK;
}

}  // namespace

extern "C" int i2r_encoder_kv(const i2r_encoder_desc* d, void* stream) {
    EncK k;
    int rc = fill(d, k);
    if (rc) return rc;
    if (d->dtype != 0) {
        // one workgroup per I2R_KV_TB consecutive 32-token blocks of a group (n_qtiles64 = sum of ceil(len / 64) = sum of ceil(blocks / 2))
        constexpr int TB = I2R_KV_TB;
        const unsigned nblk = (unsigned)(TB == 2 ? d->n_qtiles64 : d->n_qtiles32);
        I2R_CHECK_ARG(nblk > 0, "i2r_encoder_kv: n_qtiles%d", TB == 2 ? 64 : 32);
        const bool c6 = d->cs == 96;
        if (d->dtype == 1) {
            if (c6) i2r_launch((enc_kv_lp_k<6, 6, 1, TB>), dim3(nblk), dim3(64), 0, (hipStream_t)stream, k);
            else i2r_launch((enc_kv_lp_k<6, 5, 1, TB>), dim3(nblk), dim3(64), 0, (hipStream_t)stream, k);
        } else {
            if (c6) i2r_launch((enc_kv_lp_k<6, 6, 2, TB>), dim3(nblk), dim3(64), 0, (hipStream_t)stream, k);
            else i2r_launch((enc_kv_lp_k<6, 5, 2, TB>), dim3(nblk), dim3(64), 0, (hipStream_t)stream, k);
        }
    } else {
        I2R_CHECK_ARG(d->n_qtiles16 > 0, "i2r_encoder_kv: n_qtiles16");
        if (d->cs == 96)
            i2r_launch(enc_kv_k<6>, dim3((unsigned)d->n_qtiles16), dim3(256), 0, (hipStream_t)stream, k);
        else
            i2r_launch(enc_kv_k<5>, dim3((unsigned)d->n_qtiles16), dim3(256), 0, (hipStream_t)stream, k);
    }
    I2R_CHECK_LAUNCH("i2r_encoder_kv");
    return I2R_OK;
}

namespace {
template <int QF, int DT>
void launch_lp(const EncK& k, bool c6, unsigned grid, hipStream_t st) {
    if (c6) i2r_launch((enc_layer_lp_k<6, 12, QF, DT, 6>), dim3(grid), dim3(64), 0, st, k);
    else i2r_launch((enc_layer_lp_k<6, 12, QF, DT, 5>), dim3(grid), dim3(64), 0, st, k);
}
}  // namespace

extern "C" int i2r_encoder_layer(const i2r_encoder_desc* d, void* stream) {
    EncK k;
    int rc = fill(d, k);
    if (rc) return rc;
    if (d->dtype != 0) {
        // 64 queries per wave when that still gives >= 2 waves per SIMD-pair of the chip, else 16 (more, shorter waves)
        const bool big = d->n_qtiles64 >= 512;
        k.n_qblk = big ? d->n_qtiles64 : d->n_qtiles16;
        const unsigned grid = (unsigned)((k.n_qblk + 7) / 8 * 8);  // (XCD-banded work items inside the kernel)
        const bool c6 = d->cs == 96;
#ifdef I2R_TUNING
        static const int qf_env = getenv("I2R_ENC_QF") ? atoi(getenv("I2R_ENC_QF")) : 0;  // tuning switch: 2 = 32 queries per wave
        if (qf_env == 2 && big) {
            k.n_qblk = d->n_qtiles32;
            const unsigned g2 = (unsigned)((d->n_qtiles32 + 7) / 8 * 8);
            if (d->dtype == 1) launch_lp<2, 1>(k, c6, g2, (hipStream_t)stream); else launch_lp<2, 2>(k, c6, g2, (hipStream_t)stream);
            I2R_CHECK_LAUNCH("i2r_encoder_layer");
            return I2R_OK;
        }
#endif
        if (big && d->n_qtiles192 > 0) {  // long groups: four waves per workgroup share the K / V stream through LDS (two workgroups per CU)
            k.n_qblk = d->n_qtiles192;
            const unsigned g4 = (unsigned)((d->n_qtiles192 + 7) / 8 * 8);
            if (d->dtype == 1) {
                if (c6) i2r_launch((enc_layer_lp4_k<6, 12, 1, 6>), dim3(g4), dim3(256), 0, (hipStream_t)stream, k);
                else i2r_launch((enc_layer_lp4_k<6, 12, 1, 5>), dim3(g4), dim3(256), 0, (hipStream_t)stream, k);
            } else {
                if (c6) i2r_launch((enc_layer_lp4_k<6, 12, 2, 6>), dim3(g4), dim3(256), 0, (hipStream_t)stream, k);
                else i2r_launch((enc_layer_lp4_k<6, 12, 2, 5>), dim3(g4), dim3(256), 0, (hipStream_t)stream, k);
            }
        } else if (d->dtype == 1) {
            if (big) launch_lp<4, 1>(k, c6, grid, (hipStream_t)stream); else launch_lp<1, 1>(k, c6, grid, (hipStream_t)stream);
        } else {
            if (big) launch_lp<4, 2>(k, c6, grid, (hipStream_t)stream); else launch_lp<1, 2>(k, c6, grid, (hipStream_t)stream);
        }
        I2R_CHECK_LAUNCH("i2r_encoder_layer");
        return I2R_OK;
    }
    I2R_CHECK_ARG(d->n_qtiles16 > 0 && d->n_qtiles32 > 0, "i2r_encoder_layer: n_qtiles16 / n_qtiles32");
    // two 16-query tiles per workgroup halve the K / V traffic per MFMA; worth it once there are enough workgroups for the chip
#ifdef I2R_TUNING
    static const int qt_env = getenv("I2R_ENC_QT") ? atoi(getenv("I2R_ENC_QT")) : 0;  // tuning switch: force 1 or 2
#else
    constexpr int qt_env = 0;
#endif
    const int qt = qt_env ? qt_env : (d->n_qtiles32 >= 512 ? 2 : 1);
    k.n_qblk = qt == 2 ? d->n_qtiles32 : d->n_qtiles16;
    unsigned grid = (unsigned)((k.n_qblk + 7) / 8 * 8);  // (XCD-major tile order inside the kernel)
    // between one and two tiles per CU (256 CUs = 8 XCDs x 32): split just enough tiles by keys that every CU gets two workgroups
    const int t_x = (d->n_qtiles16 + 7) / 8;
    const bool split = qt == 1 && d->split_ws && d->split_cnt && t_x > 32 && t_x < 64;
    if (split) {
        k.tiles_per_xcd = t_x;
        k.full_per_xcd = 2 * t_x - 64;
        grid = 512;
    }
    if (d->cs == 96) {
        if (qt == 2) i2r_launch((enc_layer4_k<6, 12, 2>), dim3(grid), dim3(256), 0, (hipStream_t)stream, k);
        else if (split) i2r_launch((enc_layer4_k<6, 12, 1, 2>), dim3(grid), dim3(256), 0, (hipStream_t)stream, k);
        else i2r_launch((enc_layer4_k<6, 12, 1>), dim3(grid), dim3(256), 0, (hipStream_t)stream, k);
    } else {
        if (qt == 2) i2r_launch((enc_layer4_k<5, 12, 2>), dim3(grid), dim3(256), 0, (hipStream_t)stream, k);
        else if (split) i2r_launch((enc_layer4_k<5, 12, 1, 2>), dim3(grid), dim3(256), 0, (hipStream_t)stream, k);
        else i2r_launch((enc_layer4_k<5, 12, 1>), dim3(grid), dim3(256), 0, (hipStream_t)stream, k);
    }
    I2R_CHECK_LAUNCH("i2r_encoder_layer");
    return I2R_OK;
}
