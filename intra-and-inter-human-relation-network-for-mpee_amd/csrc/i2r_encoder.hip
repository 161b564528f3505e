// Fused DETR-style encoder layer for variable-length token groups (inter-human relation module and the
// TransPose-H intra-human encoder), fp32 on the matrix pipe.
//
// Formulation ("token-per-lane-column"): every GEMM is computed TRANSPOSED,  Y^T[f][t] = W[f][:] . X^T[:][t],
// with the weight matrix as the MFMA A operand (read as 16-byte rows of the reference's own row-major
// [out][in] nn.Linear weights) and the activations as the B operand.  The 16x16x4 fp32 MFMA returns
// D with col = l&15 = token and rows 4*(l>>4)+r = feature -- which is exactly the B-operand register
// image the NEXT GEMM needs (token = l&15, k-group = l>>4, 4 consecutive features per float4).  So
// q-proj -> S^T = K Q^T -> softmax -> O^T = V^T P^T -> out-proj -> +res -> LN1 -> FFN1 -> ReLU -> FFN2 -> +res -> LN2
// chains register-to-register: no LDS transposes, no intermediate HBM round trips.  Only K / V^T tiles of the
// group go through LDS (shared by the waves of a workgroup).  Softmax / LayerNorm reductions over features
// or keys are in-lane sums + two __shfl_xor (16, 32).
#include <stdlib.h>

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
    int n_tok, n_tok_pad, n_grp, d, cs, dff_pad, pos_period;
    float ln_eps, qscale;
    // 16-bit MFMA mode: weights as bf16/f16 [out][in] with the columns of every 32-block permuted to the MFMA operand order
    const void* w_in_lp; const void* w_out_lp; const void* w1_lp; const void* w2_lp;
    // fused K/V projection of the NEXT layer (enc_layer4_k): its in_proj (k, v rows used), destination buffers; null = none
    const float* next_w_in; const float* next_b_in; float* next_kbuf; float* next_vbuf;
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

// ---- K / V projection: one wave = 32 tokens ----
template <int DC>
__global__ __launch_bounds__(64) void enc_kv_k(const EncK p) {
    const int lane = threadIdx.x, li = lane & 15, g = lane >> 4;
    const int cs = DC * 16;
    const int t0 = blockIdx.x * 32;
    f32x4 xs[2][DC], xq[2][DC];
    int tok[2];
#pragma unroll
    for (int tf = 0; tf < 2; ++tf) {
        tok[tf] = t0 + tf * 16 + li;
        const int row = min(tok[tf], p.n_tok - 1);
        const int prow = p.pos_period > 0 ? row % p.pos_period : row;
#pragma unroll
        for (int c = 0; c < DC; ++c) {
            xs[tf][c] = ld4(p.src + (size_t)row * cs + 16 * c + 4 * g);
            xq[tf][c] = xs[tf][c];
            if (p.pos) xq[tf][c] += ld4(p.pos + (size_t)prow * cs + 16 * c + 4 * g);
        }
    }
    // K rows of w_in: [cs, 2cs); V rows: [2cs, 3cs)
#pragma unroll
    for (int nt = 0; nt < DC; ++nt) {
        f32x4 ak[2], av[2];
        ak[0] = ak[1] = ld4(p.b_in + cs + 16 * nt + 4 * g);
        av[0] = av[1] = ld4(p.b_in + 2 * cs + 16 * nt + 4 * g);
#pragma unroll
        for (int c = 0; c < DC; ++c) {
            const f32x4 wk = ld4(p.w_in + (size_t)(cs + 16 * nt + li) * cs + 16 * c + 4 * g);
            const f32x4 wv = ld4(p.w_in + (size_t)(2 * cs + 16 * nt + li) * cs + 16 * c + 4 * g);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int tf = 0; tf < 2; ++tf) {
                    ak[tf] = mfma16(wk[s], xq[tf][c][s], ak[tf]);
                    av[tf] = mfma16(wv[s], xs[tf][c][s], av[tf]);
                }
            }
        }
#pragma unroll
        for (int tf = 0; tf < 2; ++tf) {
            if (tok[tf] < p.n_tok) {
                *reinterpret_cast<f32x4*>(p.kbuf + (size_t)tok[tf] * cs + 16 * nt + 4 * g) = ak[tf];
#pragma unroll
                for (int r = 0; r < 4; ++r) p.vbuf[(size_t)(16 * nt + 4 * g + r) * p.n_tok_pad + tok[tf]] = av[tf][r];
            }
        }
    }
}

// LayerNorm over the real d features of each token column (features live in y[nt][r] x 4 lanes)
template <int DC>
__device__ __forceinline__ void layer_norm(f32x4 (&y)[DC], const float* w, const float* b, int d, float eps, int g) {
    float s = 0.f;
#pragma unroll
    for (int nt = 0; nt < DC; ++nt) s += (y[nt][0] + y[nt][1]) + (y[nt][2] + y[nt][3]);
    const float mean = xsum(s) / (float)d;
    float v = 0.f;
#pragma unroll
    for (int nt = 0; nt < DC; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float t = (16 * nt + 4 * g + r < d) ? y[nt][r] - mean : 0.f;
            v += t * t;
        }
    const float rstd = rsqrtf(xsum(v) / (float)d + eps);
#pragma unroll
    for (int nt = 0; nt < DC; ++nt) {
        const f32x4 wv = ld4(w + 16 * nt + 4 * g), bv = ld4(b + 16 * nt + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) y[nt][r] = (y[nt][r] - mean) * rstd * wv[r] + bv[r];
    }
}

// one output fragment (16 features x 16 tokens) of  Y^T = W X^T (+bias):  acc += sum_c W[16nt+li][16c+4g..] * x[c]
template <int KC>
__device__ __forceinline__ void load_wrow(f32x4 (&wr)[KC], const float* W, int ld, int nt, int li, int g) {
    const float* row = W + (size_t)(16 * nt + li) * ld + 4 * g;
#pragma unroll
    for (int c = 0; c < KC; ++c) wr[c] = ld4(row + 16 * c);
}
template <int KC>
__device__ __forceinline__ f32x4 frag_mm(const f32x4 (&wr)[KC], const f32x4 (&x)[KC], f32x4 acc) {
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = mfma16(wr[c][s], x[c][s], acc);
    return acc;
}
// Y^T[NF frags] = W[NF*16, KC*16] X^T + b, weight rows software-pipelined one fragment ahead (two named register sets)
template <int NF, int KC, typename Epi>
__device__ __forceinline__ void gemm_T(const float* W, const float* bias, int ld, const f32x4 (&x)[KC], int li, int g, Epi epi) {
    f32x4 w0[KC], w1[KC];
    load_wrow<KC>(w0, W, ld, 0, li, g);
#pragma unroll
    for (int nt = 0; nt < NF; nt += 2) {
        if (nt + 1 < NF) load_wrow<KC>(w1, W, ld, nt + 1, li, g);
        epi(nt, frag_mm<KC>(w0, x, ld4(bias + 16 * nt + 4 * g)));
        if (nt + 1 < NF) {
            if (nt + 2 < NF) load_wrow<KC>(w0, W, ld, nt + 2, li, g);
            epi(nt + 1, frag_mm<KC>(w1, x, ld4(bias + 16 * (nt + 1) + 4 * g)));
        }
    }
}

// ---- attention + output projection + LN1 + FFN + LN2: NW waves x 16 queries; K / V^T fragments stream from L2
//      straight into registers (no LDS, no barrier), prefetched one 16-key fragment ahead ----
template <int DC, int FC, int NW>
__global__ __launch_bounds__(NW * 64) void enc_layer_k(const EncK p) {
    constexpr int cs = DC * 16, dff = FC * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    constexpr int QT = NW * 16;

    // which (group, query tile) is this workgroup?
    int b = blockIdx.x, gs = 0, ge = 0;
    for (int grp = 0; grp < p.n_grp; ++grp) {
        gs = p.grp_off[grp];
        ge = p.grp_off[grp + 1];
        const int nq = (ge - gs + QT - 1) / QT;
        if (b < nq) break;
        b -= nq;
    }
    const int q0 = gs + b * QT + wave * 16;
    if (q0 >= ge) return;  // (whole wave past the end of the group)
    const int qtok = q0 + li;
    const bool qvalid = qtok < ge;
    const int qrow = qvalid ? qtok : ge - 1;
    const int prow = p.pos_period > 0 ? qrow % p.pos_period : qrow;

    f32x4 xs[DC], q[DC];
    {
        f32x4 xq[DC];
#pragma unroll
        for (int c = 0; c < DC; ++c) {
            xs[c] = ld4(p.src + (size_t)qrow * cs + 16 * c + 4 * g);
            xq[c] = xs[c];
            if (p.pos) xq[c] += ld4(p.pos + (size_t)prow * cs + 16 * c + 4 * g);
        }
        gemm_T<DC, DC>(p.w_in, p.b_in, cs, xq, li, g, [&](int nt, f32x4 a) { q[nt] = a * p.qscale; });
    }

    f32x4 o[DC];
#pragma unroll
    for (int nt = 0; nt < DC; ++nt) o[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -__builtin_inff(), l_run = 0.f;

    // A operands of one 16-key fragment: K rows (lane = key li) and V^T rows (lane = dim li), 4 consecutive k each
    auto fetch_kv = [&](int k0, f32x4(&ka)[DC], f32x4(&va)[DC]) {
        const int krow = min(k0 + li, ge - 1);  // clamp: rows past the group are masked below
        const float* kp = p.kbuf + (size_t)krow * cs + 4 * g;
        const float* vp = p.vbuf + (size_t)li * p.n_tok_pad + k0 + 4 * g;
#pragma unroll
        for (int c = 0; c < DC; ++c) {
            ka[c] = ld4(kp + 16 * c);
            va[c] = ld4(vp + (size_t)16 * c * p.n_tok_pad);
        }
    };
    auto attend = [&](int k0, const f32x4(&ka)[DC], const f32x4(&va)[DC]) {
        f32x4 st = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < DC; ++c)
#pragma unroll
            for (int s = 0; s < 4; ++s) st = mfma16(ka[c][s], q[c][s], st);  // S^T[key 4g+r][query li]
        float mx = -__builtin_inff();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (k0 + 4 * g + r >= ge) st[r] = -__builtin_inff();
            mx = fmaxf(mx, st[r]);
        }
        mx = xmax(mx);
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __expf(m_run - m_new);
        float ls = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            st[r] = __expf(st[r] - m_new);
            ls += st[r];
        }
        l_run = l_run * alpha + ls;
        m_run = m_new;
#pragma unroll
        for (int nt = 0; nt < DC; ++nt) {
            f32x4 acc = o[nt] * alpha;
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = mfma16(va[nt][s], st[s], acc);  // O^T[dim][query] += V^T[dim][key] P^T
            o[nt] = acc;
        }
    };
    {
        f32x4 ka0[DC], va0[DC], ka1[DC], va1[DC];
        fetch_kv(gs, ka0, va0);
        int k0 = gs;
        for (; k0 + 32 <= ge; k0 += 32) {
            fetch_kv(k0 + 16, ka1, va1);
            attend(k0, ka0, va0);
            fetch_kv(min(k0 + 32, ge - 1) & ~3, ka0, va0);  // (look-ahead past the end is clamped and unused)
            attend(k0 + 16, ka1, va1);
        }
        if (k0 < ge) {
            if (k0 + 16 < ge) fetch_kv(k0 + 16, ka1, va1);
            attend(k0, ka0, va0);
            if (k0 + 16 < ge) attend(k0 + 16, ka1, va1);
        }
    }
    {
        const float inv = 1.f / xsum(l_run);
#pragma unroll
        for (int nt = 0; nt < DC; ++nt) o[nt] *= inv;
    }

    // out-proj + residual + LN1
    f32x4 x1[DC];
    gemm_T<DC, DC>(p.w_out, p.b_out, cs, o, li, g, [&](int nt, f32x4 a) { x1[nt] = xs[nt] + a; });
    layer_norm<DC>(x1, p.ln1_w, p.ln1_b, p.d, p.ln_eps, g);

    // FFN
    f32x4 h[FC];
    gemm_T<FC, DC>(p.w1, p.b1, cs, x1, li, g, [&](int ft, f32x4 a) {
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = fmaxf(a[r], 0.f);
        h[ft] = a;
    });
    f32x4 y[DC];
    gemm_T<DC, FC>(p.w2, p.b2, dff, h, li, g, [&](int nt, f32x4 a) { y[nt] = x1[nt] + a; });
    layer_norm<DC>(y, p.ln2_w, p.ln2_b, p.d, p.ln_eps, g);
    if (qvalid) {
#pragma unroll
        for (int nt = 0; nt < DC; ++nt) *reinterpret_cast<f32x4*>(p.out + (size_t)qtok * cs + 16 * nt + 4 * g) = y[nt];
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
template <int KC>
__device__ __forceinline__ f32x4 row_mm(const float* W, int ld, int nt, const f32x4 (&x)[KC], f32x4 acc, int li, int g) {
    f32x4 w[KC];
    load_wrow<KC>(w, W, ld, nt, li, g);
    return frag_mm<KC>(w, x, acc);
}

template <int DC, int FC>
__global__ __launch_bounds__(256) void enc_layer4_k(const EncK p) {
    constexpr int cs = DC * 16, dff = FC * 16;
    // LDS (float4 units): exchange area X[max(FC,DC)][64] + partial-O area O[4][DC][64] + (m,l) area ML[4][2][16]
    __shared__ f32x4 Xs[(FC > DC ? FC : DC) * 64];
    __shared__ f32x4 Os[4 * DC * 64];
    __shared__ float MLs[4 * 2 * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g = lane >> 4;

    int b = blockIdx.x, gs = 0, ge = 0;
    for (int grp = 0; grp < p.n_grp; ++grp) {
        gs = p.grp_off[grp];
        ge = p.grp_off[grp + 1];
        const int nq = (ge - gs + 15) >> 4;
        if (b < nq) break;
        b -= nq;
    }
    const int qtok = gs + b * 16 + li;
    const bool qvalid = qtok < ge;
    const int qrow = qvalid ? qtok : ge - 1;
    const int prow = p.pos_period > 0 ? qrow % p.pos_period : qrow;

    // ---- inputs (every wave keeps the tile's src and src+pos fragments: 2 x DC float4) ----
    f32x4 xs[DC], xq[DC];
#pragma unroll
    for (int c = 0; c < DC; ++c) {
        xs[c] = ld4(p.src + (size_t)qrow * cs + 16 * c + 4 * g);
        xq[c] = xs[c];
        if (p.pos) xq[c] += ld4(p.pos + (size_t)prow * cs + 16 * c + 4 * g);
    }
    // ---- q projection, output fragments split over the waves, exchanged through LDS ----
    for (int nt = wave; nt < DC; nt += 4)
        Xs[nt * 64 + lane] = row_mm<DC>(p.w_in, cs, nt, xq, ld4(p.b_in + 16 * nt + 4 * g), li, g) * p.qscale;
    __syncthreads();
    f32x4 q[DC];
#pragma unroll
    for (int c = 0; c < DC; ++c) q[c] = Xs[c * 64 + lane];

    // ---- attention over this wave's key fragments (kf = wave, wave+4, ...), prefetched one fragment ahead ----
    f32x4 o[DC];
#pragma unroll
    for (int nt = 0; nt < DC; ++nt) o[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -__builtin_inff(), l_run = 0.f;
    auto fetch_kv = [&](int k0, f32x4(&ka)[DC], f32x4(&va)[DC]) {
        const int krow = min(k0 + li, ge - 1);
        const float* kp = p.kbuf + (size_t)krow * cs + 4 * g;
        const float* vp = p.vbuf + (size_t)li * p.n_tok_pad + k0 + 4 * g;
#pragma unroll
        for (int c = 0; c < DC; ++c) {
            ka[c] = ld4(kp + 16 * c);
            va[c] = ld4(vp + (size_t)16 * c * p.n_tok_pad);
        }
    };
    auto attend = [&](int k0, const f32x4(&ka)[DC], const f32x4(&va)[DC]) {
        f32x4 st = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < DC; ++c)
#pragma unroll
            for (int s = 0; s < 4; ++s) st = mfma16(ka[c][s], q[c][s], st);
        float mx = -__builtin_inff();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (k0 + 4 * g + r >= ge) st[r] = -__builtin_inff();
            mx = fmaxf(mx, st[r]);
        }
        mx = xmax(mx);
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __expf(m_run - m_new);
        float ls = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            st[r] = __expf(st[r] - m_new);
            ls += st[r];
        }
        l_run = l_run * alpha + ls;
        m_run = m_new;
#pragma unroll
        for (int nt = 0; nt < DC; ++nt) {
            f32x4 acc = o[nt] * alpha;
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = mfma16(va[nt][s], st[s], acc);
            o[nt] = acc;
        }
    };
    {
        f32x4 ka0[DC], va0[DC], ka1[DC], va1[DC];
        int k0 = gs + wave * 16;
        if (k0 < ge) fetch_kv(k0, ka0, va0);
        for (; k0 + 64 < ge; k0 += 128) {          // two fragments (k0, k0+64) per trip, each prefetched under the other
            fetch_kv(k0 + 64, ka1, va1);
            attend(k0, ka0, va0);
            if (k0 + 128 < ge) fetch_kv(k0 + 128, ka0, va0);
            attend(k0 + 64, ka1, va1);
        }
        if (k0 < ge) attend(k0, ka0, va0);
    }
    // ---- merge the four partial softmax states ----
    l_run = xsum(l_run);
    if (g == 0) {
        MLs[(wave * 2 + 0) * 16 + li] = m_run;
        MLs[(wave * 2 + 1) * 16 + li] = l_run;
    }
#pragma unroll
    for (int nt = 0; nt < DC; ++nt) Os[(wave * DC + nt) * 64 + lane] = o[nt];
    __syncthreads();
    f32x4 oc[DC];
    {
        float mw[4], m = -__builtin_inff();
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            mw[w] = MLs[(w * 2) * 16 + li];
            m = fmaxf(m, mw[w]);
        }
        float l = 0.f, sc[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            sc[w] = __expf(mw[w] - m);  // a wave without keys has m = -inf, l = 0, O = 0 -> scale 0
            l += MLs[(w * 2 + 1) * 16 + li] * sc[w];
        }
        const float inv = 1.f / l;
#pragma unroll
        for (int nt = 0; nt < DC; ++nt) {
            f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < 4; ++w) a += Os[(w * DC + nt) * 64 + lane] * sc[w];
            oc[nt] = a * inv;
        }
    }
    // ---- out-proj + residual (split by output fragment) -> LDS -> LayerNorm 1 on the full row in every wave ----
    for (int nt = wave; nt < DC; nt += 4) {
        f32x4 xsn;
#pragma unroll
        for (int c = 0; c < DC; ++c)
            if (c == nt) xsn = xs[c];
        Xs[nt * 64 + lane] = xsn + row_mm<DC>(p.w_out, cs, nt, oc, ld4(p.b_out + 16 * nt + 4 * g), li, g);
    }
    __syncthreads();
    f32x4 x1[DC];
#pragma unroll
    for (int c = 0; c < DC; ++c) x1[c] = Xs[c * 64 + lane];
    layer_norm<DC>(x1, p.ln1_w, p.ln1_b, p.d, p.ln_eps, g);
    __syncthreads();  // everyone has read Xs before FFN1 overwrites it
    // ---- FFN ----
    for (int ft = wave; ft < FC; ft += 4) {
        f32x4 a = row_mm<DC>(p.w1, cs, ft, x1, ld4(p.b1 + 16 * ft + 4 * g), li, g);
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = fmaxf(a[r], 0.f);
        Xs[ft * 64 + lane] = a;
    }
    __syncthreads();
    f32x4 h[FC];
#pragma unroll
    for (int c = 0; c < FC; ++c) h[c] = Xs[c * 64 + lane];
    __syncthreads();
    for (int nt = wave; nt < DC; nt += 4) {
        f32x4 x1n;
#pragma unroll
        for (int c = 0; c < DC; ++c)
            if (c == nt) x1n = x1[c];
        Xs[nt * 64 + lane] = x1n + row_mm<FC>(p.w2, dff, nt, h, ld4(p.b2 + 16 * nt + 4 * g), li, g);
    }
    __syncthreads();
    f32x4 y[DC];
#pragma unroll
    for (int c = 0; c < DC; ++c) y[c] = Xs[c * 64 + lane];
    layer_norm<DC>(y, p.ln2_w, p.ln2_b, p.d, p.ln_eps, g);
    if (qvalid) {
        for (int nt = wave; nt < DC; nt += 4) {
            f32x4 yn;
#pragma unroll
            for (int c = 0; c < DC; ++c)
                if (c == nt) yn = y[c];
            *reinterpret_cast<f32x4*>(p.out + (size_t)qtok * cs + 16 * nt + 4 * g) = yn;
        }
    }
    // ---- K / V of the next layer from the layer output still in registers ----
    if (p.next_w_in) {
        f32x4 yq[DC];
#pragma unroll
        for (int c = 0; c < DC; ++c) {
            yq[c] = y[c];
            if (p.pos) yq[c] += ld4(p.pos + (size_t)prow * cs + 16 * c + 4 * g);
        }
        for (int f = wave; f < 2 * DC; f += 4) {  // fragments 0..DC-1: K rows, DC..2DC-1: V rows
            const bool isv = f >= DC;
            const int nt = isv ? f - DC : f;
            const int wrow = (isv ? 2 : 1) * cs;
            const f32x4 a = row_mm<DC>(p.next_w_in + (size_t)wrow * cs, cs, nt, isv ? y : yq, ld4(p.next_b_in + wrow + 16 * nt + 4 * g), li, g);
            if (qvalid) {
                if (!isv) {
                    *reinterpret_cast<f32x4*>(p.next_kbuf + (size_t)qtok * cs + 16 * nt + 4 * g) = a;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) p.next_vbuf[(size_t)(16 * nt + 4 * g + r) * p.n_tok_pad + qtok] = a[r];
                }
            }
        }
    }
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

// K (permuted rows, 16-bit) and V^T (permuted 32-key blocks, 16-bit); projections themselves stay on the exact-fp32 MFMA
template <int DC, int DT>
__global__ __launch_bounds__(64) void enc_kv_lp_k(const EncK p) {
    const int lane = threadIdx.x, li = lane & 15, g = lane >> 4;
    const int cs = DC * 16;
    const int t0 = blockIdx.x * 32;
    f32x4 xs[2][DC], xq[2][DC];
    int tok[2];
#pragma unroll
    for (int tf = 0; tf < 2; ++tf) {
        tok[tf] = t0 + tf * 16 + li;
        const int row = min(tok[tf], p.n_tok - 1);
        const int prow = p.pos_period > 0 ? row % p.pos_period : row;
#pragma unroll
        for (int c = 0; c < DC; ++c) {
            xs[tf][c] = ld4(p.src + (size_t)row * cs + 16 * c + 4 * g);
            xq[tf][c] = xs[tf][c];
            if (p.pos) xq[tf][c] += ld4(p.pos + (size_t)prow * cs + 16 * c + 4 * g);
        }
    }
    unsigned short* k16 = reinterpret_cast<unsigned short*>(p.kbuf);
    unsigned short* v16 = reinterpret_cast<unsigned short*>(p.vbuf);
#pragma unroll
    for (int c = 0; c < DC / 2; ++c) {
        f32x4 ak[2][2], av[2][2];  // [half][tf]
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int nt = 2 * c + half;
            ak[half][0] = ak[half][1] = ld4(p.b_in + cs + 16 * nt + 4 * g);
            av[half][0] = av[half][1] = ld4(p.b_in + 2 * cs + 16 * nt + 4 * g);
#pragma unroll
            for (int cc = 0; cc < DC; ++cc) {
                const f32x4 wk = ld4(p.w_in + (size_t)(cs + 16 * nt + li) * cs + 16 * cc + 4 * g);
                const f32x4 wv = ld4(p.w_in + (size_t)(2 * cs + 16 * nt + li) * cs + 16 * cc + 4 * g);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int tf = 0; tf < 2; ++tf) {
                        ak[half][tf] = mfma16(wk[s], xq[tf][cc][s], ak[half][tf]);
                        av[half][tf] = mfma16(wv[s], xs[tf][cc][s], av[half][tf]);
                    }
            }
        }
#pragma unroll
        for (int tf = 0; tf < 2; ++tf) {
            if (tok[tf] >= p.n_tok) continue;
            // K row: 8 consecutive 16-bit elements = this lane's slice of 32-block c
            *reinterpret_cast<f32x4*>(k16 + (size_t)tok[tf] * cs + c * 32 + g * 8) = pack8<DT>(ak[0][tf], ak[1][tf]);
            // V^T: element (feature, key) lives at  feature * n_tok_pad + 32*(key/32) + 8*((key%16)/4) + 4*((key%32)/16) + key%4
            const int t = tok[tf];
            const size_t kpos = (size_t)(t & ~31) + 8 * ((t & 15) >> 2) + 4 * ((t & 31) >> 4) + (t & 3);
            const f32x4 pv = pack8<DT>(av[0][tf], av[1][tf]);  // elements 0-3: features 32c+4g+r, 4-7: 32c+16+4g+r
            const unsigned short* e = reinterpret_cast<const unsigned short*>(&pv);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v16[(size_t)(32 * c + 4 * g + r) * p.n_tok_pad + kpos] = e[r];
                v16[(size_t)(32 * c + 16 + 4 * g + r) * p.n_tok_pad + kpos] = e[4 + r];
            }
        }
    }
}

// Y^T[NF frags] = W16[NF*16, KC*32] X^T + b for QF token fragments at once (the weight rows are fetched once per fragment row)
template <int NF, int KC, int QF, int DT, typename Epi>
__device__ __forceinline__ void gemm_T_lp(const void* W, const float* bias, int ld, const f32x4 (&x)[QF][KC], int li, int g, Epi epi) {
    f32x4 w0[KC], w1[KC];
    auto loadw = [&](f32x4(&w)[KC], int nt) {
#pragma unroll
        for (int c = 0; c < KC; ++c) w[c] = ld16(W, (size_t)(16 * nt + li) * ld + c * 32 + g * 8);
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

template <int DC, int FC, int QF, int DT>
__global__ __launch_bounds__(64) void enc_layer_lp_k(const EncK p) {
    constexpr int cs = DC * 16, dff = FC * 16, KC = DC / 2, FKC = FC / 2;
    const int lane = threadIdx.x, li = lane & 15, g = lane >> 4;
    constexpr int QT = QF * 16;
    int b = blockIdx.x, gs = 0, ge = 0;
    for (int grp = 0; grp < p.n_grp; ++grp) {
        gs = p.grp_off[grp];
        ge = p.grp_off[grp + 1];
        const int nq = (ge - gs + QT - 1) / QT;
        if (b < nq) break;
        b -= nq;
    }
    const int q0 = gs + b * QT;
    int qrow[QF];
#pragma unroll
    for (int qf = 0; qf < QF; ++qf) qrow[qf] = min(q0 + qf * 16 + li, ge - 1);

    // ---- q projection: B operand = packed (src + pos) ----
    f32x4 qB[QF][KC];
    {
        f32x4 xB[QF][KC];
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) {
            const int prow = p.pos_period > 0 ? qrow[qf] % p.pos_period : qrow[qf];
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                f32x4 a0 = ld4(p.src + (size_t)qrow[qf] * cs + 32 * c + 4 * g), a1 = ld4(p.src + (size_t)qrow[qf] * cs + 32 * c + 16 + 4 * g);
                if (p.pos) {
                    a0 += ld4(p.pos + (size_t)prow * cs + 32 * c + 4 * g);
                    a1 += ld4(p.pos + (size_t)prow * cs + 32 * c + 16 + 4 * g);
                }
                xB[qf][c] = pack8<DT>(a0, a1);
            }
        }
        f32x4 q32[QF][DC];
        gemm_T_lp<DC, KC, QF, DT>(p.w_in_lp, p.b_in, cs, xB, li, g, [&](int nt, int qf, f32x4 a) { q32[qf][nt] = a * p.qscale; });
#pragma unroll
        for (int qf = 0; qf < QF; ++qf)
#pragma unroll
            for (int c = 0; c < KC; ++c) qB[qf][c] = pack8<DT>(q32[qf][2 * c], q32[qf][2 * c + 1]);
    }

    // ---- flash attention over 32-key blocks; K rows / V^T blocks stream from L2 into registers, one block ahead ----
    f32x4 o[QF][DC];
    float m_run[QF], l_run[QF];
#pragma unroll
    for (int qf = 0; qf < QF; ++qf) {
        m_run[qf] = -__builtin_inff();
        l_run[qf] = 0.f;
#pragma unroll
        for (int nt = 0; nt < DC; ++nt) o[qf][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    auto fetch_kv = [&](int k0, f32x4(&ka)[2][KC], f32x4(&va)[DC]) {
#pragma unroll
        for (int kf = 0; kf < 2; ++kf) {
            const int krow = min(k0 + 16 * kf + li, ge - 1);
#pragma unroll
            for (int c = 0; c < KC; ++c) ka[kf][c] = ld16(p.kbuf, (size_t)krow * cs + c * 32 + g * 8);
        }
#pragma unroll
        for (int nt = 0; nt < DC; ++nt) va[nt] = ld16(p.vbuf, (size_t)(16 * nt + li) * p.n_tok_pad + k0 + g * 8);
    };
    auto attend = [&](int k0, const f32x4(&ka)[2][KC], const f32x4(&va)[DC]) {
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) {
            f32x4 st[2];
            float mx = -__builtin_inff();
#pragma unroll
            for (int kf = 0; kf < 2; ++kf) {
                f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < KC; ++c) a = mfma32_lp<DT>(ka[kf][c], qB[qf][c], a);  // S^T[key 16kf+4g+r][query li]
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (k0 + 16 * kf + 4 * g + r >= ge) a[r] = -__builtin_inff();
                    mx = fmaxf(mx, a[r]);
                }
                st[kf] = a;
            }
            mx = xmax(mx);
            const float m_new = fmaxf(m_run[qf], mx);
            const float alpha = __expf(m_run[qf] - m_new);
            float ls = 0.f;
#pragma unroll
            for (int kf = 0; kf < 2; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    st[kf][r] = __expf(st[kf][r] - m_new);
                    ls += st[kf][r];
                }
            l_run[qf] = l_run[qf] * alpha + ls;
            m_run[qf] = m_new;
            const f32x4 pB = pack8<DT>(st[0], st[1]);  // keys {4g+r} U {16+4g+r} of the block: the order V^T blocks are stored in
#pragma unroll
            for (int nt = 0; nt < DC; ++nt) o[qf][nt] = mfma32_lp<DT>(va[nt], pB, o[qf][nt] * alpha);
        }
    };
    {
        f32x4 ka0[2][KC], va0[DC], ka1[2][KC], va1[DC];
        fetch_kv(gs, ka0, va0);
        int k0 = gs;
        for (; k0 + 64 <= ge; k0 += 64) {
            fetch_kv(k0 + 32, ka1, va1);
            attend(k0, ka0, va0);
            fetch_kv(min(k0 + 64, (ge - 1) & ~31), ka0, va0);  // (look-ahead past the end re-reads the last block, unused)
            attend(k0 + 32, ka1, va1);
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
        const float inv = 1.f / xsum(l_run[qf]);
#pragma unroll
        for (int c = 0; c < KC; ++c) oB[qf][c] = pack8<DT>(o[qf][2 * c] * inv, o[qf][2 * c + 1] * inv);
    }
    f32x4 x1[QF][DC];
    gemm_T_lp<DC, KC, QF, DT>(p.w_out_lp, p.b_out, cs, oB, li, g, [&](int nt, int qf, f32x4 a) {
        x1[qf][nt] = ld4(p.src + (size_t)qrow[qf] * cs + 16 * nt + 4 * g) + a;
    });
    f32x4 x1B[QF][KC];
#pragma unroll
    for (int qf = 0; qf < QF; ++qf) {
        layer_norm<DC>(x1[qf], p.ln1_w, p.ln1_b, p.d, p.ln_eps, g);
#pragma unroll
        for (int c = 0; c < KC; ++c) x1B[qf][c] = pack8<DT>(x1[qf][2 * c], x1[qf][2 * c + 1]);
    }
    f32x4 hB[QF][FKC];
    {
        f32x4 hprev[QF];
        gemm_T_lp<FC, KC, QF, DT>(p.w1_lp, p.b1, cs, x1B, li, g, [&](int ft, int qf, f32x4 a) {
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] = fmaxf(a[r], 0.f);
            if (ft & 1) hB[qf][ft >> 1] = pack8<DT>(hprev[qf], a); else hprev[qf] = a;
        });
    }
    gemm_T_lp<DC, FKC, QF, DT>(p.w2_lp, p.b2, dff, hB, li, g, [&](int nt, int qf, f32x4 a) { x1[qf][nt] += a; });
#pragma unroll
    for (int qf = 0; qf < QF; ++qf) {
        layer_norm<DC>(x1[qf], p.ln2_w, p.ln2_b, p.d, p.ln_eps, g);
        const int qtok = q0 + qf * 16 + li;
        if (qtok < ge) {
#pragma unroll
            for (int nt = 0; nt < DC; ++nt) *reinterpret_cast<f32x4*>(p.out + (size_t)qtok * cs + 16 * nt + 4 * g) = x1[qf][nt];
        }
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
    k.dff_pad = d->dff_pad; k.pos_period = d->pos_period; k.ln_eps = d->ln_eps;
    k.qscale = 1.0f / sqrtf((float)d->d);
    k.w_in_lp = d->w_in_lp; k.w_out_lp = d->w_out_lp; k.w1_lp = d->w1_lp; k.w2_lp = d->w2_lp;
    k.next_w_in = d->next_w_in; k.next_b_in = d->next_b_in; k.next_kbuf = d->next_kbuf; k.next_vbuf = d->next_vbuf;
    I2R_CHECK_ARG(!d->next_w_in || (d->next_b_in && d->next_kbuf && d->next_vbuf && d->next_kbuf != d->kbuf && d->next_vbuf != d->vbuf && d->dtype == 0),
                  "i2r_encoder: fused next-layer K/V needs its own buffers (fp32 mode only)");
    I2R_CHECK_ARG(d->dtype >= 0 && d->dtype <= 2, "i2r_encoder: dtype %d", d->dtype);
    if (d->dtype != 0)
        I2R_CHECK_ARG(d->cs == 96 && d->w_in_lp && d->w_out_lp && d->w1_lp && d->w2_lp && d->n_qtiles16 > 0 && d->n_qtiles64 > 0,
                      "i2r_encoder: the 16-bit MFMA mode needs cs == 96 and the permuted 16-bit weights");
    return I2R_OK;
}

}  // namespace

extern "C" int i2r_encoder_kv(const i2r_encoder_desc* d, void* stream) {
    EncK k;
    int rc = fill(d, k);
    if (rc) return rc;
    const unsigned nblk = (unsigned)((d->n_tok + 31) / 32);
    if (d->dtype == 1)
        hipLaunchKernelGGL((enc_kv_lp_k<6, 1>), dim3(nblk), dim3(64), 0, (hipStream_t)stream, k);
    else if (d->dtype == 2)
        hipLaunchKernelGGL((enc_kv_lp_k<6, 2>), dim3(nblk), dim3(64), 0, (hipStream_t)stream, k);
    else if (d->cs == 96)
        hipLaunchKernelGGL(enc_kv_k<6>, dim3(nblk), dim3(64), 0, (hipStream_t)stream, k);
    else
        hipLaunchKernelGGL(enc_kv_k<5>, dim3(nblk), dim3(64), 0, (hipStream_t)stream, k);
    I2R_CHECK_LAUNCH("i2r_encoder_kv");
    return I2R_OK;
}

extern "C" int i2r_encoder_layer(const i2r_encoder_desc* d, void* stream) {
    EncK k;
    int rc = fill(d, k);
    if (rc) return rc;
    I2R_CHECK_ARG(d->n_qtiles32 > 0, "i2r_encoder_layer: n_qtiles32");
    if (d->dtype != 0) {
        // 64 queries per wave when that still gives >= 2 waves per SIMD-pair of the chip, else 16 (more, shorter waves)
        const bool big = d->n_qtiles64 >= 512;
        const unsigned grid = (unsigned)(big ? d->n_qtiles64 : d->n_qtiles16);
        if (d->dtype == 1) {
            if (big) hipLaunchKernelGGL((enc_layer_lp_k<6, 12, 4, 1>), dim3(grid), dim3(64), 0, (hipStream_t)stream, k);
            else hipLaunchKernelGGL((enc_layer_lp_k<6, 12, 1, 1>), dim3(grid), dim3(64), 0, (hipStream_t)stream, k);
        } else {
            if (big) hipLaunchKernelGGL((enc_layer_lp_k<6, 12, 4, 2>), dim3(grid), dim3(64), 0, (hipStream_t)stream, k);
            else hipLaunchKernelGGL((enc_layer_lp_k<6, 12, 1, 2>), dim3(grid), dim3(64), 0, (hipStream_t)stream, k);
        }
        I2R_CHECK_LAUNCH("i2r_encoder_layer");
        return I2R_OK;
    }
    static const int v2 = getenv("I2R_ENC_V2") ? atoi(getenv("I2R_ENC_V2")) : 0;  // tuning switch: the one-wave-per-fragment kernel
    if (v2 && !d->next_w_in) {
        constexpr int NW = 2;
        if (d->cs == 96)
            hipLaunchKernelGGL((enc_layer_k<6, 12, NW>), dim3((unsigned)d->n_qtiles32), dim3(NW * 64), 0, (hipStream_t)stream, k);
        else
            hipLaunchKernelGGL((enc_layer_k<5, 12, NW>), dim3((unsigned)d->n_qtiles32), dim3(NW * 64), 0, (hipStream_t)stream, k);
    } else {
        I2R_CHECK_ARG(d->n_qtiles16 > 0, "i2r_encoder_layer: n_qtiles16");
        if (d->cs == 96)
            hipLaunchKernelGGL((enc_layer4_k<6, 12>), dim3((unsigned)d->n_qtiles16), dim3(256), 0, (hipStream_t)stream, k);
        else
            hipLaunchKernelGGL((enc_layer4_k<5, 12>), dim3((unsigned)d->n_qtiles16), dim3(256), 0, (hipStream_t)stream, k);
    }
    I2R_CHECK_LAUNCH("i2r_encoder_layer");
    return I2R_OK;
}
