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
    return I2R_OK;
}

}  // namespace

extern "C" int i2r_encoder_kv(const i2r_encoder_desc* d, void* stream) {
    EncK k;
    int rc = fill(d, k);
    if (rc) return rc;
    const unsigned nblk = (unsigned)((d->n_tok + 31) / 32);
    if (d->cs == 96)
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
    constexpr int NW = 2;
    if (d->cs == 96)
        hipLaunchKernelGGL((enc_layer_k<6, 12, NW>), dim3((unsigned)d->n_qtiles32), dim3(NW * 64), 0, (hipStream_t)stream, k);
    else
        hipLaunchKernelGGL((enc_layer_k<5, 12, NW>), dim3((unsigned)d->n_qtiles32), dim3(NW * 64), 0, (hipStream_t)stream, k);
    I2R_CHECK_LAUNCH("i2r_encoder_layer");
    return I2R_OK;
}
