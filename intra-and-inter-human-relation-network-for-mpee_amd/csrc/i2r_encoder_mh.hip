// Multi-head softmax(q k^T) v over variable-length token groups: the attention core of nn.MultiheadAttention with MODEL.N_HEAD > 1
// (or of a pre-norm layer), for the DETR-style encoders of the reference (interformer_pureMulti.py:171-243, attention.py:37-112,
// transpose_h.py:168-240).  The fused single-head post-norm layer kernels (i2r_encoder.hip) cover every shipped yaml; this kernel is
// the general form behind the same config keys: the host composes a layer from 1x1 i2r_conv launches (q|k, v, out-proj, FFN),
// i2r_layernorm and this launch.
//
// One wave per (16-query tile of a group, head), flash-style over the group's keys in tiles of 16, everything on the fp32 matrix pipe
// (v_mfma_f32_16x16x4_f32) without LDS:
//   S^T = K Q^T     A = K rows (lane (li, g) holds key li, dims 16u + 4g + c as one 16-byte load), B = Q^T (same dims of query li):
//                   the contraction index of step (u, c) runs over g, i.e. over the four dims 16u + 4g + c -- any enumeration of the
//                   head's dims works as long as A and B agree, and this one makes both operands whole float4 loads;
//   D layout        lane (g, li) holds S^T[key 4g + r][query li], r < 4: the softmax statistics of query li live in lanes li + 16 g
//                   (two xor-shuffles), and p[r] IS the B operand of
//   O^T += V^T P^T  step r contracts over the keys 4g + r (A = V[key 4g + r][dim 16 db + li]); D = O^T[dim 16 db + 4g + r][query li],
//                   so the running rescale of query li never leaves the lane and the result is one float4 store per dim block.
// Head dims are padded to hp = a multiple of 16 by the host (zero weight rows: pad dims of q, k, v are exactly 0).
#include "i2r_common.h"

namespace {

struct MhK {
    const float* qk;   // [n_tok, qk_cs]: q of head h at channel h*hp, k at k_off + h*hp (q already scaled by head_dim^-0.5)
    const float* v;    // [n_tok, v_cs]
    float* out;        // [n_tok, out_cs]
    const int* grp_off;
    int n_grp, heads, hp, k_off, qk_cs, v_cs, out_cs;
};

constexpr float kLog2e = 1.4426950408889634f;

template <int HB>  // hp / 16
__global__ __launch_bounds__(256) void enc_mh_attn_k(const MhK p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int head = blockIdx.y * (blockDim.x >> 6) + wave;
    if (head >= p.heads) return;  // (no barrier below)
    int t = blockIdx.x, gi = 0, g0 = 0, g1 = 0;
    for (; gi < p.n_grp; ++gi) {
        g0 = p.grp_off[gi];
        g1 = p.grp_off[gi + 1];
        const int nt = (g1 - g0 + 15) >> 4;
        if (t < nt) break;
        t -= nt;
    }
    if (gi >= p.n_grp) return;
    const int q0 = g0 + t * 16;
    const int hc = head * p.hp + 4 * g;
    const int qrow = min(q0 + li, g1 - 1);
    f32x4 q[HB], o[HB];
#pragma unroll
    for (int u = 0; u < HB; ++u) {
        q[u] = *reinterpret_cast<const f32x4*>(p.qk + (size_t)qrow * p.qk_cs + hc + 16 * u) * kLog2e;  // (exp2 below)
        o[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    float m = -INFINITY, l = 0.f;
    for (int k0 = g0; k0 < g1; k0 += 16) {
        const int krow = min(k0 + li, g1 - 1);
        const float* kp = p.qk + (size_t)krow * p.qk_cs + p.k_off + hc;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < HB; ++u) {
            const f32x4 kk = *reinterpret_cast<const f32x4*>(kp + 16 * u);
#pragma unroll
            for (int c = 0; c < 4; ++c) s = mfma16(kk[c], q[u][c], s);
        }
        // the V operands of this key tile (rows clamped: their probabilities are zero)
        float vv[4][HB];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* vp = p.v + (size_t)min(k0 + 4 * g + r, g1 - 1) * p.v_cs + head * p.hp + li;
#pragma unroll
            for (int db = 0; db < HB; ++db) vv[r][db] = vp[16 * db];
        }
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (k0 + 4 * g + r >= g1) s[r] = -INFINITY;
            mx = fmaxf(mx, s[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m, mx);  // (finite: key k0 of every tile exists)
        const float alpha = __builtin_amdgcn_exp2f(m - m_new);
        float pr[4], ps = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pr[r] = __builtin_amdgcn_exp2f(s[r] - m_new);
            ps += pr[r];
        }
        l = l * alpha + ps;
        m = m_new;
#pragma unroll
        for (int db = 0; db < HB; ++db) o[db] *= alpha;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int db = 0; db < HB; ++db) o[db] = mfma16(vv[r][db], pr[r], o[db]);
    }
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const float inv = 1.f / l;
    if (q0 + li < g1) {
        float* op = p.out + (size_t)(q0 + li) * p.out_cs;
#pragma unroll
        for (int db = 0; db < HB; ++db) *reinterpret_cast<f32x4*>(op + hc + 16 * db) = o[db] * inv;
        // the columns behind the last head are the zero-weight pad columns of the out-proj: keep them finite
        if (head == 0)
            for (int c = p.heads * p.hp + 4 * g; c < p.out_cs; c += 16) *reinterpret_cast<f32x4*>(op + c) = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
}

}  // namespace

extern "C" int i2r_mh_attention(const i2r_mh_attn_args* a, void* stream) {
    I2R_CHECK_ARG(a && a->qk && a->v && a->out && a->grp_off, "i2r_mh_attention: null pointer");
    I2R_CHECK_ARG(a->heads > 0 && a->hp > 0 && a->hp % 16 == 0 && a->hp <= 256, "i2r_mh_attention: heads=%d hp=%d (hp: multiple of 16, <= 256)", a->heads, a->hp);
    const int hs = a->heads * a->hp;
    I2R_CHECK_ARG(a->k_off >= hs && a->k_off % 4 == 0 && a->qk_cs >= a->k_off + hs && a->v_cs >= hs && a->out_cs >= hs && a->qk_cs % 4 == 0 &&
                      a->v_cs % 4 == 0 && a->out_cs % 4 == 0,
                  "i2r_mh_attention: row strides qk=%d (k at %d) v=%d out=%d for %d heads x %d", a->qk_cs, a->k_off, a->v_cs, a->out_cs, a->heads, a->hp);
    I2R_CHECK_ARG(a->n_grp > 0 && a->n_qtiles16 > 0 && a->n_qtiles16 < (1 << 30), "i2r_mh_attention: n_grp=%d n_qtiles16=%d", a->n_grp, a->n_qtiles16);
    MhK k{a->qk, a->v, a->out, a->grp_off, a->n_grp, a->heads, a->hp, a->k_off, a->qk_cs, a->v_cs, a->out_cs};
    typedef void (*fn_t)(const MhK);
    static const fn_t fns[16] = {enc_mh_attn_k<1>,  enc_mh_attn_k<2>,  enc_mh_attn_k<3>,  enc_mh_attn_k<4>,  enc_mh_attn_k<5>,  enc_mh_attn_k<6>,
                                 enc_mh_attn_k<7>,  enc_mh_attn_k<8>,  enc_mh_attn_k<9>,  enc_mh_attn_k<10>, enc_mh_attn_k<11>, enc_mh_attn_k<12>,
                                 enc_mh_attn_k<13>, enc_mh_attn_k<14>, enc_mh_attn_k<15>, enc_mh_attn_k<16>};
    const int wpb = a->heads >= 4 ? 4 : a->heads;  // waves (= heads) per workgroup
    hipLaunchKernelGGL(fns[a->hp / 16 - 1], dim3((unsigned)a->n_qtiles16, (unsigned)((a->heads + wpb - 1) / wpb)), dim3(64 * wpb), 0, (hipStream_t)stream, k);
    I2R_CHECK_LAUNCH("i2r_mh_attention");
    return I2R_OK;
}
