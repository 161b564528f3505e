// Multi-head softmax(q k^T) v over variable-length token groups: the attention core of nn.MultiheadAttention with MODEL.N_HEAD > 1
// (or of a pre-norm layer), for the DETR-style encoders of the reference (interformer_pureMulti.py:171-243, attention.py:37-112,
// transpose_h.py:168-240).  The fused single-head post-norm layer kernels (i2r_encoder.hip) cover every shipped yaml; this kernel is
// the general form behind the same config keys: the host composes a layer from 1x1 i2r_conv launches (q|k, v, out-proj, FFN),
// i2r_layernorm and this launch.
//
// One wave per (tile of 16 NQ queries of a group, head), flash-style over the group's keys in tiles of 16 (the next tile's operands
// fetched under the current one's arithmetic), everything on the fp32 matrix pipe
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
    const int* key_len;  // optional: group g attends to its first key_len[g] rows only (the others are queries without being keys)
    int n_grp, heads, hp, k_off, qk_cs, v_cs, out_cs;
};

constexpr float kLog2e = 1.4426950408889634f;

template <int HB, int NQ>  // hp / 16, 16-query tiles per wave (they share every K / V fragment the wave loads)
__global__ __launch_bounds__(256) void enc_mh_attn_k(const MhK p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int head = blockIdx.y * (blockDim.x >> 6) + wave;
    if (head >= p.heads) return;  // (no barrier below)
    int t = blockIdx.x, gi = 0, g0 = 0, g1 = 0;
    for (; gi < p.n_grp; ++gi) {
        g0 = p.grp_off[gi];
        g1 = p.grp_off[gi + 1];
        const int nt = (g1 - g0 + 16 * NQ - 1) / (16 * NQ);
        if (t < nt) break;
        t -= nt;
    }
    if (gi >= p.n_grp) return;
    const int q0 = g0 + t * 16 * NQ;
    const int q1 = g1;                           // queries: the whole group
    if (p.key_len) g1 = g0 + min(max(p.key_len[gi], 1), g1 - g0);  // keys: its first key_len rows (key_padding_mask of the padded persons),
                                                                  // clamped to [1, group length]: no row of another group is ever read
    const int hc = head * p.hp + 4 * g;
    f32x4 q[NQ][HB], o[NQ][HB];
    float m[NQ], l[NQ];
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
        const int qrow = min(q0 + 16 * n + li, q1 - 1);
#pragma unroll
        for (int u = 0; u < HB; ++u) {
            q[n][u] = *reinterpret_cast<const f32x4*>(p.qk + (size_t)qrow * p.qk_cs + hc + 16 * u) * kLog2e;  // (exp2 below)
            o[n][u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        m[n] = -INFINITY;
        l[n] = 0.f;
    }
    // the operands of one key tile: K rows as float4 (lane: key li, dims 16u + 4g ..), V as the A operand of the PV steps (lane: dim li of
    // block db, key 4g + r); rows clamped to the group (their probabilities are zero)
    auto load_tile = [&](int k0, f32x4(&kk)[HB], float(&vv)[4][HB]) {
        const float* kp = p.qk + (size_t)min(k0 + li, g1 - 1) * p.qk_cs + p.k_off + hc;
#pragma unroll
        for (int u = 0; u < HB; ++u) kk[u] = *reinterpret_cast<const f32x4*>(kp + 16 * u);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* vp = p.v + (size_t)min(k0 + 4 * g + r, g1 - 1) * p.v_cs + head * p.hp + li;
#pragma unroll
            for (int db = 0; db < HB; ++db) vv[r][db] = vp[16 * db];
        }
    };
    f32x4 kk[HB], kn[HB];
    float vv[4][HB], vn[4][HB];
    load_tile(g0, kk, vv);
    for (int k0 = g0; k0 < g1; k0 += 16) {
        load_tile(min(k0 + 16, g1 - 1), kn, vn);  // (the next tile flies under this one's arithmetic; past the end: a harmless re-read)
#pragma unroll
        for (int n = 0; n < NQ; ++n) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < HB; ++u)
#pragma unroll
                for (int c = 0; c < 4; ++c) s = mfma16(kk[u][c], q[n][u][c], s);
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (k0 + 4 * g + r >= g1) s[r] = -INFINITY;
                mx = fmaxf(mx, s[r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m[n], mx);  // (finite: key k0 of every tile exists)
            const float alpha = __builtin_amdgcn_exp2f(m[n] - m_new);
            float pr[4], ps = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pr[r] = __builtin_amdgcn_exp2f(s[r] - m_new);
                ps += pr[r];
            }
            l[n] = l[n] * alpha + ps;
            m[n] = m_new;
#pragma unroll
            for (int db = 0; db < HB; ++db) o[n][db] *= alpha;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int db = 0; db < HB; ++db) o[n][db] = mfma16(vv[r][db], pr[r], o[n][db]);
        }
#pragma unroll
        for (int u = 0; u < HB; ++u) kk[u] = kn[u];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int db = 0; db < HB; ++db) vv[r][db] = vn[r][db];
    }
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
        float ls = l[n];
        ls += __shfl_xor(ls, 16);
        ls += __shfl_xor(ls, 32);
        const float inv = 1.f / ls;
        if (q0 + 16 * n + li < q1) {
            float* op = p.out + (size_t)(q0 + 16 * n + li) * p.out_cs;
#pragma unroll
            for (int db = 0; db < HB; ++db) *reinterpret_cast<f32x4*>(op + hc + 16 * db) = o[n][db] * inv;
            // the columns behind the last head are the zero-weight pad columns of the out-proj: keep them finite
            if (head == 0)
                for (int c = p.heads * p.hp + 4 * g; c < p.out_cs; c += 16) *reinterpret_cast<f32x4*>(op + c) = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
}

// crop i of out = crop map[i] of src, or zeros (map[i] < 0): the reference's padding_tensor (interformer.py:230-249) for the consumers that
// need the padded persons as ROWS (ATTENTION_TYPE window: their tokens are queries whose outputs the final view() redistributes)
__global__ __launch_bounds__(256) void rows_gather_k(const f32x4* __restrict__ src, f32x4* __restrict__ out, const int* __restrict__ map, int n_out, int per_crop4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)n_out * per_crop4) return;
    const int crop = (int)(i / per_crop4), r = (int)(i - (long long)crop * per_crop4);
    const int m = map[crop];
    out[i] = m >= 0 ? src[(size_t)m * per_crop4 + r] : (f32x4){0.f, 0.f, 0.f, 0.f};
}

// GeneralTransformerBlock.forward of attention.py (:1025-1029): the attention output [L, B, C] (L = P H W tokens of an image, B images) goes
// through permute(0, 2, 1).contiguous().view(B, C, P, H, W) -- a REINTERPRETATION of the [L, C, B] memory, not a transpose -- and
// permute(0, 2, 1, 3, 4).view(B P, C, H, W); get_valid_output then keeps the real persons.  Element (b', p', c', y, x) of the result is flat
// element f = (((b' C + c') P + p') H + y) W + x of [L, C, B], i.e. o[l = f / (C B)][b = f % B][c = (f / B) % C].  Restated as is.
// o: [B][L][cs] rows (image-major, as the attention launch writes them); out: NHWC crops of the real persons, crop s = (b', p') = pm[s].
__global__ __launch_bounds__(256) void view_scramble_k(const float* __restrict__ o, float* __restrict__ out, const int* __restrict__ pm, int n_out, int B, int P,
                                                        int C, int cs, int HW) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)n_out * HW * cs) return;
    const int c = (int)(i % cs);
    const long long t = i / cs;
    const int yx = (int)(t % HW), s = (int)(t / HW);
    float v = 0.f;
    if (c < C) {
        const int bp = pm[s], b1 = bp / P, p1 = bp - b1 * P;
        const long long f = (((long long)b1 * C + c) * P + p1) * HW + yx;
        const int b = (int)(f % B);
        const long long r = f / B;
        const int cc = (int)(r % C);
        const long long l = r / C;
        v = o[((long long)b * P * HW + l) * cs + cc];
    }
    out[i] = v;
}

}  // namespace

extern "C" int i2r_rows_gather(const float* src, float* out, const int32_t* map, int32_t n_out, int32_t floats_per_crop, void* stream) {
    I2R_CHECK_ARG(src && out && map && src != out, "i2r_rows_gather: bad pointers");
    I2R_CHECK_ARG(n_out > 0 && floats_per_crop > 0 && floats_per_crop % 4 == 0, "i2r_rows_gather: n_out=%d floats_per_crop=%d", n_out, floats_per_crop);
    const long long n4 = (long long)n_out * (floats_per_crop / 4);
    I2R_CHECK_ARG((n4 + 255) / 256 < (1ll << 31), "i2r_rows_gather: grid");
    i2r_launch(rows_gather_k, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const f32x4*>(src),
                       reinterpret_cast<f32x4*>(out), map, n_out, floats_per_crop / 4);
    I2R_CHECK_LAUNCH("i2r_rows_gather");
    return I2R_OK;
}

extern "C" int i2r_view_scramble(const float* o, float* out, const int32_t* person_map, int32_t n_out, int32_t n_images, int32_t max_persons, int32_t c,
                                 int32_t cs, int32_t hw, void* stream) {
    I2R_CHECK_ARG(o && out && person_map && o != out, "i2r_view_scramble: bad pointers");
    I2R_CHECK_ARG(n_out > 0 && n_images > 0 && max_persons > 0 && c > 0 && c <= cs && hw > 0, "i2r_view_scramble: sizes");
    const long long n = (long long)n_out * hw * cs;
    I2R_CHECK_ARG((n + 255) / 256 < (1ll << 31) && (long long)n_images * max_persons * hw * cs < (1ll << 40), "i2r_view_scramble: grid");
    i2r_launch(view_scramble_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, o, out, person_map, n_out, n_images, max_persons, c, cs, hw);
    I2R_CHECK_LAUNCH("i2r_view_scramble");
    return I2R_OK;
}

extern "C" int i2r_mh_attention(const i2r_mh_attn_args* a, void* stream) {
    I2R_CHECK_ARG(a && a->qk && a->v && a->out && a->grp_off, "i2r_mh_attention: null pointer");
    I2R_CHECK_ARG(a->heads > 0 && a->hp > 0 && a->hp % 16 == 0 && a->hp <= 256, "i2r_mh_attention: heads=%d hp=%d (hp: multiple of 16, <= 256)", a->heads, a->hp);
    const int hs = a->heads * a->hp;
    I2R_CHECK_ARG(a->k_off >= hs && a->k_off % 4 == 0 && a->qk_cs >= a->k_off + hs && a->v_cs >= hs && a->out_cs >= hs && a->qk_cs % 4 == 0 &&
                      a->v_cs % 4 == 0 && a->out_cs % 4 == 0,
                  "i2r_mh_attention: row strides qk=%d (k at %d) v=%d out=%d for %d heads x %d", a->qk_cs, a->k_off, a->v_cs, a->out_cs, a->heads, a->hp);
    I2R_CHECK_ARG(a->n_grp > 0 && a->n_qtiles64 > 0 && a->n_qtiles64 <= a->n_qtiles32 && a->n_qtiles32 <= a->n_qtiles16 && a->n_qtiles16 < (1 << 30),
                  "i2r_mh_attention: n_grp=%d n_qtiles16/32/64=%d/%d/%d", a->n_grp, a->n_qtiles16, a->n_qtiles32, a->n_qtiles64);
    MhK k{a->qk, a->v, a->out, a->grp_off, a->key_len, a->n_grp, a->heads, a->hp, a->k_off, a->qk_cs, a->v_cs, a->out_cs};
    typedef void (*fn_t)(const MhK);
    // query tiles per wave (they share the K / V fragments a wave loads; registers: q and o are NQ x HB fragments): up to 64 queries for
    // heads of <= 32 dims, 32 up to 96 dims, 16 beyond -- but never so few waves that the chip's 1024 SIMDs go unfilled (an inter-human
    // stack over a few hundred tokens per image keeps 16-query tiles)
    static const fn_t fn1[16] = {enc_mh_attn_k<1, 1>,  enc_mh_attn_k<2, 1>,  enc_mh_attn_k<3, 1>,  enc_mh_attn_k<4, 1>,  enc_mh_attn_k<5, 1>,  enc_mh_attn_k<6, 1>,
                                 enc_mh_attn_k<7, 1>,  enc_mh_attn_k<8, 1>,  enc_mh_attn_k<9, 1>,  enc_mh_attn_k<10, 1>, enc_mh_attn_k<11, 1>, enc_mh_attn_k<12, 1>,
                                 enc_mh_attn_k<13, 1>, enc_mh_attn_k<14, 1>, enc_mh_attn_k<15, 1>, enc_mh_attn_k<16, 1>};
    static const fn_t fn2[6] = {enc_mh_attn_k<1, 2>, enc_mh_attn_k<2, 2>, enc_mh_attn_k<3, 2>, enc_mh_attn_k<4, 2>, enc_mh_attn_k<5, 2>, enc_mh_attn_k<6, 2>};
    static const fn_t fn4[2] = {enc_mh_attn_k<1, 4>, enc_mh_attn_k<2, 4>};
    const int hb = a->hp / 16;
    constexpr long long kMinWaves = 4096;
    fn_t fn = fn1[hb - 1];
    int n_tiles = a->n_qtiles16;
    if (hb <= 2 && (long long)a->n_qtiles64 * a->heads >= kMinWaves) {
        fn = fn4[hb - 1];
        n_tiles = a->n_qtiles64;
    } else if (hb <= 6 && (long long)a->n_qtiles32 * a->heads >= kMinWaves) {
        fn = fn2[hb - 1];
        n_tiles = a->n_qtiles32;
    }
    const int wpb = a->heads >= 4 ? 4 : a->heads;  // waves (= heads) per workgroup
    i2r_launch(fn, dim3((unsigned)n_tiles, (unsigned)((a->heads + wpb - 1) / wpb)), dim3(64 * wpb), 0, (hipStream_t)stream, k);
    I2R_CHECK_LAUNCH("i2r_mh_attention");
    return I2R_OK;
}
