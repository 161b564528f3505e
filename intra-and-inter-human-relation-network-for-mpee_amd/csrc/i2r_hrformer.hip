// HRFormer-B building blocks that are not convolutions: channel LayerNorm, 7x7-window multi-head attention,
// depth-wise 3x3 conv (+folded BN, +GELU/ReLU) and bilinear-upsample-accumulate of the multi-scale fuse.
// All are HBM / LDS-bound glue around the MFMA conv kernel (window attention is 5 % of the HRFormer FLOPs).
#include "i2r_conv.h"  // (ld_act4 / st_act4: fp32 or bf16 / f16 activation storage)

namespace {

__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

__device__ __forceinline__ f32x4 act4(f32x4 v, int act) {
    if (act == 1) {
        v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
    } else if (act == 2) {
        v[0] = gelu_exact(v[0]); v[1] = gelu_exact(v[1]); v[2] = gelu_exact(v[2]); v[3] = gelu_exact(v[3]);
    }
    return v;
}

// ---- LayerNorm over the c real channels of each pixel; 16 lanes per pixel, values cached in registers ----
constexpr int kLnMaxChunks = 10;  // 16 lanes x 10 float4 = 640 channels
template <int ODT>  // storage type of the output: 0 fp32, 1 bf16, 2 f16 (16-bit modes: it only feeds a 16-bit conv)
__global__ __launch_bounds__(256) void layernorm_k(const float* __restrict__ in, const float* __restrict__ w,
                                                   const float* __restrict__ b, float* __restrict__ out, int npix, int c,
                                                   int cs, float eps) {
    const int l16 = threadIdx.x & 15;
    const long long pix = ((long long)blockIdx.x * 256 + threadIdx.x) >> 4;
    const bool ok = pix < npix;
    const int nch = cs >> 2;
    const float* row = in + (size_t)(ok ? pix : 0) * cs;
    f32x4 v[kLnMaxChunks];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < kLnMaxChunks; ++k) {
        const int ch = l16 + 16 * k;
        v[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (ch < nch) v[k] = *reinterpret_cast<const f32x4*>(row + ch * 4);
        s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);  // pad channels are exact zeros
    }
    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);
    const float mean = s / (float)c;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < kLnMaxChunks; ++k) {
        const int ch = l16 + 16 * k;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float t = (ch * 4 + e < c) ? v[k][e] - mean : 0.f;
            q += t * t;
        }
    }
    q += __shfl_xor(q, 1); q += __shfl_xor(q, 2); q += __shfl_xor(q, 4); q += __shfl_xor(q, 8);
    const float rstd = rsqrtf(q / (float)c + eps);
    if (!ok) return;
#pragma unroll
    for (int k = 0; k < kLnMaxChunks; ++k) {
        const int ch = l16 + 16 * k;
        if (ch < nch) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(w + ch * 4), bv = *reinterpret_cast<const f32x4*>(b + ch * 4);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[k][e] - mean) * rstd * wv[e] + bv[e];  // padded w = b = 0 -> 0
            st_act4<ODT>(out, (size_t)pix * cs + ch * 4, o, ODT != 0);
        }
    }
}

// ---- 7x7 window attention on the fp32 matrix pipe: one workgroup (4 waves) per (crop, window, head) ----
// qkv: [n, h, w, 3*hs] (q | k | v, each hs = heads*HP wide; head hh owns channels hh*HP .. hh*HP+hd-1, HP = head_dim padded
// to a multiple of 4 with exactly-zero pad channels; q already carries the hd^-1/2 scale: the host folds it into q_proj).
// Tokens of the zero-padded window border are not stored: their projections equal the bias vector (the LayerNorm output is
// padded with zeros BEFORE q/k/v_proj, hrformer.py:947-956).
// Transposed formulation as in the encoder kernel: S^T[key][query] = K Q^T and O^T[dim][query] = V^T P^T, so the softmax
// output fragment (D layout) is the B operand of the second product without leaving the registers.  The 49 tokens are padded
// to 4 fragments of 16 (keys >= 49 masked to -inf; V^T columns >= 49 zero), head_dim 39 -> 3 k-blocks of 16 (dims >= 40 zero).
// Q / K rows and V^T go through LDS once per workgroup (coalesced 16-byte gathers; row strides 44 / 68 floats keep the
// ds_read_b128 operand fetches bank-conflict free); wave w owns query fragment w.
template <int HP>
__global__ __launch_bounds__(256) void window_attn_k(const float* __restrict__ qkv, const float* __restrict__ bias,
                                                     float* __restrict__ out, int n_img, int h, int w, int hs, int heads,
                                                     int hd, int nwy, int nwx, int pad_top, int pad_left) {
    static_assert(HP == 40, "3 k-blocks of 16 with the upper half of the last one zero");
    constexpr int QS = 44, VS = 68, J4 = HP / 4;
    __shared__ __attribute__((aligned(16))) float Qs[64 * QS];
    __shared__ __attribute__((aligned(16))) float Ks[64 * QS];
    __shared__ __attribute__((aligned(16))) float Vt[48 * VS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
    int bid = blockIdx.x;
    const int hh = bid % heads; bid /= heads;
    const int wx = bid % nwx; bid /= nwx;
    const int wy = bid % nwy;
    const int img = bid / nwy;
    const f32x4 z4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    // rows 49..63 of Q / K and the key columns 49..63 of V^T: zero (Q, V^T) or anything finite (K: masked)
    for (int i = tid; i < 15 * QS / 4; i += 256) {
        reinterpret_cast<f32x4*>(Qs + 49 * QS)[i] = z4;
        reinterpret_cast<f32x4*>(Ks + 49 * QS)[i] = z4;
    }
    for (int i = tid; i < 48 * 5; i += 256) {  // V^T[dim][48..67]: the column of token 48 is rewritten below
        const int d = i / 5, c4 = i - d * 5;
        *reinterpret_cast<f32x4*>(Vt + d * VS + 48 + c4 * 4) = z4;
    }
    for (int i = tid; i < 8 * VS / 4; i += 256) reinterpret_cast<f32x4*>(Vt + 40 * VS)[i] = z4;  // dims 40..47
    __syncthreads();
    // ---- gather: 49 tokens x (q, k, v) x HP/4 pieces of 16 bytes ----
    for (int e = tid; e < 3 * 49 * J4; e += 256) {
        const int part = e / (49 * J4), r = e - part * 49 * J4;
        const int t = r / J4, j4 = r - t * J4;
        const int ty = t / 7, tx = t - ty * 7;
        const int y = wy * 7 + ty - pad_top, x = wx * 7 + tx - pad_left;
        const bool inside = y >= 0 && y < h && x >= 0 && x < w;
        const float* src = inside ? qkv + (((size_t)img * h + y) * w + x) * 3 * hs : bias;
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + part * hs + hh * HP + j4 * 4);
        if (part == 0) *reinterpret_cast<f32x4*>(Qs + t * QS + j4 * 4) = v;
        else if (part == 1) *reinterpret_cast<f32x4*>(Ks + t * QS + j4 * 4) = v;
        else {
#pragma unroll
            for (int i = 0; i < 4; ++i) Vt[(j4 * 4 + i) * VS + t] = v[i];
        }
    }
    __syncthreads();
    // ---- S^T = K Q^T for this wave's 16 queries ----
    f32x4 bq[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) bq[c] = (c < 2 || g < 2) ? *reinterpret_cast<const f32x4*>(Qs + (wave * 16 + li) * QS + 16 * c + 4 * g) : z4;
    f32x4 st[4];
#pragma unroll
    for (int kf = 0; kf < 4; ++kf) {
        st[kf] = z4;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const f32x4 ka = (c < 2 || g < 2) ? *reinterpret_cast<const f32x4*>(Ks + (kf * 16 + li) * QS + 16 * c + 4 * g) : z4;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) st[kf] = mfma16(ka[s4], bq[c][s4], st[kf]);
        }
    }
    // keys 49..63 do not exist (row 4g + r of fragment 3)
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (4 * g + r >= 1) st[3][r] = -__builtin_inff();
    float mx = -__builtin_inff();
#pragma unroll
    for (int kf = 0; kf < 4; ++kf) mx = fmaxf(mx, fmaxf(fmaxf(st[kf][0], st[kf][1]), fmaxf(st[kf][2], st[kf][3])));
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int kf = 0; kf < 4; ++kf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            st[kf][r] = __expf(st[kf][r] - mx);
            sum += st[kf][r];
        }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float inv = 1.f / sum;
    // ---- O^T = V^T P^T ----
    const int t = wave * 16 + li;  // this lane's query token
    const int ty = t / 7, tx = t - ty * 7;
    const int y = wy * 7 + ty - pad_top, x = wx * 7 + tx - pad_left;
    const bool store = t < 49 && y >= 0 && y < h && x >= 0 && x < w;
    float* orow = out + (((size_t)img * h + (store ? y : 0)) * w + (store ? x : 0)) * hs + hh * HP;
#pragma unroll
    for (int df = 0; df < 3; ++df) {
        f32x4 o = z4;
#pragma unroll
        for (int kf = 0; kf < 4; ++kf) {
            const f32x4 va = *reinterpret_cast<const f32x4*>(Vt + (df * 16 + li) * VS + kf * 16 + 4 * g);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) o = mfma16(va[s4], st[kf][s4], o);
        }
        if (store && 16 * df + 4 * g < HP) *reinterpret_cast<f32x4*>(orow + 16 * df + 4 * g) = o * inv;
    }
}

// ---- depth-wise 3x3 conv, pad 1, stride 1|2, + bias (BN folded) + activation; thread = pixel x 4 channels ----
__global__ __launch_bounds__(256) void dwconv3x3_k(const float* __restrict__ in, const float* __restrict__ w,
                                                   const float* __restrict__ bias, float* __restrict__ out, int n_img, int in_h,
                                                   int in_w, int out_h, int out_w, int c4, int cs, int stride, int act) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const int cg = (int)(gid % c4);
    const long long pix = gid / c4;
    if (pix >= (long long)n_img * out_h * out_w) return;
    const int ox = (int)(pix % out_w);
    const int oy = (int)((pix / out_w) % out_h);
    const int img = (int)(pix / ((long long)out_w * out_h));
    f32x4 acc = *reinterpret_cast<const f32x4*>(bias + cg * 4);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * stride - 1 + ky;
        if (iy < 0 || iy >= in_h) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox * stride - 1 + kx;
            if (ix < 0 || ix >= in_w) continue;
            const f32x4 v = *reinterpret_cast<const f32x4*>(in + ((size_t)(img * in_h + iy) * in_w + ix) * cs + cg * 4);
            const f32x4 wv = *reinterpret_cast<const f32x4*>(w + (ky * 3 + kx) * cs + cg * 4);
            acc[0] = fmaf(v[0], wv[0], acc[0]); acc[1] = fmaf(v[1], wv[1], acc[1]);
            acc[2] = fmaf(v[2], wv[2], acc[2]); acc[3] = fmaf(v[3], wv[3], acc[3]);
        }
    }
    *reinterpret_cast<f32x4*>(out + (size_t)pix * cs + cg * 4) = act4(acc, act);
}

// stride-1 variant: one thread = 4 vertically adjacent output pixels x 4 channels -- 18 input loads and 9 weight loads
// for 4 outputs instead of 36 + 36 (the kernel is L1 / addresser bound, not HBM bound, with one output per thread)
template <int DT>  // activation storage of in AND out: 0 fp32, 1 bf16, 2 f16 (the MLP hidden tensor of the 16-bit modes)
__global__ __launch_bounds__(256) void dwconv3x3_s1_k(const float* __restrict__ in, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ out, int n_img, int h,
                                                      int wd, int c4, int cs, int act) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const int cg = (int)(gid % c4);
    const long long col = gid / c4;
    const int hs = (h + 3) >> 2;  // strips per column
    if (col >= (long long)n_img * hs * wd) return;
    const int ox = (int)(col % wd);
    const int oy0 = (int)((col / wd) % hs) * 4;
    const int img = (int)(col / ((long long)wd * hs));
    f32x4 wv[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wv[t] = *reinterpret_cast<const f32x4*>(w + t * cs + cg * 4);
    const f32x4 b = *reinterpret_cast<const f32x4*>(bias + cg * 4);
    f32x4 acc[4] = {b, b, b, b};
#pragma unroll
    for (int r = 0; r < 6; ++r) {  // input rows oy0-1 .. oy0+4
        const int iy = oy0 - 1 + r;
        if (iy < 0 || iy >= h) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox - 1 + kx;
            if (ix < 0 || ix >= wd) continue;
            const f32x4 v = ld_act4<DT>(in, ((size_t)(img * h + iy) * wd + ix) * cs + cg * 4, DT != 0);
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const int ky = r - o;  // output row oy0 + o takes input row oy0 + o - 1 + ky
                if (ky >= 0 && ky < 3) {
                    const f32x4 ww = wv[ky * 3 + kx];
                    acc[o][0] = fmaf(v[0], ww[0], acc[o][0]); acc[o][1] = fmaf(v[1], ww[1], acc[o][1]);
                    acc[o][2] = fmaf(v[2], ww[2], acc[o][2]); acc[o][3] = fmaf(v[3], ww[3], acc[o][3]);
                }
            }
        }
    }
#pragma unroll
    for (int o = 0; o < 4; ++o)
        if (oy0 + o < h) st_act4<DT>(out, ((size_t)(img * h + oy0 + o) * wd + ox) * cs + cg * 4, act4(acc[o], act), DT != 0);
}

// ---- out = act(((res + up(low_0)) + up(low_1)) + up(low_2)), up = bilinear up-sampling (align_corners = False) by an integer scale;
//      NT terms in ONE pass, summed in that order: bit-identical to NT passes that hand the fp32 sum through memory ----
struct UpK {
    const float* low[3]; const float* res; float* out;
    int n_img, H, W, scale[3], c4, cs, act;
};
template <int NT>
__global__ __launch_bounds__(256) void upsample_add_k(const UpK p) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const int cg = (int)(gid % p.c4);
    const long long pix = gid / p.c4;
    if (pix >= (long long)p.n_img * p.H * p.W) return;
    const int ox = (int)(pix % p.W);
    const int oy = (int)((pix / p.W) % p.H);
    const int img = (int)(pix / ((long long)p.W * p.H));
    f32x4 r = *reinterpret_cast<const f32x4*>(p.res + (size_t)pix * p.cs + cg * 4);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int scale = p.scale[t], lh = p.H / scale, lw = p.W / scale;
        // PyTorch upsample_bilinear2d, align_corners=False: src = max((dst + 0.5) / scale - 0.5, 0)
        const float rs = 1.f / (float)scale;
        const float sy = fmaxf(((float)oy + 0.5f) * rs - 0.5f, 0.f), sx = fmaxf(((float)ox + 0.5f) * rs - 0.5f, 0.f);
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = y0 + (y0 < lh - 1 ? 1 : 0), x1 = x0 + (x0 < lw - 1 ? 1 : 0);
        const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
        const float* base = p.low[t] + (size_t)img * lh * lw * p.cs + cg * 4;
        const f32x4 v00 = *reinterpret_cast<const f32x4*>(base + ((size_t)y0 * lw + x0) * p.cs);
        const f32x4 v01 = *reinterpret_cast<const f32x4*>(base + ((size_t)y0 * lw + x1) * p.cs);
        const f32x4 v10 = *reinterpret_cast<const f32x4*>(base + ((size_t)y1 * lw + x0) * p.cs);
        const f32x4 v11 = *reinterpret_cast<const f32x4*>(base + ((size_t)y1 * lw + x1) * p.cs);
        {
#pragma clang fp contract(off)  // (the same roundings whatever NT: a term is formed on its own, then added)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float term = hy * (hx * v00[e] + lx * v01[e]) + ly * (hx * v10[e] + lx * v11[e]);
                r[e] = r[e] + term;
            }
        }
    }
    *reinterpret_cast<f32x4*>(p.out + (size_t)pix * p.cs + cg * 4) = act4(r, p.act);
}

}  // namespace

extern "C" int i2r_layernorm(const float* in, const float* w, const float* b, float* out, int32_t npix, int32_t c, int32_t cs,
                             float eps, int32_t out_dt, void* stream) {
    I2R_CHECK_ARG(in && w && b && out, "i2r_layernorm: null pointer");
    I2R_CHECK_ARG(out_dt >= 0 && out_dt <= 2, "i2r_layernorm: out_dt %d", out_dt);
    I2R_CHECK_ARG(c > 0 && c <= cs && cs % 4 == 0 && cs <= 16 * 4 * kLnMaxChunks, "i2r_layernorm: c=%d cs=%d", c, cs);
    const long long nthr = (long long)npix * 16;
    typedef void (*ln_fn)(const float*, const float*, const float*, float*, int, int, int, float);
    static const ln_fn fns[3] = {layernorm_k<0>, layernorm_k<1>, layernorm_k<2>};
    i2r_launch(fns[out_dt], dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, w, b, out, npix, c, cs, eps);
    I2R_CHECK_LAUNCH("i2r_layernorm");
    return I2R_OK;
}

extern "C" int i2r_window_attn(const float* qkv, const float* bias_qkv, float* out, int32_t n_img, int32_t h, int32_t w, int32_t c,
                               int32_t hs, int32_t heads, void* stream) {
    I2R_CHECK_ARG(qkv && bias_qkv && out, "i2r_window_attn: null pointer");
    I2R_CHECK_ARG(heads > 0 && c % heads == 0, "i2r_window_attn: c=%d heads=%d", c, heads);
    const int hd = c / heads;
    I2R_CHECK_ARG(hd > 36 && hd <= 40 && hs == heads * 40, "i2r_window_attn: head_dim %d / part width %d (need 37..40 and heads*40)", hd, hs);
    const int nwy = (h + 6) / 7, nwx = (w + 6) / 7;
    const int pad_top = (nwy * 7 - h) / 2, pad_left = (nwx * 7 - w) / 2;
    const long long nblk = (long long)n_img * nwy * nwx * heads;
    I2R_CHECK_ARG(nblk < (1ll << 31), "i2r_window_attn: grid");
    i2r_launch(window_attn_k<40>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, qkv, bias_qkv, out, n_img, h, w,
                       hs, heads, hd, nwy, nwx, pad_top, pad_left);
    I2R_CHECK_LAUNCH("i2r_window_attn");
    return I2R_OK;
}

extern "C" int i2r_dwconv3x3(const float* in, const float* w, const float* bias, float* out, int32_t n_img, int32_t in_h,
                             int32_t in_w, int32_t c, int32_t cs, int32_t stride, int32_t act, int32_t dt, void* stream) {
    I2R_CHECK_ARG(in && w && bias && out && in != out, "i2r_dwconv3x3: bad pointers");
    I2R_CHECK_ARG(dt >= 0 && dt <= 2 && (dt == 0 || stride == 1), "i2r_dwconv3x3: 16-bit storage (dt=%d) is built for the stride-1 MLP conv", dt);
    I2R_CHECK_ARG(c > 0 && c <= cs && cs % 4 == 0 && (stride == 1 || stride == 2) && act >= 0 && act <= 2, "i2r_dwconv3x3: args");
    const int out_h = (in_h - 1) / stride + 1, out_w = (in_w - 1) / stride + 1;
    if (stride == 1) {
        const long long nthr = (long long)n_img * ((in_h + 3) / 4) * in_w * (cs / 4);
        typedef void (*dw_fn)(const float*, const float*, const float*, float*, int, int, int, int, int, int);
        static const dw_fn fns[3] = {dwconv3x3_s1_k<0>, dwconv3x3_s1_k<1>, dwconv3x3_s1_k<2>};
        i2r_launch(fns[dt], dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, w, bias, out, n_img, in_h, in_w,
                           cs / 4, cs, act);
    } else {
        const long long nthr = (long long)n_img * out_h * out_w * (cs / 4);
        i2r_launch(dwconv3x3_k, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, w, bias, out, n_img,
                           in_h, in_w, out_h, out_w, cs / 4, cs, stride, act);
    }
    I2R_CHECK_LAUNCH("i2r_dwconv3x3");
    return I2R_OK;
}

extern "C" int i2r_upsample_bilinear_add_multi(const i2r_up_args* a, void* stream) {
    I2R_CHECK_ARG(a && a->low && a->res && a->out, "i2r_upsample_bilinear_add: null pointer");
    const int nt = 1 + (a->low2 != nullptr) + (a->low2 && a->low3);
    I2R_CHECK_ARG(a->scale >= 1 && a->c <= a->cs && a->cs % 4 == 0 && a->act >= 0 && a->act <= 2 && (a->low2 || !a->low3), "i2r_upsample_bilinear_add: args");
    UpK k;
    k.low[0] = a->low; k.low[1] = a->low2; k.low[2] = a->low3; k.res = a->res; k.out = a->out;
    k.n_img = a->n_img; k.H = a->low_h * a->scale; k.W = a->low_w * a->scale; k.scale[0] = a->scale; k.scale[1] = a->scale2; k.scale[2] = a->scale3;
    k.c4 = a->cs / 4; k.cs = a->cs; k.act = a->act;
    for (int t = 1; t < nt; ++t)
        I2R_CHECK_ARG(k.scale[t] >= 1 && k.H % k.scale[t] == 0 && k.W % k.scale[t] == 0, "i2r_upsample_bilinear_add: term %d: scale %d does not divide %dx%d", t, k.scale[t], k.H, k.W);
    const long long nthr = (long long)k.n_img * k.H * k.W * k.c4;
    const dim3 grid((unsigned)((nthr + 255) / 256));
    if (nt == 1) i2r_launch(upsample_add_k<1>, grid, dim3(256), 0, (hipStream_t)stream, k);
    else if (nt == 2) i2r_launch(upsample_add_k<2>, grid, dim3(256), 0, (hipStream_t)stream, k);
    else i2r_launch(upsample_add_k<3>, grid, dim3(256), 0, (hipStream_t)stream, k);
    I2R_CHECK_LAUNCH("i2r_upsample_bilinear_add");
    return I2R_OK;
}

extern "C" int i2r_upsample_bilinear_add(const float* low, const float* res, float* out, int32_t n_img, int32_t low_h,
                                         int32_t low_w, int32_t scale, int32_t c, int32_t cs, int32_t act, void* stream) {
    i2r_up_args a = {};
    a.low = low; a.res = res; a.out = out; a.n_img = n_img; a.low_h = low_h; a.low_w = low_w; a.scale = scale; a.c = c; a.cs = cs; a.act = act;
    return i2r_upsample_bilinear_add_multi(&a, stream);
}
