// Boundary / glue kernels: NCHW stem conv, NHWC max-pool, NHWC -> NCHW heatmap head.  All HBM-bound
// element-wise-class work (a few % of the forward); written for coalesced 16-byte accesses.
#include "i2r_conv.h"  // (st_act4: fp32 / bf16 / f16 activation stores)

#ifndef I2R_STEM_FR
#define I2R_STEM_FR 2
#endif

namespace {

// 3x3 stride-2 pad-1 conv with tiny cin (3 = RGB crop, 1 = person box mask) + folded BN + ReLU.
// thread = one output pixel x 16 output channels; the G = cout/16 threads of a pixel are adjacent lanes and own INTERLEAVED 16-byte
// pieces of the pixel's channel row, so that every store instruction of a wave writes, per pixel, G x 16 contiguous bytes from G
// adjacent lanes (cout = 64: the four lanes of a quad fill one 64-byte segment, the texture addresser's fast case; with 16
// consecutive channels per lane every lane of a store hit its own segment and the kernel ran at 1.8 TB/s):
//   fp32 output:   lane cg owns the channel quadruples i G + cg, i = 0..3 (four 16-byte stores);
//   16-bit output: lane cg owns the channel octets i G + cg, i = 0..1 (two 16-byte stores of 8 packed values).
// w: [9][CIN][cout].
template <int CIN, int ODT>
__global__ __launch_bounds__(256) void stem_conv_k(const float* __restrict__ in, const float* __restrict__ w,
                                                   const float* __restrict__ bias, float* __restrict__ out, int n_img,
                                                   int in_h, int in_w, int out_h, int out_w, int cout, int out_cs, int n_src, int n_valid) {
    extern __shared__ __attribute__((aligned(16))) float wl[];  // [9*CIN][cout] weights + [cout] bias
    for (int i = threadIdx.x; i < (9 * CIN + 1) * cout; i += 256) wl[i] = i < 9 * CIN * cout ? w[i] : bias[i - 9 * CIN * cout];
    __syncthreads();
    const int groups = cout >> 4;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const int cg = (int)(gid % groups);
    const long long pix_raw = gid / groups;
    const bool valid = pix_raw < (long long)n_img * out_h * out_w;
    const long long pix = valid ? pix_raw : 0;
    const int ox = (int)(pix % out_w);
    const int oy = (int)((pix / out_w) % out_h);
    const int img = (int)(pix / ((long long)out_w * out_h));
    // images >= n_src are the horizontally MIRRORED copies of images 0..n_src-1 (flip test batched into one forward);
    // slots n_valid..n_src-1 are capacity padding of the program (they re-read the last real crop, their results are dropped)
    const int simg = min(img % n_src, n_valid - 1);
    const bool mirror = img >= n_src;

    // input taps first (9*CIN scalars), then the FMAs against the LDS-resident weights (same address across the 16 lanes
    // of a channel group -> LDS broadcast); keeps the kernel at ~60 VGPRs instead of hoisting 108 weight loads
    float xin[9 * CIN];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * 2 - 1 + ky;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox * 2 - 1 + kx;
            const bool ok = valid && iy >= 0 && iy < in_h && ix >= 0 && ix < in_w;
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci)
                xin[(ky * 3 + kx) * CIN + ci] = ok ? in[((size_t)(simg * CIN + ci) * in_h + iy) * in_w + (mirror ? in_w - 1 - ix : ix)] : 0.f;
        }
    }
    float acc[16];
    // float4 index (within a [cout] row) of register quadruple q of this lane
    auto f4 = [&](int q) { return ODT == 0 ? q * groups + cg : ((q >> 1) * groups + cg) * 2 + (q & 1); };
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(wl + 9 * CIN * cout + f4(q) * 4);
        acc[q * 4] = bv[0]; acc[q * 4 + 1] = bv[1]; acc[q * 4 + 2] = bv[2]; acc[q * 4 + 3] = bv[3];
    }
#pragma unroll 3
    for (int t = 0; t < 9 * CIN; ++t) {
        const float x = xin[t];
        const f32x4* wr = reinterpret_cast<const f32x4*>(wl + t * cout);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 wv = wr[f4(q)];
            acc[q * 4 + 0] = fmaf(x, wv[0], acc[q * 4 + 0]);
            acc[q * 4 + 1] = fmaf(x, wv[1], acc[q * 4 + 1]);
            acc[q * 4 + 2] = fmaf(x, wv[2], acc[q * 4 + 2]);
            acc[q * 4 + 3] = fmaf(x, wv[3], acc[q * 4 + 3]);
        }
    }
    if (!valid) return;
    if constexpr (ODT == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            st_act4<0>(out, (size_t)pix * out_cs + f4(q) * 4,
                       (f32x4){fmaxf(acc[q * 4], 0.f), fmaxf(acc[q * 4 + 1], 0.f), fmaxf(acc[q * 4 + 2], 0.f), fmaxf(acc[q * 4 + 3], 0.f)}, false);
    } else {  // the tower keeps its activations in 16 bit (bf16 / f16): 8 values = 16 bytes per store
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            unsigned short h[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = fmaxf(acc[i * 8 + e], 0.f);
                if constexpr (ODT == 1) h[e] = __builtin_bit_cast(unsigned short, (__bf16)v);
                else h[e] = __builtin_bit_cast(unsigned short, (_Float16)v);
            }
            uint4 u;
            u.x = h[0] | ((unsigned)h[1] << 16); u.y = h[2] | ((unsigned)h[3] << 16);
            u.z = h[4] | ((unsigned)h[5] << 16); u.w = h[6] | ((unsigned)h[7] << 16);
            *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(out) + (size_t)pix * out_cs + (i * groups + cg) * 8) = u;
        }
    }
}

// The same stem on the fp32 matrix pipe, for cout == 64 (every stem of the shipped models).  GEMM view: M = output pixels (16 per
// fragment, consecutive in the linear [img][oy][ox] order), N = 64 channels, K = 9 * CIN taps-times-channels padded to whole
// 4-wide MFMA steps (CIN = 3: 27 -> 28, CIN = 1: 9 -> 12).
//  * A: lane (m = l & 15, g = l >> 4) of step s supplies input value k = 4 s + g = (tap, channel) of pixel m: ONE scalar load per
//    lane and step straight from the NCHW boundary tensor (28 loads per pixel instead of the 108 of the VALU kernel).
//  * B: the weights of a wave live in 4 * KS registers, loaded once and reused over FR fragments.  The COLUMN of fragment nt that
//    lane n = l & 15 feeds is chosen as channel 4 n + nt -- so after the MFMAs lane (n, g) holds, for each of its four pixels
//    4 g + r, the four CONSECUTIVE channels 4 n .. 4 n + 3 in acc[0..3][r]: the NHWC store is one 16-byte (fp32) / 8-byte (16-bit)
//    access per pixel row with no transpose at all, the 16 lanes of a group writing one pixel's whole 256 / 128 contiguous bytes.
//  * 28 MFMAs per 16 pixels: 9 us of matrix-pipe time for 32 crops; the kernel is bound by its 3.1 MB per crop of output.
template <int CIN, int ODT, int FR>
__global__ __launch_bounds__(256) void stem_mfma_k(const float* __restrict__ in, const float* __restrict__ w,
                                                   const float* __restrict__ bias, float* __restrict__ out, int n_img,
                                                   int in_h, int in_w, int out_h, int out_w, int out_cs, int n_src, int n_valid) {
    constexpr int K = 9 * CIN, KS = (K + 3) / 4, COUT = 64;
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int total = n_img * out_h * out_w;  // (< 2^31: checked by the launcher; 32-bit index arithmetic throughout)
    const int pw = wave * (FR * 16);          // first pixel of this wave's FR fragments
    if (pw >= total) return;
    // (tap, channel) of this lane's k = 4 s + g in every step
    int kdy[KS], kdx[KS], kci[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int k = min(4 * s + g, K - 1), tap = k / CIN;
        kci[s] = k - tap * CIN;
        kdy[s] = tap / 3 - 1;
        kdx[s] = tap % 3 - 1;
    }
    // A side first: the input values of ALL FR fragments are in flight together (one memory latency per wave, not per fragment)
    float a[FR][KS];
#pragma unroll
    for (int f = 0; f < FR; ++f) {
        const int p = pw + f * 16 + n;
        const bool a_ok = p < total;
        const int pa = a_ok ? p : total - 1;
        const int row = pa / out_w, ox = pa - row * out_w, img = row / out_h, oy = row - img * out_h;
        // images >= n_src: horizontally mirrored copies (flip test); slots n_valid .. n_src - 1: capacity padding (see stem_conv_k)
        const int simg = min(img % n_src, n_valid - 1);
        const bool mirror = img >= n_src;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int iy = oy * 2 + kdy[s], ix = ox * 2 + kdx[s];
            const bool ok = a_ok && 4 * s + g < K && iy >= 0 && iy < in_h && ix >= 0 && ix < in_w;
            a[f][s] = ok ? in[((size_t)(simg * CIN + kci[s]) * in_h + iy) * in_w + (mirror ? in_w - 1 - ix : ix)] : 0.f;
        }
    }
    // this lane's weight column of every step (k = 4 s + g) and fragment (channel 4 n + nt); rows k >= K are padding
    float b[KS][4];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int k = 4 * s + g;
        const f32x4 wv = k < K ? *reinterpret_cast<const f32x4*>(w + k * COUT + 4 * n) : (f32x4){0.f, 0.f, 0.f, 0.f};
        b[s][0] = wv[0]; b[s][1] = wv[1]; b[s][2] = wv[2]; b[s][3] = wv[3];
    }
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + 4 * n);
#pragma unroll
    for (int f = 0; f < FR; ++f) {
        f32x4 acc[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = (f32x4){bv[nt], bv[nt], bv[nt], bv[nt]};
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = mfma16(a[f][s], b[s][nt], acc[nt]);
        // D side: pixels 4 g + r of the fragment, channels 4 n .. 4 n + 3
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int pd = pw + f * 16 + 4 * g + r;
            if (pd >= total) continue;
            const f32x4 v = {fmaxf(acc[0][r], 0.f), fmaxf(acc[1][r], 0.f), fmaxf(acc[2][r], 0.f), fmaxf(acc[3][r], 0.f)};
            st_act4<ODT>(out, (size_t)pd * out_cs + 4 * n, v, ODT != 0);
        }
    }
}

// PositionEmbeddingImage mode 'res' front end (position_embedding.py:14-17,93-95): conv_pre (1 -> 3, 3x3, pad 1, no bias) followed
// by torchvision resnet18's conv1 (3 -> 64, 7x7, stride 2, pad 3) + bn1 (folded) + ReLU, boundary NCHW mask -> NHWC.
// The two convolutions are NOT merged into one 9x9 filter: conv1 zero-pads conv_pre's OUTPUT, so near the border the composite is
// not a convolution of the mask.  One workgroup = an 8x8 output tile x 64 channels: the 23x23 mask patch goes to LDS, the 21x21x3
// conv_pre patch is computed into LDS (exact zeros outside the image), then thread (pixel, 16-channel group) runs the 147-tap
// dot products against the LDS-resident filter (VALU-bound: 0.23 GFLOP per crop).
__global__ __launch_bounds__(256) void pe_res_stem_k(const float* __restrict__ in, const float* __restrict__ w_pre,
                                                     const float* __restrict__ w7, const float* __restrict__ bias,
                                                     float* __restrict__ out, int n_img, int in_h, int in_w, int out_h, int out_w,
                                                     int out_cs, int n_src, int n_valid, int tiles_x, int tiles_y) {
    constexpr int T = 8, PW = 2 * T + 5, MW = PW + 2, COUT = 64;
    __shared__ __attribute__((aligned(16))) float wl[147 * COUT + COUT];
    __shared__ float ms[MW * MW];
    __shared__ float pre[3 * PW * PW];
    __shared__ float wp[27];
    const int tid = threadIdx.x;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int img = bid / tiles_y;
    const int simg = min(img % n_src, n_valid - 1);
    const bool mirror = img >= n_src;
    for (int i = tid; i < (147 * COUT + COUT) / 4; i += 256)
        reinterpret_cast<f32x4*>(wl)[i] = i < 147 * COUT / 4 ? reinterpret_cast<const f32x4*>(w7)[i] : reinterpret_cast<const f32x4*>(bias)[i - 147 * COUT / 4];
    if (tid < 27) wp[tid] = w_pre[tid];
    const int oy0 = ty * T, ox0 = tx * T;
    const int py0 = 2 * oy0 - 3, px0 = 2 * ox0 - 3;  // image coordinate of pre[.][0][0]
    for (int i = tid; i < MW * MW; i += 256) {
        const int y = py0 - 1 + i / MW, x = px0 - 1 + i % MW;
        float v = 0.f;
        if (y >= 0 && y < in_h && x >= 0 && x < in_w) v = in[((size_t)simg * in_h + y) * in_w + (mirror ? in_w - 1 - x : x)];
        ms[i] = v;
    }
    __syncthreads();
    for (int i = tid; i < 3 * PW * PW; i += 256) {
        const int c = i / (PW * PW), r = i - c * PW * PW;
        const int yy = r / PW, xx = r - yy * PW;
        const int y = py0 + yy, x = px0 + xx;
        float a = 0.f;
        if (y >= 0 && y < in_h && x >= 0 && x < in_w) {  // conv1 pads conv_pre's output with zeros
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) a = fmaf(ms[(yy + ky) * MW + xx + kx], wp[(ky * 3 + kx) * 3 + c], a);
        }
        pre[i] = a;
    }
    __syncthreads();
    const int cg = tid & 3, p = tid >> 2;
    const int ly = p >> 3, lx = p & 7;
    float acc[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(wl + 147 * COUT + cg * 16 + q * 4);
        acc[q * 4] = bv[0]; acc[q * 4 + 1] = bv[1]; acc[q * 4 + 2] = bv[2]; acc[q * 4 + 3] = bv[3];
    }
    for (int ky = 0; ky < 7; ++ky)
        for (int kx = 0; kx < 7; ++kx) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float x = pre[c * PW * PW + (2 * ly + ky) * PW + 2 * lx + kx];
                const f32x4* wr = reinterpret_cast<const f32x4*>(wl + ((ky * 7 + kx) * 3 + c) * COUT + cg * 16);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 wv = wr[q];
                    acc[q * 4 + 0] = fmaf(x, wv[0], acc[q * 4 + 0]);
                    acc[q * 4 + 1] = fmaf(x, wv[1], acc[q * 4 + 1]);
                    acc[q * 4 + 2] = fmaf(x, wv[2], acc[q * 4 + 2]);
                    acc[q * 4 + 3] = fmaf(x, wv[3], acc[q * 4 + 3]);
                }
            }
        }
    const int oy = oy0 + ly, ox = ox0 + lx;
    if (oy >= out_h || ox >= out_w) return;
    f32x4* o = reinterpret_cast<f32x4*>(out + (((size_t)img * out_h + oy) * out_w + ox) * out_cs + cg * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q)
        o[q] = (f32x4){fmaxf(acc[q * 4], 0.f), fmaxf(acc[q * 4 + 1], 0.f), fmaxf(acc[q * 4 + 2], 0.f), fmaxf(acc[q * 4 + 3], 0.f)};
}

// nn.MaxPool2d(3, 2, 1) on NHWC; thread = one output pixel x 4 channels (padding behaves as -inf)
__global__ __launch_bounds__(256) void maxpool_k(const float* __restrict__ in, float* __restrict__ out, int n_img,
                                                 int in_h, int in_w, int out_h, int out_w, int c4, int in_cs,
                                                 int out_cs) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const int cg = (int)(gid % c4);
    const long long pix = gid / c4;
    if (pix >= (long long)n_img * out_h * out_w) return;
    const int ox = (int)(pix % out_w);
    const int oy = (int)((pix / out_w) % out_h);
    const int img = (int)(pix / ((long long)out_w * out_h));
    const float ninf = -__builtin_inff();
    f32x4 m = (f32x4){ninf, ninf, ninf, ninf};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * 2 - 1 + ky;
        if (iy < 0 || iy >= in_h) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox * 2 - 1 + kx;
            if (ix < 0 || ix >= in_w) continue;
            const f32x4 v = *reinterpret_cast<const f32x4*>(in + ((size_t)(img * in_h + iy) * in_w + ix) * in_cs + cg * 4);
            m[0] = fmaxf(m[0], v[0]); m[1] = fmaxf(m[1], v[1]); m[2] = fmaxf(m[2], v[2]); m[3] = fmaxf(m[3], v[3]);
        }
    }
    *reinterpret_cast<f32x4*>(out + (size_t)pix * out_cs + cg * 4) = m;
}

// final 1x1 conv with bias: NHWC features -> NCHW heatmaps.  HBM-bound (reads the [S, h, w, cin] map once).  Four lanes share a pixel:
// lane q of a quad takes the channel groups 4 i + q, so each load instruction of a wave covers 16 pixels x 64 contiguous bytes (the
// quad rule of the texture addresser; one pixel per lane -- 384-byte strides -- took 4x the addresser time and left 384 workgroups
// for 256 CUs).  The weights sit in LDS as [cin / 4][JP] float4 (lanes of equal q read the same address: broadcast); the four
// partial sums of a joint are combined by a transposing reduction (3 shuffles per 4 joints) that leaves joint 4 k + q on lane q,
// which writes it to the NCHW plane of that joint (16 consecutive pixels = 64 bytes per lane group).
template <int JP>
__global__ __launch_bounds__(256) void head_k(const float* __restrict__ in, const float* __restrict__ w,
                                              const float* __restrict__ bias, float* __restrict__ out, int n_img, int hw,
                                              int cin, int in_cs, int cout) {
    extern __shared__ __attribute__((aligned(16))) f32x4 wl4[];  // [cin / 4][JP]
    const int cin4 = cin >> 2;
    for (int i = threadIdx.x; i < cin4 * JP; i += 256) {
        const int c4 = i / JP, j = i - c4 * JP;
        wl4[i] = j < cout ? *reinterpret_cast<const f32x4*>(w + (size_t)j * cin + c4 * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    const int q = threadIdx.x & 3;
    const long long pix_raw = ((long long)blockIdx.x * 256 + threadIdx.x) >> 2;
    const bool valid = pix_raw < (long long)n_img * hw;
    const long long pix = valid ? pix_raw : 0;
    float acc[JP];
#pragma unroll
    for (int j = 0; j < JP; ++j) acc[j] = 0.f;
    const f32x4* x = reinterpret_cast<const f32x4*>(in + (size_t)pix * in_cs);
    for (int c4 = q; c4 < cin4; c4 += 4) {  // (all loads of a pixel in flight at once measured slower: 28 vs 23 us at 32 crops)
        const f32x4 v = x[c4];
        const f32x4* wr = wl4 + c4 * JP;
#pragma unroll
        for (int j = 0; j < JP; ++j) {
            const f32x4 wv = wr[j];
            acc[j] = fmaf(v[0], wv[0], fmaf(v[1], wv[1], fmaf(v[2], wv[2], fmaf(v[3], wv[3], acc[j]))));
        }
    }
    const int img = (int)(pix / hw);
    const int p = (int)(pix - (long long)img * hw);
#pragma unroll
    for (int k = 0; k < JP / 4; ++k) {
        // transposing reduction over the quad: lane q ends with the full sum of joint 4 k + q
        const float a0 = acc[4 * k], a1 = acc[4 * k + 1], a2 = acc[4 * k + 2], a3 = acc[4 * k + 3];
        const bool o1 = q & 1, o2 = q & 2;
        const float k0 = (o1 ? a1 : a0) + __shfl_xor(o1 ? a0 : a1, 1);
        const float k1 = (o1 ? a3 : a2) + __shfl_xor(o1 ? a2 : a3, 1);
        const float r = (o2 ? k1 : k0) + __shfl_xor(o2 ? k0 : k1, 2);
        const int j = 4 * k + q;
        if (valid && j < cout) out[((size_t)img * cout + j) * hw + p] = r + bias[j];
    }
}

// The same head on the fp32 matrix pipe (cin a multiple of 16, <= 128).  M = 16 consecutive pixels per fragment, N = joints (one or
// two 16-wide fragments), K = cin with the conv kernel's k-permutation: lane (m = l & 15, g = l >> 4) loads the 16 bytes
// in[pixel m][16 j + 4 g .. + 3] and feeds them to four consecutive MFMAs, the weight fragment of lane (n, g) being
// w[joint n][16 j + 4 g + s] -- so a pixel row is read with cin / 16 sixteen-byte loads per lane, ALL of a wave's FR fragments in
// flight together, and no LDS / shuffle reduction is left.  D: lane (n, g) holds joint n of pixels 4 g .. 4 g + 3 -- four
// consecutive floats of the joint's NCHW plane: one 16-byte store.
template <int NF, int FR>
__global__ __launch_bounds__(256) void head_mfma_k(const float* __restrict__ in, const float* __restrict__ w,
                                                   const float* __restrict__ bias, float* __restrict__ out, int total, int hw,
                                                   int cin, int in_cs, int cout) {
    constexpr int CJ = 8;  // 16-channel steps provided for (cin <= 128)
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int pw = wave * (FR * 16);
    if (pw >= total) return;
    const int cj = cin >> 4;
    f32x4 va[FR][CJ];
#pragma unroll
    for (int f = 0; f < FR; ++f) {
        const int p = pw + f * 16 + n;
        const bool ok = p < total;
        const f32x4* x = reinterpret_cast<const f32x4*>(in + (size_t)(ok ? p : total - 1) * in_cs) + g;
#pragma unroll
        for (int j = 0; j < CJ; ++j) va[f][j] = (ok && j < cj) ? x[j * 4] : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    f32x4 wb[NF][CJ];
    float bj[NF];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
        const int jn = nf * 16 + n;
        bj[nf] = jn < cout ? bias[jn] : 0.f;
#pragma unroll
        for (int j = 0; j < CJ; ++j)
            wb[nf][j] = (jn < cout && j < cj) ? *reinterpret_cast<const f32x4*>(w + (size_t)jn * cin + j * 16 + g * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const bool vec = (hw & 3) == 0;  // then a lane's four pixels lie in one image and are 16-byte aligned in the joint's plane
#pragma unroll
    for (int f = 0; f < FR; ++f) {
        f32x4 acc[NF];
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) acc[nf] = (f32x4){bj[nf], bj[nf], bj[nf], bj[nf]};
#pragma unroll
        for (int j = 0; j < CJ; ++j)
            if (j < cj) {
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int nf = 0; nf < NF; ++nf) acc[nf] = mfma16(va[f][j][s], wb[nf][j][s], acc[nf]);
            }
        const int pd = pw + f * 16 + 4 * g;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
            const int jn = nf * 16 + n;
            if (jn >= cout || pd >= total) continue;
            if (vec && pd + 3 < total) {
                const int img = pd / hw, p = pd - img * hw;
                *reinterpret_cast<f32x4*>(out + ((size_t)img * cout + jn) * hw + p) = acc[nf];
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = pd + r;
                    if (q >= total) break;
                    const int img = q / hw, p = q - img * hw;
                    out[((size_t)img * cout + jn) * hw + p] = acc[nf][r];
                }
            }
        }
    }
}

// closing pass of an HRNet fuse sum: out = act((base + up(t1)) + up(t2)), nearest-neighbour up-sampling by 2^k.  thread = 4 channels
// of one output pixel (consecutive lanes = consecutive channel quadruples of a pixel, then the next pixel: fully coalesced);
// HBM-bound: reads base once, writes out once, the low-resolution maps come from L2.
template <int DT>
__global__ __launch_bounds__(256) void fuse_up_add_k(const float* base, const float* __restrict__ t1, int sh1,  // (base may alias out)
                                                     const float* __restrict__ t2, int sh2, float* out, long long nq, int h,
                                                     int w, int cs4, int act) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= nq) return;
    const int c4 = (int)(gid % cs4);
    const long long pix = gid / cs4;
    const int x = (int)(pix % w);
    const int y = (int)((pix / w) % h);
    const long long n = pix / ((long long)w * h);
    const bool h16 = DT != 0;
    f32x4 v = ld_act4<DT>(base, (size_t)gid * 4, h16);
    {
        const int hh = h >> sh1, ww = w >> sh1;
        v += ld_act4<DT>(t1, ((size_t)(n * hh + (y >> sh1)) * ww + (x >> sh1)) * cs4 * 4 + c4 * 4, h16);
    }
    if (t2) {
        const int hh = h >> sh2, ww = w >> sh2;
        v += ld_act4<DT>(t2, ((size_t)(n * hh + (y >> sh2)) * ww + (x >> sh2)) * cs4 * 4 + c4 * 4, h16);
    }
    if (act == 1) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
    st_act4<DT>(out, (size_t)gid * 4, v, h16);
}

}  // namespace

extern "C" int i2r_fuse_up_add(const float* base, const float* t1, int32_t s1, const float* t2, int32_t s2, float* out, int32_t n_img,
                               int32_t h, int32_t w, int32_t cs, int32_t act, int32_t dt, void* stream) {
    I2R_CHECK_ARG(base && t1 && out, "i2r_fuse_up_add: null pointer");
    auto shift = [](int s) { int k = 0; while ((1 << k) < s) ++k; return (1 << k) == s ? k : -1; };
    const int sh1 = shift(s1), sh2 = t2 ? shift(s2) : 0;
    I2R_CHECK_ARG(sh1 >= 1 && sh2 >= 0 && h % s1 == 0 && w % s1 == 0 && (!t2 || (h % s2 == 0 && w % s2 == 0)),
                  "i2r_fuse_up_add: scales %d / %d must be powers of two dividing %dx%d", s1, s2, h, w);
    I2R_CHECK_ARG(cs > 0 && cs % 4 == 0 && (act == 0 || act == 1) && dt >= 0 && dt <= 2, "i2r_fuse_up_add: cs=%d act=%d dt=%d", cs, act, dt);
    I2R_CHECK_ARG(t1 != out && t2 != out, "i2r_fuse_up_add: a low-resolution term aliases out");
    const long long nq = (long long)n_img * h * w * (cs / 4);
    const unsigned nblk = (unsigned)((nq + 255) / 256);
    typedef void (*fn_t)(const float*, const float*, int, const float*, int, float*, long long, int, int, int, int);
    static const fn_t fns[3] = {fuse_up_add_k<0>, fuse_up_add_k<1>, fuse_up_add_k<2>};
    i2r_launch(fns[dt], dim3(nblk), dim3(256), 0, (hipStream_t)stream, base, t1, sh1, t2, sh2, out, nq, h, w, cs / 4, act);
    I2R_CHECK_LAUNCH("i2r_fuse_up_add");
    return I2R_OK;
}

extern "C" int i2r_stem_conv(const float* in_nchw, const float* w, const float* bias, float* out_nhwc, int32_t n_img,
                             int32_t cin, int32_t in_h, int32_t in_w, int32_t cout, int32_t out_cs, int32_t n_src, int32_t n_valid,
                             int32_t out_dt, void* stream) {
    I2R_CHECK_ARG(in_nchw && w && bias && out_nhwc, "i2r_stem_conv: null pointer");
    I2R_CHECK_ARG(out_dt >= 0 && out_dt <= 2, "i2r_stem_conv: out_dt %d (0 fp32, 1 bf16, 2 f16)", out_dt);
    I2R_CHECK_ARG(n_src >= 1 && (n_img == n_src || n_img == 2 * n_src) && n_valid >= 1 && n_valid <= n_src,
                  "i2r_stem_conv: n_img=%d n_src=%d n_valid=%d", n_img, n_src, n_valid);
    I2R_CHECK_ARG(cout > 0 && cout % 16 == 0 && out_cs >= cout && out_cs % 4 == 0, "i2r_stem_conv: cout=%d out_cs=%d", cout, out_cs);
    I2R_CHECK_ARG(cin == 1 || cin == 3, "i2r_stem_conv: cin=%d (1 or 3)", cin);
    const int out_h = (in_h - 1) / 2 + 1, out_w = (in_w - 1) / 2 + 1;
    const long long nthr = (long long)n_img * out_h * out_w * (cout / 16);
    const unsigned nblk = (unsigned)((nthr + 255) / 256);
    if (cout == 64) {  // matrix-pipe stem (every shipped model); other widths keep the VALU kernel below
        typedef void (*mfma_fn)(const float*, const float*, const float*, float*, int, int, int, int, int, int, int, int);
        constexpr int FR = I2R_STEM_FR;  // fragments (16 pixels each) per wave
        static const mfma_fn mf[2][3] = {{stem_mfma_k<1, 0, FR>, stem_mfma_k<1, 1, FR>, stem_mfma_k<1, 2, FR>},
                                         {stem_mfma_k<3, 0, FR>, stem_mfma_k<3, 1, FR>, stem_mfma_k<3, 2, FR>}};
        const long long frags = ((long long)n_img * out_h * out_w + 15) / 16;
        const long long nb = (frags + 4 * FR - 1) / (4 * FR);
        I2R_CHECK_ARG(nb > 0 && frags * 16 + 64 * FR < (1ll << 31), "i2r_stem_conv: grid");
        i2r_launch(mf[cin == 3][out_dt], dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, in_nchw, w, bias, out_nhwc, n_img,
                           in_h, in_w, out_h, out_w, out_cs, n_src, n_valid);
        I2R_CHECK_LAUNCH("i2r_stem_conv");
        return I2R_OK;
    }
    typedef void (*stem_fn)(const float*, const float*, const float*, float*, int, int, int, int, int, int, int, int, int);
    static const stem_fn fns[2][3] = {{stem_conv_k<1, 0>, stem_conv_k<1, 1>, stem_conv_k<1, 2>}, {stem_conv_k<3, 0>, stem_conv_k<3, 1>, stem_conv_k<3, 2>}};
    i2r_launch(fns[cin == 3][out_dt], dim3(nblk), dim3(256), (size_t)(9 * cin + 1) * cout * sizeof(float), (hipStream_t)stream, in_nchw, w,
                       bias, out_nhwc, n_img, in_h, in_w, out_h, out_w, cout, out_cs, n_src, n_valid);
    I2R_CHECK_LAUNCH("i2r_stem_conv");
    return I2R_OK;
}

extern "C" int i2r_pe_res_stem(const float* mask_nchw, const float* w_pre, const float* w7, const float* bias, float* out_nhwc,
                               int32_t n_img, int32_t in_h, int32_t in_w, int32_t cout, int32_t out_cs, int32_t n_src, int32_t n_valid,
                               void* stream) {
    I2R_CHECK_ARG(mask_nchw && w_pre && w7 && bias && out_nhwc, "i2r_pe_res_stem: null pointer");
    I2R_CHECK_ARG(n_src >= 1 && (n_img == n_src || n_img == 2 * n_src) && n_valid >= 1 && n_valid <= n_src,
                  "i2r_pe_res_stem: n_img=%d n_src=%d n_valid=%d", n_img, n_src, n_valid);
    I2R_CHECK_ARG(cout == 64 && out_cs >= cout && out_cs % 4 == 0, "i2r_pe_res_stem: cout=%d (resnet18 conv1 has 64) out_cs=%d", cout, out_cs);
    const int out_h = (in_h - 1) / 2 + 1, out_w = (in_w - 1) / 2 + 1;
    const int tiles_y = (out_h + 7) / 8, tiles_x = (out_w + 7) / 8;
    i2r_launch(pe_res_stem_k, dim3((unsigned)(n_img * tiles_y * tiles_x)), dim3(256), 0, (hipStream_t)stream, mask_nchw, w_pre, w7,
                       bias, out_nhwc, n_img, in_h, in_w, out_h, out_w, out_cs, n_src, n_valid, tiles_x, tiles_y);
    I2R_CHECK_LAUNCH("i2r_pe_res_stem");
    return I2R_OK;
}

namespace {
// PositionEmbeddingImage mode 'cat_vec' (position_embedding.py:19-23, 69-87): the bbox mask of a person max-pooled `rate` times
// (MaxPool2d(3, 2, 1): r cascaded pools = one max over the window [i 2^r - (2^r - 1), i 2^r + (2^r - 1)] clipped to the image, the
// padding being -inf), flattened, through nn.Linear(th*tw, vec) -- one vector per person, repeated over all th*tw tokens of the person.
// One workgroup per person crop: pooled map and vector in LDS, then the broadcast store into channels [c0, c0 + vec) of the token rows
// (zeros up to c_end: the row padding, when this is the last part of the row).
__global__ __launch_bounds__(256) void pe_cat_vec_k(const float* __restrict__ mask, const float* __restrict__ wfc, const float* __restrict__ bfc,
                                                    float* __restrict__ out, int in_h, int in_w, int th, int tw, int rate, int vec, int out_cs, int c0,
                                                    int c_end, int n_src, int n_valid) {
    extern __shared__ float sm[];  // pooled[th*tw] | v[c_end - c0]
    const int img = blockIdx.x, tid = threadIdx.x;
    const bool mirror = img >= n_src;
    const int src = min(mirror ? img - n_src : img, n_valid - 1);
    const float* m = mask + (size_t)src * in_h * in_w;
    const int P = th * tw, R = (1 << rate) - 1;
    float* pooled = sm;
    float* v = sm + P;
    for (int p = tid; p < P; p += 256) {
        const int iy = p / tw, ix = p - iy * tw;
        const int y0 = max((iy << rate) - R, 0), y1 = min((iy << rate) + R, in_h - 1);
        const int x0 = max((ix << rate) - R, 0), x1 = min((ix << rate) + R, in_w - 1);
        float mx = -INFINITY;
        for (int y = y0; y <= y1; ++y)
            for (int x = x0; x <= x1; ++x) mx = fmaxf(mx, m[y * in_w + (mirror ? in_w - 1 - x : x)]);
        pooled[p] = mx;
    }
    __syncthreads();
    for (int c = tid; c < c_end - c0; c += 256) {
        float a = 0.f;
        if (c < vec) {
            a = bfc[c];
            for (int p = 0; p < P; ++p) a = fmaf(wfc[(size_t)c * P + p], pooled[p], a);
        }
        v[c] = a;
    }
    __syncthreads();
    const int nc = c_end - c0;
    float* o = out + (size_t)img * P * out_cs + c0;
    for (int i = tid; i < P * nc; i += 256) {
        const int t = i / nc, c = i - t * nc;
        o[(size_t)t * out_cs + c] = v[c];
    }
}
}  // namespace

extern "C" int i2r_pe_cat_vec(const i2r_pe_cat_vec_args* a, void* stream) {
    I2R_CHECK_ARG(a && a->in && a->w && a->bias && a->out, "i2r_pe_cat_vec: null pointer");
    I2R_CHECK_ARG(a->n_src >= 1 && (a->n_img == a->n_src || a->n_img == 2 * a->n_src) && a->n_valid >= 1 && a->n_valid <= a->n_src,
                  "i2r_pe_cat_vec: n_img=%d n_src=%d n_valid=%d", a->n_img, a->n_src, a->n_valid);
    I2R_CHECK_ARG(a->rate >= 0 && a->rate <= 8 && a->th >= 1 && a->tw >= 1 && ((a->in_h - 1) >> a->rate) + 1 == a->th && ((a->in_w - 1) >> a->rate) + 1 == a->tw,
                  "i2r_pe_cat_vec: %dx%d pooled %d times is not %dx%d", a->in_h, a->in_w, a->rate, a->th, a->tw);
    I2R_CHECK_ARG(a->vec >= 1 && a->c0 >= 0 && a->c_end >= a->c0 + a->vec && a->c_end <= a->out_cs, "i2r_pe_cat_vec: channels [%d, %d + %d) .. %d of %d",
                  a->c0, a->c0, a->vec, a->c_end, a->out_cs);
    const size_t lds = (size_t)(a->th * a->tw + a->c_end - a->c0) * sizeof(float);
    I2R_CHECK_ARG(lds <= 64 * 1024, "i2r_pe_cat_vec: %zu bytes of LDS", lds);
    i2r_launch(pe_cat_vec_k, dim3((unsigned)a->n_img), dim3(256), lds, (hipStream_t)stream, a->in, a->w, a->bias, a->out, a->in_h, a->in_w,
                       a->th, a->tw, a->rate, a->vec, a->out_cs, a->c0, a->c_end, a->n_src, a->n_valid);
    I2R_CHECK_LAUNCH("i2r_pe_cat_vec");
    return I2R_OK;
}

extern "C" int i2r_maxpool3x3s2(const float* in, float* out, int32_t n_img, int32_t in_h, int32_t in_w, int32_t c,
                                int32_t in_cs, int32_t out_cs, void* stream) {
    I2R_CHECK_ARG(in && out && in != out, "i2r_maxpool3x3s2: bad pointers");
    I2R_CHECK_ARG(c > 0 && c % 4 == 0 && in_cs % 4 == 0 && out_cs % 4 == 0 && c <= in_cs && c <= out_cs,
                  "i2r_maxpool3x3s2: c=%d in_cs=%d out_cs=%d", c, in_cs, out_cs);
    const int out_h = (in_h - 1) / 2 + 1, out_w = (in_w - 1) / 2 + 1;
    const long long nthr = (long long)n_img * out_h * out_w * (c / 4);
    i2r_launch(maxpool_k, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, out, n_img,
                       in_h, in_w, out_h, out_w, c / 4, in_cs, out_cs);
    I2R_CHECK_LAUNCH("i2r_maxpool3x3s2");
    return I2R_OK;
}

extern "C" int i2r_head(const float* in, const float* w, const float* bias, float* out_nchw, int32_t n_img, int32_t h,
                        int32_t w_, int32_t cin, int32_t in_cs, int32_t cout, void* stream) {
    I2R_CHECK_ARG(in && w && bias && out_nchw, "i2r_head: null pointer");
    I2R_CHECK_ARG(cin % 4 == 0 && in_cs % 4 == 0 && cin <= in_cs && cout >= 1 && cout <= 32, "i2r_head: cin=%d cout=%d", cin, cout);
    const long long npix = (long long)n_img * h * w_;
    if (cin % 16 == 0 && cin <= 128 && npix + 256 < (1ll << 31)) {  // matrix-pipe head (every shipped model); else the VALU kernel below
        constexpr int FR = 2;
        typedef void (*mfma_fn)(const float*, const float*, const float*, float*, int, int, int, int, int);
        const mfma_fn mf = cout <= 16 ? head_mfma_k<1, FR> : head_mfma_k<2, FR>;
        const long long waves = (npix + 16 * FR - 1) / (16 * FR);
        i2r_launch(mf, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, in, w, bias, out_nchw, (int)npix, h * w_, cin,
                           in_cs, cout);
        I2R_CHECK_LAUNCH("i2r_head");
        return I2R_OK;
    }
    const unsigned nblk = (unsigned)((npix * 4 + 255) / 256);  // 4 lanes per pixel
    typedef void (*head_fn)(const float*, const float*, const float*, float*, int, int, int, int, int);
    const int jp = cout <= 16 ? 16 : cout <= 20 ? 20 : 32;
    const head_fn fn = jp == 16 ? head_k<16> : jp == 20 ? head_k<20> : head_k<32>;
    i2r_launch(fn, dim3(nblk), dim3(256), (size_t)(cin / 4) * jp * sizeof(f32x4), (hipStream_t)stream, in, w, bias, out_nchw, n_img, h * w_,
                       cin, in_cs, cout);
    I2R_CHECK_LAUNCH("i2r_head");
    return I2R_OK;
}
