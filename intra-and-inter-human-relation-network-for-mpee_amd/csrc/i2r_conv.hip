// Fused convolution for gfx950: implicit GEMM on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32).
//
// Design (MI355X-first, see DESIGN.md "conv_igemm_f32"):
//  * activations NHWC fp32; one workgroup (4 waves) owns a TH x TW tile of output pixels x a block of
//    output channels; GEMM M = pixels, N = cout, K = taps x cin.
//  * the input PATCH of the tile (halo included) is staged ONCE per channel chunk into LDS in a
//    channel-group-major image  lds[cg][patch_pixel] (float4 = 4 consecutive channels), so all taps of the
//    filter read it at shifted addresses: im2col happens in the LDS address, HBM/L2 sees each input
//    element once per workgroup.  A fragments are ds_read_b128: lane (pixel = l&15, g = l>>4) reads
//    channels 4g..4g+3 of a 16-channel step -> feeds FOUR consecutive MFMAs (k permuted so that MFMA step s
//    contracts channels {4g+s}); with plane strides that are multiples of 256 B the 16-lane groups of
//    ds_read_b128 hit distinct bank slots.
//  * weights are pre-packed "k4":  w[tap][cin/4][cout_pad][4]  so a B fragment (cout = l&15, g) is one 16-byte
//    global load per 4 MFMAs, streamed from L2 straight to registers (no LDS, no barrier in the K loop);
//    the fp32 MFMA is slow enough (32 cycles) that 1 KiB of L2 traffic per 4*MT MFMAs per wave hides.
//  * epilogue fuses folded-BN bias, up to two residual inputs, ReLU and the nearest-neighbour upsample
//    scatter of the HRNet fuse layers (each destination element is owned by exactly one lane, so
//    accumulating into `out` in place through res1 == out is race-free).
#include "i2r_common.h"

namespace {

struct ConvK {
    const float* in;
    const float* in2;
    const float* w;
    const float* bias;
    const float* res1;
    const float* res2;
    const float* res_post;
    float* out;
    int n_img, in_h, in_w, in_cs, cin;
    int conv_h, conv_w, out_h, out_w, out_cs, cout, cout_pad;
    int stride, iy0, ix0, ntaps;
    int tap_kh, tap_kw;  // taps form a dense kh x kw grid, row-major: tap t sits at patch offset (t / kw, t % kw)
    int out_step, out_off_y, out_off_x, rep, relu;
    int tile_h, tile_w, tiles_y, tiles_x, n_cblk;
    int ph, pw, plane;  // patch dims (pixels) and plane stride (float4 slots, multiple of 16)
    int ck;             // channels staged per pass (multiple of 16)
};

constexpr int kMaxPP = 5;  // patch pixels per thread (256 threads) -> patches up to 1280 pixels

template <int MT, int NT, int WM, int WN>
__global__ __launch_bounds__(256) void conv_igemm_f32(const ConvK p) {
    extern __shared__ __attribute__((aligned(16))) f32x4 lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 15, g = lane >> 4;

    int bid = blockIdx.x;
    const int cb = bid % p.n_cblk;
    bid /= p.n_cblk;
    const int tile_x = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int tile_y = bid % p.tiles_y;
    const int img = bid / p.tiles_y;
    const int oy0 = tile_y * p.tile_h, ox0 = tile_x * p.tile_w;
    const int py0 = oy0 * p.stride + p.iy0, px0 = ox0 * p.stride + p.ix0;
    const int phw = p.ph * p.pw;

    // patch pixel -> global float offset of channel 0 (clamped to 0 outside the image, with a validity bit);
    // channel independent, so computed once
    int goff[kMaxPP];
    bool gval[kMaxPP];
#pragma unroll
    for (int j = 0; j < kMaxPP; ++j) {
        const int pp = tid + j * 256;
        goff[j] = 0;
        gval[j] = false;
        if (pp < phw) {
            const int py = pp / p.pw, px = pp - py * p.pw;
            const int iy = py0 + py, ix = px0 + px;
            if (iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w) {
                goff[j] = ((img * p.in_h + iy) * p.in_w + ix) * p.in_cs;
                gval[j] = true;
            }
        }
    }
    const int npp = (phw + 255) >> 8;  // patch pixels per thread actually in use (wave-uniform)

    // A-fragment patch-pixel base of this lane's pixel in each M fragment
    const int tile_px = p.tile_h * p.tile_w;
    int ppix[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int m = (wm * MT + mt) * 16 + li;
        if (m >= tile_px) m = 0;  // padded rows compute garbage-free duplicates; masked at the store
        const int ty = m / p.tile_w, tx = m - ty * p.tile_w;
        ppix[mt] = ty * p.stride * p.pw + tx * p.stride;
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int n_base = (cb * WN + wn) * NT * 16;
    const int cin4 = p.cin >> 2;
    // lane's weight pointer: float4 index ((tap*cin4 + cg) * cout_pad + n)
    const f32x4* wq = reinterpret_cast<const f32x4*>(p.w) + n_base + li;

    for (int c0 = 0; c0 < p.cin; c0 += p.ck) {
        const int ckc = min(p.ck, p.cin - c0);
        const int ncg = ckc >> 2;
        if (c0 != 0) __syncthreads();
        // ---- stage the patch: lds[cg][pp] = in[pixel(pp)][c0 + 4cg .. +3]; 4 channel groups per trip so the
        //      global loads of a trip are all in flight before the first LDS store waits for them ----
        for (int cg0 = 0; cg0 < ncg; cg0 += 4) {
            f32x4 v[4][kMaxPP];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < kMaxPP; ++j)
                    if (j < npp) v[u][j] = *reinterpret_cast<const f32x4*>(p.in + goff[j] + c0 + (cg0 + u) * 4);
            if (p.in2) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < kMaxPP; ++j)
                        if (j < npp) v[u][j] += *reinterpret_cast<const f32x4*>(p.in2 + goff[j] + c0 + (cg0 + u) * 4);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < kMaxPP; ++j) {
                    const int pp = tid + j * 256;
                    if (j < npp && pp < phw)
                        lds[(cg0 + u) * p.plane + pp] = gval[j] ? v[u][j] : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
        }
        __syncthreads();

        // ---- K loop over (tap, 16-channel step), software-pipelined one step ahead with two named register
        //      sets (no register copies, so the compiler's vmcnt/lgkmcnt waits land at the first use) ----
        const int ncs = ckc >> 4;
        const int nit = p.ntaps * ncs;
        const f32x4* const wp0 = wq + (size_t)((c0 >> 2) + g) * p.cout_pad;  // step (tap 0, cs 0) of this chunk
        const f32x4* wp = wp0;
        const size_t inc_cs = (size_t)4 * p.cout_pad;
        const size_t inc_tap = (size_t)(cin4 - (ncs - 1) * 4) * p.cout_pad;
        int cs_n = 0, tx_n = 0, ty_n = 0;  // position of the next step to fetch
        auto fetch = [&](f32x4(&a)[MT], f32x4(&b)[NT]) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b[nt] = wp[nt * 16];
            const int abase = (cs_n * 4 + g) * p.plane + ty_n * p.pw + tx_n;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[mt] = lds[abase + ppix[mt]];
            if (++cs_n == ncs) {
                cs_n = 0;
                wp += inc_tap;
                if (++tx_n == p.tap_kw) {
                    tx_n = 0;
                    if (++ty_n == p.tap_kh) { ty_n = 0; wp = wp0; }  // wrap: the look-ahead past the last step stays in bounds
                }
            } else {
                wp += inc_cs;
            }
        };
        auto fma_step = [&](const f32x4(&a)[MT], const f32x4(&b)[NT]) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma16(a[mt][s], b[nt][s], acc[mt][nt]);
        };
        f32x4 a0[MT], a1[MT], b0[NT], b1[NT];
        fetch(a0, b0);
        int it = 0;
        for (; it + 2 <= nit; it += 2) {
            fetch(a1, b1);
            fma_step(a0, b0);
            fetch(a0, b0);  // unconditional (wraps after the last step) so both halves keep counted waits
            fma_step(a1, b1);
        }
        if (it < nit) fma_step(a0, b0);
    }

    // ---- epilogue: D layout col = l&15 -> cout, rows 4g+r -> pixels ----
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = n_base + nt * 16 + li;
        if (n >= p.cout) continue;
        const float bv = p.bias[n];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m0 = (wm * MT + mt) * 16 + g * 4;
            int ty = m0 / p.tile_w, tx = m0 - ty * p.tile_w;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int oy = oy0 + ty, ox = ox0 + tx;
                if (m0 + r < tile_px && oy < p.conv_h && ox < p.conv_w) {
                    const float v0 = acc[mt][nt][r] + bv;
                    const int by = oy * p.out_step + p.out_off_y, bx = ox * p.out_step + p.out_off_x;
                    for (int ry = 0; ry < p.rep; ++ry)
                        for (int rx = 0; rx < p.rep; ++rx) {
                            const size_t o = ((size_t)(img * p.out_h + by + ry) * p.out_w + bx + rx) * p.out_cs + n;
                            float v = v0;
                            if (p.res1) v += p.res1[o];
                            if (p.res2) v += p.res2[o];
                            if (p.relu) v = fmaxf(v, 0.f);
                            if (p.res_post) v += p.res_post[o];
                            p.out[o] = v;
                        }
                }
                if (++tx == p.tile_w) { tx = 0; ++ty; }
            }
        }
    }
}

typedef void (*conv_fn)(const ConvK);

template <int MT, int NT, int WN>
conv_fn pick() {
    return conv_igemm_f32<MT, NT, 4 / WN, WN>;
}

template <int NT, int WN>
conv_fn pick_mt(int mt) {
    switch (mt) {
        case 1: return pick<1, NT, WN>();
        case 2: return pick<2, NT, WN>();
        case 3: return pick<3, NT, WN>();
        case 4: return pick<4, NT, WN>();
    }
    return nullptr;
}

template <int NT>
conv_fn pick_wn(int wn, int mt) {
    switch (wn) {
        case 1: return pick_mt<NT, 1>(mt);
        case 2: return pick_mt<NT, 2>(mt);
        case 4: return pick_mt<NT, 4>(mt);
    }
    return nullptr;
}

conv_fn pick_kernel(int nt, int wn, int mt) {
    switch (nt) {
        case 1: return pick_wn<1>(wn, mt);
        case 2: return pick_wn<2>(wn, mt);
        case 3: return pick_wn<3>(wn, mt);
        case 4: return pick_wn<4>(wn, mt);
        case 5: return pick_wn<5>(wn, mt);
    }
    return nullptr;
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

}  // namespace

extern "C" int i2r_conv(const i2r_conv_desc* d, void* stream) {
    I2R_CHECK_ARG(d && d->in && d->w && d->bias && d->out, "i2r_conv: null pointer");
    I2R_CHECK_ARG(d->cin > 0 && d->cin % 16 == 0 && d->cin <= d->in_cs && d->in_cs % 4 == 0,
                  "i2r_conv: cin=%d must be a multiple of 16 and <= in_cs=%d (in_cs %% 4 == 0)", d->cin, d->in_cs);
    I2R_CHECK_ARG(d->cout > 0 && d->cout_pad % 16 == 0 && d->cout <= d->cout_pad && d->cout <= d->out_cs,
                  "i2r_conv: cout=%d cout_pad=%d out_cs=%d", d->cout, d->cout_pad, d->out_cs);
    I2R_CHECK_ARG(d->stride == 1 || d->stride == 2, "i2r_conv: stride %d", d->stride);
    I2R_CHECK_ARG(d->ntaps >= 1 && d->ntaps <= I2R_MAX_TAPS, "i2r_conv: ntaps %d", d->ntaps);
    I2R_CHECK_ARG(d->rep >= 1 && d->out_step >= 1, "i2r_conv: rep/out_step");
    I2R_CHECK_ARG((d->conv_h - 1) * d->out_step + d->out_off_y + d->rep <= d->out_h &&
                      (d->conv_w - 1) * d->out_step + d->out_off_x + d->rep <= d->out_w,
                  "i2r_conv: destination grid exceeds out tensor");
    I2R_CHECK_ARG(d->in2 != (const float*)d->out && d->in != (const float*)d->out, "i2r_conv: out aliases in");

    int max_dy = 0, max_dx = 0;
    for (int t = 0; t < d->ntaps; ++t) {
        I2R_CHECK_ARG(d->dy[t] >= 0 && d->dx[t] >= 0, "i2r_conv: negative tap offset");
        if (d->dy[t] > max_dy) max_dy = d->dy[t];
        if (d->dx[t] > max_dx) max_dx = d->dx[t];
    }

    // ---- fragment decomposition of the output channels ----
    const int nfrag = d->cout_pad / 16;
    int nt = 0;
    for (int cand : {3, 4, 5, 2, 1})
        if (nfrag % cand == 0) { nt = cand; break; }
    int wn = d->wn;
    if (wn == 0) {
        const int nb = nfrag / nt;
        wn = (nb % 4 == 0) ? 4 : (nb % 2 == 0) ? 2 : 1;
        // small-channel layers prefer more pixels per workgroup; keep wn <= 2 unless cout is wide
        if (wn == 4 && d->cout_pad < 192) wn = 2;
    }
    I2R_CHECK_ARG((wn == 1 || wn == 2 || wn == 4) && (nfrag / nt) % wn == 0, "i2r_conv: wn=%d does not divide cout", wn);
    const int wm = 4 / wn;

    // ---- tile ----
    int th = d->tile_h, tw = d->tile_w, mt = d->mt;
    if (th == 0 || tw == 0) {
        tw = d->conv_w <= 16 ? d->conv_w : (d->conv_w % 16 == 0 ? 16 : (d->conv_w % 12 == 0 ? 12 : 8));
        if (mt == 0) mt = 2;
        th = (wm * mt * 16) / tw;
        if (th < 1) th = 1;
        if (th > d->conv_h) th = d->conv_h;
    }
    if (mt == 0) mt = cdiv(th * tw, wm * 16);
    I2R_CHECK_ARG(mt >= 1 && mt <= 4 && th * tw <= wm * mt * 16, "i2r_conv: tile %dx%d does not fit wm=%d mt=%d", th, tw, wm, mt);

    ConvK k;
    k.in = d->in; k.in2 = d->in2; k.w = d->w; k.bias = d->bias; k.res1 = d->res1; k.res2 = d->res2; k.res_post = d->res_post; k.out = d->out;
    k.n_img = d->n_img; k.in_h = d->in_h; k.in_w = d->in_w; k.in_cs = d->in_cs; k.cin = d->cin;
    k.conv_h = d->conv_h; k.conv_w = d->conv_w; k.out_h = d->out_h; k.out_w = d->out_w; k.out_cs = d->out_cs;
    k.cout = d->cout; k.cout_pad = d->cout_pad; k.stride = d->stride; k.iy0 = d->iy0; k.ix0 = d->ix0;
    k.ntaps = d->ntaps;
    k.out_step = d->out_step; k.out_off_y = d->out_off_y; k.out_off_x = d->out_off_x; k.rep = d->rep; k.relu = d->relu;
    k.tile_h = th; k.tile_w = tw;
    k.tiles_y = cdiv(d->conv_h, th); k.tiles_x = cdiv(d->conv_w, tw);
    k.n_cblk = nfrag / (nt * wn);
    k.ph = (th - 1) * d->stride + max_dy + 1;
    k.pw = (tw - 1) * d->stride + max_dx + 1;
    I2R_CHECK_ARG(k.ph * k.pw <= kMaxPP * 256, "i2r_conv: patch %dx%d too large", k.ph, k.pw);
    k.plane = cdiv(k.ph * k.pw, 16) * 16;
    k.tap_kw = max_dx + 1;
    k.tap_kh = max_dy + 1;
    for (int t = 0; t < d->ntaps; ++t)
        I2R_CHECK_ARG(d->dy[t] == t / k.tap_kw && d->dx[t] == t % k.tap_kw && d->ntaps == (max_dy + 1) * (max_dx + 1),
                      "i2r_conv: taps must form a dense row-major kh x kw grid");
    int ck = d->ck;
    if (ck == 0) {
        const int budget = 72 * 1024;  // two workgroups per CU
        ck = d->cin;
        while (ck > 16 && (ck / 4) * k.plane * 16 > budget) {
            // largest multiple of 16 below that still divides evenly enough (any multiple works: tail handled)
            ck -= 16;
        }
    }
    I2R_CHECK_ARG(ck % 16 == 0 && ck >= 16, "i2r_conv: ck=%d", ck);
    k.ck = ck;
    const size_t lds_bytes = (size_t)(ck / 4) * k.plane * 16;
    I2R_CHECK_ARG(lds_bytes <= 160 * 1024, "i2r_conv: LDS %zu B", lds_bytes);

    conv_fn fn = pick_kernel(nt, wn, mt);
    I2R_CHECK_ARG(fn != nullptr, "i2r_conv: no kernel for nt=%d wn=%d mt=%d", nt, wn, mt);
    if (lds_bytes > 64 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    const long long nblk = (long long)d->n_img * k.tiles_y * k.tiles_x * k.n_cblk;
    I2R_CHECK_ARG(nblk > 0 && nblk < (1ll << 31), "i2r_conv: grid");
    hipLaunchKernelGGL(fn, dim3((unsigned)nblk), dim3(256), lds_bytes, (hipStream_t)stream, k);
    I2R_CHECK_LAUNCH("i2r_conv");
    return I2R_OK;
}
