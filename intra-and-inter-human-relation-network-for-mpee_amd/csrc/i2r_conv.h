// Shared pieces of the implicit-GEMM convolution kernels (internal): kernel-side descriptor, epilogue, activation load / store helpers.
// Included by i2r_conv.hip (fp32 matrix pipe + the launch logic) and by the 16-bit translation units (i2r_conv_lp.inc).
#pragma once
#include "i2r_common.h"

// Tuning hooks (ablation switches, phase stamps, env overrides) exist only in a -DI2R_TUNING build (tools/ scripts build one with
// __graft_entry__.build(defines=("I2R_TUNING",))); the product library has none of them: no env var changes what a kernel does.
#ifdef I2R_TUNING
#define I2R_DBG(p) ((p).dbg)
#else
#define I2R_DBG(p) 0
#endif

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
    int wn;             // waves along cout (1, 2, 4); waves along pixels = 4 / wn
    int dtype;          // 0 fp32 MFMA, 1 bf16, 2 f16 (fp32 accumulate)
    int in16, out16;    // 16-bit modes: activation storage of the input / of out + res1 + res2 + res_post (0 = fp32, 1 = 16-bit of `dtype`)
    int dbg;            // ablation switches (env I2R_CONV_DBG; tuning only): 1 no epilogue, 2 no staging loads, 4 no weight loads
    // index decode without integer divisions (a runtime division is ~35 vector-ALU instructions; a workgroup's prologue had a dozen):
    // ceil(2^32 / d) for d = n_cblk, tiles_x, tiles_y, pw, tile_w (div_m below); wn_log = log2(wn); npass = channel chunks per workgroup
    unsigned m_cblk, m_tx, m_ty, m_pw, m_tw;
    // byte sizes of in (= in2), the packed weights and out (= res1 / res2 / res_post) for the buffer descriptors of the kernels: global
    // accesses are buffer instructions with 32-bit lane offsets, a lane at kOOB reads zeros and its stores are dropped
    unsigned in_bytes, w_bytes, out_bytes;
    int wn_log, npass;
    // Winograd F(2x2, 3x3) kernels (i2r_conv_wino.hip; algo == 1): tiles_y / tiles_x count FRAGMENTS (16 Winograd tiles, 2^w_fwlog across)
    // per crop, ph / pw / plane describe one fragment's raw patch, whose rows have w_pitch slots with the odd columns at + w_half
    int algo, w_fwlog, w_pitch, w_half, w_nfrag, w_rcp;  // w_rcp = ceil(65536 / pw): patch row of a pixel index without a division
    int w_band;                                          // fragment groups per XCD band (launches without a dispatch table, i2r_conv_wino.hip)
    unsigned w_m_cblk, w_m_img, w_m_tx;                  // ceil(2^32 / d) for d = n_cblk, fragments per crop, fragments per row (item decode)
};

// n / d through m = ceil(2^32 / d): exact while n * (m * d - 2^32) < 2^32, i.e. for every n < 2^32 / d (prepare() checks the ranges);
// d == 1 has no 32-bit reciprocal
__device__ __forceinline__ int div_m(int n, int d, unsigned m) { return d == 1 ? n : (int)__umulhi((unsigned)n, m); }

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
// lane offset beyond every buffer the conv kernels describe (tensors are < 2 GiB: checked by the host) that stays there when a chunk /
// piece offset of a few KiB is added
constexpr unsigned kOOB = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, base ? bytes : 0, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_ld16(const __amdgpu_buffer_rsrc_t& rs, unsigned voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
}
// 16-byte store with a SCALAR offset.  A buffer_store_dwordx4 reads its data registers a little after it issues; the compiler keeps
// two wait states before a VALU write of those registers only when the store has no SGPR soffset and lets the very next instruction
// overwrite them when it has one.  On MI355X that lost data whenever other waves shared the SIMD (conv1x1_pair_k beside a second
// program: 16 lanes of a fragment stored the NEXT fragment's value; tools/race_bisect.py, DESIGN.md "store-data hazard").  The asm
// keeps the data registers alive for two more wait states; tools/isa_store_hazard.py (tests/test_host.py) scans every kernel of
// the built library for stores left with fewer.
__device__ __forceinline__ void buf_st16(const __amdgpu_buffer_rsrc_t& rs, unsigned voff, int soff, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, voff, soff, 0);
    asm volatile("s_nop 1" ::"v"(v));
}

constexpr int kMaxPP = 5;  // patch pixels per thread (256 threads) -> patches up to 1280 pixels

constexpr int kMaxGroups = 4;
struct ConvGroupK {
    ConvK g[kMaxGroups];
    int blk_end[kMaxGroups];  // exclusive prefix sums of workgroups per group
    int n;
    const int* blk_map;       // optional dispatch-order table: entry = (group << 24) | workgroup index within the group
};

// 4 consecutive channels of an activation tensor at ELEMENT offset `off`: fp32 (16 bytes) or, in the 16-bit kernels when h16 is set,
// bf16 / f16 storage (8 bytes; BASELINE configs 3-5 keep the activations of the conv towers in 16 bit: half the HBM traffic)
template <int DT>
__device__ __forceinline__ f32x4 ld_act4(const float* base, size_t off, bool h16) {
    if constexpr (DT != 0) {
        if (h16) {
            const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + off);
            if constexpr (DT == 1)
                return (f32x4){__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xFFFF0000u)};
            else {
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                const h4 h = __builtin_bit_cast(h4, u);
                return (f32x4){(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
            }
        }
    }
    return *reinterpret_cast<const f32x4*>(base + off);
}
template <int DT>
__device__ __forceinline__ void st_act4(float* base, size_t off, f32x4 v, bool h16) {
    if constexpr (DT != 0) {
        if (h16) {
            uint2 u;
            if constexpr (DT == 1) {
                typedef __bf16 b4 __attribute__((ext_vector_type(4)));
                const b4 b = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                u = __builtin_bit_cast(uint2, b);
            } else {
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                const h4 h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                u = __builtin_bit_cast(uint2, h);
            }
            *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(base) + off) = u;
            return;
        }
    }
    *reinterpret_cast<f32x4*>(base + off) = v;
}

// the same through a buffer descriptor at BYTE offset voff (a lane at kOOB: loads return zeros, stores are dropped)
template <int DT>
__device__ __forceinline__ f32x4 buf_ld_act4(const __amdgpu_buffer_rsrc_t& rs, unsigned voff, bool h16) {
    if constexpr (DT != 0) {
        if (h16) {
            const u32x2 u = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, 0, 0);
            if constexpr (DT == 1)
                return (f32x4){__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xFFFF0000u)};
            else {
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                const h4 h = __builtin_bit_cast(h4, u);
                return (f32x4){(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
            }
        }
    }
    return buf_ld16(rs, voff, 0);
}
template <int DT>
__device__ __forceinline__ void buf_st_act4(const __amdgpu_buffer_rsrc_t& rs, unsigned voff, f32x4 v, bool h16) {
    if constexpr (DT != 0) {
        if (h16) {
            u32x2 u;
            if constexpr (DT == 1) {
                typedef __bf16 b4 __attribute__((ext_vector_type(4)));
                const b4 b = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                u = __builtin_bit_cast(u32x2, b);
            } else {
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                const h4 h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                u = __builtin_bit_cast(u32x2, h);
            }
            __builtin_amdgcn_raw_buffer_store_b64(u, rs, voff, 0, 0);
            return;
        }
    }
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, voff, 0, 0);
}

// ---- epilogue (shared by the fp32 and the bf16/f16 MFMA bodies: the C/D register layout is dtype independent) ----
// D layout: lane (li = l&15, g) holds channel n = nt*16 + li of pixels 4g + r (r = 0..3).  A 4x4 transpose inside each
// lane quad (two DPP butterfly stages, no LDS) turns that into: lane (q = li>>2, j = li&3, g) holds channels
// nt*16 + 4q .. +3 of pixel 4g + j, so bias / residual / store are 16-byte accesses (4x fewer VMEM instructions;
// the scalar-store epilogue measured 24 % of the kernel).  A second step swaps the roles of q and j across the 16 lanes of a
// row (ds_bpermute, lane 4j+q <- lane 4q+j): then the four CONSECUTIVE lanes of a quad hold the four 16-byte pieces of ONE
// pixel's 64 contiguous bytes.  The texture addresser retires a 64-lane 16-byte access in 16 cycles only when every quad
// falls into one 64-byte segment, 64 cycles otherwise (tools/probe/load_pattern.hip) -- with residual loads that was
// 2 x MT x NT slow accesses per wave.
template <int MT, int NT, int DT>
__device__ __forceinline__ void conv_epilogue(const ConvK& p, f32x4 (&acc)[MT][NT], int img, int oy0, int ox0, int wm, int n_base,
                                              int li, int g, int tile_px) {
    if ((I2R_DBG(p) & 1) && acc[0][0][0] != 12345.678f) return;
    const int j4 = li & 3;
    const int pj = li >> 2, pq = li & 3;  // after the lane permutation: this lane's pixel (4g + pj) and 16-byte piece (pq)
    const int perm_src = (g * 16 + pq * 4 + pj) * 4;  // ds_bpermute byte address of the lane holding (q = pq, j = pj)
    auto to_pixel_major = [&](f32x4 v) {
        v = quad_transpose(v, j4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x = v[e];  // (scalar copy: __builtin_bit_cast on a vector element lvalue miscompiles)
            v[e] = __int_as_float(__builtin_amdgcn_ds_bpermute(perm_src, __float_as_int(x)));
        }
        return v;
    };
    const bool h16 = DT != 0 && p.out16;
    const unsigned esz = h16 ? 2u : 4u;
    // destination / residual tensors through buffer descriptors: pixels outside the map and channel pieces beyond the row get the
    // out-of-range offset -- their loads return zeros, their stores are dropped -- so nothing below branches on validity
    const __amdgpu_buffer_rsrc_t rs_out = make_rsrc(p.out, p.out_bytes), rs_r1 = make_rsrc(p.res1, p.out_bytes),
                                 rs_r2 = make_rsrc(p.res2, p.out_bytes), rs_rp = make_rsrc(p.res_post, p.out_bytes);
    auto finish = [&](f32x4 t, int n, bool full, unsigned voff) {
        if (p.relu == 1) {
            t[0] = fmaxf(t[0], 0.f); t[1] = fmaxf(t[1], 0.f); t[2] = fmaxf(t[2], 0.f); t[3] = fmaxf(t[3], 0.f);
        } else if (p.relu == 2) {  // exact-erf GELU (HRFormer MlpDWBN, hrformer.py:1197)
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = 0.5f * t[e] * (1.f + erff(t[e] * 0.70710678118654752f));
        }
        if (p.res_post) t += buf_ld_act4<DT>(rs_rp, voff, h16);
        if (!full) {  // channels >= cout are padding: keep them exactly zero
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (n + e >= p.cout) t[e] = 0.f;
        }
        return t;
    };
    // per-lane channel piece of every N fragment: bias fetched once, up front
    f32x4 bias[NT];
    unsigned noff[NT];  // byte offset of the piece inside a pixel's row, kOOB when the piece does not exist in the destination
    const bool tail = n_base + NT * 16 > p.cout || n_base + NT * 16 > p.out_cs;  // (wave-uniform)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = n_base + nt * 16 + pq * 4;
        const bool nok = n < p.cout_pad && (n + 4 <= p.cout || n + 4 <= p.out_cs);
        noff[nt] = nok ? (unsigned)n * esz : kOOB;
        bias[nt] = n < p.cout_pad ? *reinterpret_cast<const f32x4*>(p.bias + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    if (p.rep == 1 && !(I2R_DBG(p) & 16)) {
        // ---- every output pixel written once: issue ALL residual loads of the tile first (one memory latency instead of
        //      MT x NT dependent load -> add -> store round trips), then transform and store ----
        unsigned off[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = (wm * MT + mt) * 16 + g * 4 + pj;
            const int ty = div_m(m, p.tile_w, p.m_tw), tx = m - ty * p.tile_w;
            const int oy = oy0 + ty, ox = ox0 + tx;
            const bool pv = m < tile_px && oy < p.conv_h && ox < p.conv_w;
            off[mt] = pv ? (unsigned)(((img * p.out_h + oy * p.out_step + p.out_off_y) * p.out_w + ox * p.out_step + p.out_off_x) * p.out_cs) * esz : kOOB;
        }
        // byte offset of piece (mt, nt): pixel + piece; only a TAIL block (uniform: the wave's channels reach past cout / the row) has
        // pieces that do not exist -- elsewhere the plain sum is right, and a pixel at kOOB stays out of range with a piece offset added
        auto at = [&](int mt, int nt) { return (tail && noff[nt] == kOOB) ? kOOB : off[mt] + noff[nt]; };
        f32x4 r[MT][NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                r[mt][nt] = bias[nt];
                if (p.res1) r[mt][nt] += buf_ld_act4<DT>(rs_r1, at(mt, nt), h16);
                if (p.res2) r[mt][nt] += buf_ld_act4<DT>(rs_r2, at(mt, nt), h16);
            }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const f32x4 v = to_pixel_major(acc[mt][nt]);
                const int n = n_base + nt * 16 + pq * 4;
                const unsigned o = at(mt, nt);
                buf_st_act4<DT>(rs_out, o, finish(v + r[mt][nt], n, !tail || n + 4 <= p.cout, o), h16);
            }
        return;
    }
    // ---- nearest-neighbour upsample scatter (HRNet fuse layers): rep x rep destinations per conv pixel ----
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = (wm * MT + mt) * 16 + g * 4 + pj;
        const int ty = div_m(m, p.tile_w, p.m_tw), tx = m - ty * p.tile_w;
        const int oy = oy0 + ty, ox = ox0 + tx;
        const bool pvalid = m < tile_px && oy < p.conv_h && ox < p.conv_w;
        const int by = oy * p.out_step + p.out_off_y, bx = ox * p.out_step + p.out_off_x;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const f32x4 v = to_pixel_major(acc[mt][nt]) + bias[nt];
            const int n = n_base + nt * 16 + pq * 4;
            if (!pvalid || noff[nt] == kOOB) continue;
            for (int ry = 0; ry < p.rep; ++ry)
                for (int rx = 0; rx < p.rep; ++rx) {
                    const unsigned o = (unsigned)(((img * p.out_h + by + ry) * p.out_w + bx + rx) * p.out_cs) * esz + noff[nt];
                    f32x4 t = v;
                    if (p.res1) t += buf_ld_act4<DT>(rs_r1, o, h16);
                    if (p.res2) t += buf_ld_act4<DT>(rs_r2, o, h16);
                    buf_st_act4<DT>(rs_out, o, finish(t, n, n + 4 <= p.cout, o), h16);
                }
        }
    }
}



typedef void (*conv_fn)(const ConvGroupK);

}  // namespace

// kernel pickers of the 16-bit translation units (one per operand type, so the three conv files compile in parallel); null = no such variant
void* i2r_pick_conv_bf16(int nt, int mt, int cap, int pf);
void* i2r_pick_conv_f16(int nt, int mt, int cap, int pf);
// Winograd F(2x2, 3x3) fp32 kernels (i2r_conv_wino.hip): MT fragments x NT channel fragments per workgroup; LDS bytes of a workgroup
void* i2r_pick_conv_wino(int nt, int mt);
size_t i2r_conv_wino_lds(int nt, int mt, int plane);
