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
#include <stdlib.h>

#include <algorithm>

#include "i2r_conv.h"

namespace {

template <int MT, int NT, int CAP, int PF>
__device__ __forceinline__ void conv_body(const ConvK& p, int bid, f32x4* lds) {
    const int tid = threadIdx.x;
    // tuning aid (I2R_CONV_DBG & 8): per-workgroup phase time stamps (s_memtime) into the buffer passed as res2
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;
    const bool stamp = (I2R_DBG(p) & 8) != 0;
    const int bid_stamp = bid;
    if (stamp) ts0 = __builtin_amdgcn_s_memtime();
    const int lane = tid & 63, wave = tid >> 6;
    const int WN = p.wn;
    const int wm = wave >> p.wn_log, wn = wave & (WN - 1);
    const int li = lane & 15, g = lane >> 4;

    int q = div_m(bid, p.n_cblk, p.m_cblk);  // (host-made reciprocals: no integer division in the prologue)
    const int cb = bid - q * p.n_cblk;
    bid = q;
    q = div_m(bid, p.tiles_x, p.m_tx);
    const int tile_x = bid - q * p.tiles_x;
    bid = q;
    const int img = div_m(bid, p.tiles_y, p.m_ty);
    const int tile_y = bid - img * p.tiles_y;
    const int oy0 = tile_y * p.tile_h, ox0 = tile_x * p.tile_w;
    const int py0 = oy0 * p.stride + p.iy0, px0 = ox0 * p.stride + p.ix0;
    const int phw = p.ph * p.pw;

    // Staging assignment: thread t handles the patch pixels (t >> 2) + 64 j and, of every 4 consecutive channel groups, group
    // (t & 3) -- the four lanes of a quad fetch the 64 contiguous bytes of ONE pixel.  (With one pixel per lane every lane of
    // a load hits a different 64-byte segment and the texture addresser needs 64 instead of 16 cycles per instruction:
    // tools/probe/load_pattern.hip.)
    // patch pixel -> global float offset of channel 0 (clamped to 0 outside the image, with a validity bit);
    // channel independent, so computed once
    // (the synchronous variant PF == 0 serves patches of up to 1280 pixels: it keeps one pixel per lane, 20 offsets per thread
    // would cost a wave of occupancy)
    // (global accesses are buffer instructions: 32-bit lane byte offsets + scalar chunk offsets, no 64-bit address arithmetic on the
    //  vector ALU; a lane at kOOB -- halo outside the image, unused slots -- reads zeros without a select per load)
    constexpr int NPP = PF > 0 ? 4 * PF : kMaxPP;
    const int ul = tid & 3;
    const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(p.in, p.in_bytes), rs_in2 = make_rsrc(p.in2, p.in_bytes), rs_w = make_rsrc(p.w, p.w_bytes);
    unsigned goff[NPP];  // byte offset of the patch pixel's channel 0, kOOB outside the image
#pragma unroll
    for (int j = 0; j < NPP; ++j) {
        const int pp = PF > 0 ? (tid >> 2) + j * 64 : tid + j * 256;
        goff[j] = kOOB;
        if (pp < phw) {
            const int py = div_m(pp, p.pw, p.m_pw), px = pp - py * p.pw;
            const int iy = py0 + py, ix = px0 + px;
            if (iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w) goff[j] = (unsigned)(((img * p.in_h + iy) * p.in_w + ix) * p.in_cs) * 4u;
        }
    }
    const int npp = (phw + 255) >> 8;  // PF == 0: patch pixels per thread actually in use (wave-uniform)

    // A-fragment patch-pixel base of this lane's pixel in each M fragment
    const int tile_px = p.tile_h * p.tile_w;
    int ppix[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int m = (wm * MT + mt) * 16 + li;
        if (m >= tile_px) m = 0;  // padded rows compute garbage-free duplicates; masked at the store
        const int ty = div_m(m, p.tile_w, p.m_tw), tx = m - ty * p.tile_w;
        ppix[mt] = ty * p.stride * p.pw + tx * p.stride;
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int n_base = (cb * WN + wn) * NT * 16;
    const int cin4 = p.cin >> 2;
    // lane's weight offset: float4 index ((tap*cin4 + cg) * cout_pad + n): the lane part (n, g) is constant, the rest walks on the scalar ALU
    const unsigned wlane = (unsigned)(n_base + li + g * p.cout_pad) * 16u;

    // ---- one channel chunk: K loop over (tap, 16-channel step) reading the staged patch `buf`, software-pipelined
    //      one step ahead with two named register sets (no register copies, so the compiler's vmcnt/lgkmcnt waits land
    //      at the first use); `mid` runs right after the first operand fetch (used to launch the next chunk's loads) ----
    auto run_chunk = [&](const f32x4* buf, int c0, int ckc, auto&& mid) {
        const int ncs = ckc >> 4;
        const int nit = p.ntaps * ncs;
        const int wp0 = (c0 >> 2) * p.cout_pad * 16;  // step (tap 0, cs 0) of this chunk (scalar byte offset)
        int wp = wp0;
        const int inc_cs = 4 * p.cout_pad * 16;
        const int inc_tap = (cin4 - (ncs - 1) * 4) * p.cout_pad * 16;
        int cs_n = 0, tx_n = 0, ty_n = 0;  // position of the next step to fetch
        auto fetch = [&](f32x4(&a)[MT], f32x4(&b)[NT]) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b[nt] = buf_ld16(rs_w, wlane + nt * 256u, (I2R_DBG(p) & 4) ? wp0 : wp);
            const int abase = (cs_n * 4 + g) * p.plane + ty_n * p.pw + tx_n;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[mt] = buf[abase + ppix[mt]];
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
        mid();
        int it = 0;
        for (; it + 2 <= nit; it += 2) {
            fetch(a1, b1);
            fma_step(a0, b0);
            fetch(a0, b0);  // unconditional (wraps after the last step) so both halves keep counted waits
            fma_step(a1, b1);
        }
        if (it < nit) fma_step(a0, b0);
    };

    if constexpr (PF > 0) {
        // ---- double-buffered staging: chunks of 4*ckg channels (ckg <= CAP); the NEXT chunk's patch is loaded into
        //      registers while the MFMAs of the current chunk run, and written to the other LDS buffer afterwards
        //      (one barrier per chunk) ----
        constexpr int UQ = CAP / 4, PJ = 4 * PF;
        f32x4 v[UQ][PJ];
        const int ckg = p.ck >> 2;
        auto stage_load = [&](int c0) {
#pragma unroll
            for (int uq = 0; uq < UQ; ++uq)
                if (uq * 4 < ckg) {
#pragma unroll
                    for (int j = 0; j < PJ; ++j)
                        // (predicated on the pixel lying inside the image: halo pixels outside cost no memory access -- and the
                        //  exec-masked loads keep this block of loads together: with unconditional loads the same kernel measured 8 % slower)
                        v[uq][j] = (I2R_DBG(p) & 2) ? (f32x4){0.f, 0.f, 0.f, 0.f} : buf_ld16(rs_in, goff[j] + (uq * 4 + ul) * 16u, c0 * 4);
                }
            if (p.in2) {
#pragma unroll
                for (int uq = 0; uq < UQ; ++uq)
                    if (uq * 4 < ckg) {
#pragma unroll
                        for (int j = 0; j < PJ; ++j)
                            v[uq][j] += buf_ld16(rs_in2, goff[j] + (uq * 4 + ul) * 16u, c0 * 4);
                    }
            }
        };
        auto stage_store = [&](f32x4* buf) {
#pragma unroll
            for (int uq = 0; uq < UQ; ++uq)
                if (uq * 4 < ckg) {
#pragma unroll
                    for (int j = 0; j < PJ; ++j) {
                        const int pp = (tid >> 2) + j * 64;
                        if (pp < phw) buf[(uq * 4 + ul) * p.plane + pp] = v[uq][j];
                    }
                }
        };
        const int npass = p.npass;
        stage_load(0);
        stage_store(lds);
        __syncthreads();
        if (stamp) ts1 = __builtin_amdgcn_s_memtime();
        for (int pass = 0; pass < npass; ++pass) {
            f32x4* cur = lds + (pass & 1) * ckg * p.plane;
            f32x4* nxt = lds + ((pass + 1) & 1) * ckg * p.plane;
            const bool more = pass + 1 < npass;
            run_chunk(cur, pass * p.ck, p.ck, [&]() { if (more) stage_load((pass + 1) * p.ck); });
            if (more) stage_store(nxt);
            __syncthreads();
        }
    } else {
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
                        v[u][j] = (j < npp && !(I2R_DBG(p) & 2)) ? buf_ld16(rs_in, goff[j] + u * 16u, (c0 + cg0 * 4) * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
                if (p.in2) {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int j = 0; j < kMaxPP; ++j)
                            if (j < npp) v[u][j] += buf_ld16(rs_in2, goff[j] + u * 16u, (c0 + cg0 * 4) * 4);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < kMaxPP; ++j) {
                        const int pp = tid + j * 256;
                        if (j < npp && pp < phw)
                            lds[(cg0 + u) * p.plane + pp] = v[u][j];
                    }
            }
            __syncthreads();
            run_chunk(lds, c0, ckc, []() {});
        }
    }

    if (stamp) {
        ts2 = __builtin_amdgcn_s_memtime();
        ConvK q = p;
        q.res2 = nullptr;
        conv_epilogue<MT, NT, 0>(q, acc, img, oy0, ox0, wm, n_base, li, g, tile_px);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long ts3 = __builtin_amdgcn_s_memtime();
        if (tid == 0) {
            unsigned long long* o = reinterpret_cast<unsigned long long*>(const_cast<float*>(p.res2)) + (size_t)bid_stamp * 4;
            o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = ts3;
            if (I2R_DBG(p) & 32) {  // where did the hardware place this workgroup?  HW_ID (CU / SE / SIMD fields) | XCC_ID << 32
                const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
                const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));
                o[1] = ((unsigned long long)xcc << 32) | hw;
            }
        }
        return;
    }
    conv_epilogue<MT, NT, 0>(p, acc, img, oy0, ox0, wm, n_base, li, g, tile_px);
}

// One launch = up to kMaxGroups independent convolutions that share (MT, NT): "horizontal fusion" of the
// HRNet branches / fuse terms so that the small low-resolution convs fill the chip together with the large one.
template <int MT, int NT, int CAP, int PF>
__global__ __launch_bounds__(256) void conv_igemm_f32(const ConvGroupK grp) {
    extern __shared__ __attribute__((aligned(16))) f32x4 lds[];
    int bid = blockIdx.x, gi = 0, start = 0;
    if (grp.blk_map) {
        // host-chosen dispatch order (longest-processing-time packing over the CUs): members of a group have very
        // different K, and with every workgroup resident from the start the hardware cannot rebalance later
        const int v = __builtin_amdgcn_readfirstlane(grp.blk_map[bid]);  // (uniform: keeps the member's fields and buffer descriptors in SGPRs)
        gi = v >> 24;
        bid = v & 0xFFFFFF;
    } else {
#pragma unroll
        for (int i = 0; i < kMaxGroups - 1; ++i)
            if (i + 1 < grp.n && bid >= grp.blk_end[i]) { gi = i + 1; start = grp.blk_end[i]; }
    }
    conv_body<MT, NT, CAP, PF>(grp.g[gi], bid - start, lds);
}

// ---- persistent "chain" kernel: L dependent layers x G independent members in ONE launch, tile-level dataflow instead of a
//      chip-wide barrier (kernel boundary) per layer.  The persistent workgroups of an XCD (index % 8) pop work items
//      (layer, member, tile workgroup) from that XCD's queue, which the host sorted by layer; an item waits until the tiles of
//      the previous layer that its input patch touches (the 3x3 tile neighbourhood, all cout blocks) have signalled completion.
//      All items of one image sit in one queue, so producer and consumer meet in the same L2.  A waited-for item precedes the
//      waiter in the queue, i.e. it has been popped by a RUNNING workgroup (the grid never exceeds the resident capacity
//      i2r_conv_chain_pack reports): no deadlock. ----
struct ChainK {
    const ConvK* k;        // device [n_layers * n_members], layer-major
    const int* fbase;      // device [n_layers * n_members]: first completion counter of (layer, member)
    const int* item_ofs;   // device [9]: the item range of each XCD's queue
    const int* items;      // device: (layer << 26) | (member << 24) | workgroup index within the member
    int* flags;            // device completion counters (zeroed before the launch) + error word at [n_flags] + 8 XCC check words + 8 queue heads
    int n_layers, n_members, n_flags;
};

template <int MT, int NT, int CAP, int PF>
__global__ __launch_bounds__(256) void conv_chain_f32(const ChainK c) {
    extern __shared__ __attribute__((aligned(16))) f32x4 lds[];
    const int tid = threadIdx.x;
    __shared__ int s_next;
    const int xq = blockIdx.x & 7;  // one work queue per XCD (all items of an image live in one queue), popped with an atomic counter
    const int it0 = c.item_ofs[xq], it1 = c.item_ofs[xq + 1];
    int* const qctr = c.flags + c.n_flags + 9 + xq;
    // the schedule relies on workgroup b running on XCD b % 8 (round-robin dispatch): verify, flag a violation in the error word
    if (tid == 0) {  // (words [n_flags + 1 .. + 8]: XCC id + 1 seen by residue class b % 8; all members of a class must agree)
        const int xcc = (__builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | ((4 - 1) << 11)) & 15) + 1;
        const int old = atomicCAS(c.flags + c.n_flags + 1 + (blockIdx.x & 7), 0, xcc);
        if (old != 0 && old != xcc) c.flags[c.n_flags] = 2;
    }
    for (;;) {
        __syncthreads();  // the previous item's LDS reads (and s_next) are over
        if (tid == 0) s_next = it0 + atomicAdd(qctr, 1);
        __syncthreads();
        const int it = s_next;
        if (it >= it1) break;
        const int v = c.items[it];
        const int l = v >> 26, g = (v >> 24) & 3, bid = v & 0xFFFFFF;
        const ConvK k = c.k[l * c.n_members + g];
        // tile coordinates exactly as conv_body decodes them
        int b = bid / k.n_cblk;
        const int tile_x = b % k.tiles_x;
        b /= k.tiles_x;
        const int tile_y = b % k.tiles_y;
        const int img = b / k.tiles_y;
        if (l > 0) {
            if (tid < 9) {
                const int ny = tile_y + tid / 3 - 1, nx = tile_x + tid % 3 - 1;
                if (ny >= 0 && ny < k.tiles_y && nx >= 0 && nx < k.tiles_x) {
                    int* f = c.flags + c.fbase[(l - 1) * c.n_members + g] + (img * k.tiles_y + ny) * k.tiles_x + nx;
                    int spins = 0;
                    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < k.n_cblk) {  // (served by the L2)
                        __builtin_amdgcn_s_sleep(2);
                        if (++spins > (1 << 24)) {  // (~seconds: a scheduling bug, not a slow producer) give up loudly, do not hang the GPU
                            c.flags[c.n_flags] = 1;
                            break;
                        }
                    }
                }
            }
            __syncthreads();
            // Producer and consumer share one XCD, hence one L2: the data is coherent there and no L2 write-back is needed (a full
            // agent-scope release + acquire per item made the chain 2x slower than per-layer launches).  This CU's L1 may still
            // hold lines of the recycled activation buffers, though, and only the agent-scope invalidate drops them:
            // "buffer_inv sc0" is a no-op outside threadgroup-split mode (measured: intermittently stale patches with it).
            asm volatile("buffer_inv sc1" ::: "memory");
        }
        conv_body<MT, NT, CAP, PF>(k, bid, lds);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores have reached the L2 (the L1 is write-through)
        __syncthreads();
        if (tid == 0)
            __hip_atomic_fetch_add(c.flags + c.fbase[l * c.n_members + g] + (img * k.tiles_y + tile_y) * k.tiles_x + tile_x, 1,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}


template <int NT, int CAP, int PF>
conv_fn pick_mt(int mt) {
    switch (mt) {
        case 1: return conv_igemm_f32<1, NT, CAP, PF>;
        case 2: return conv_igemm_f32<2, NT, CAP, PF>;
        case 3: return conv_igemm_f32<3, NT, CAP, PF>;
        case 4: return conv_igemm_f32<4, NT, CAP, PF>;
    }
    return nullptr;
}

template <int CAP, int PF>
conv_fn pick_nt(int nt, int mt) {
    switch (nt) {
        case 3: return pick_mt<3, CAP, PF>(mt);
        case 4: return pick_mt<4, CAP, PF>(mt);
        case 5: return pick_mt<5, CAP, PF>(mt);
    }
    return nullptr;
}

// staging variant: pf 0 = synchronous staging (any patch size / ck); pf 1|2 = double-buffered chunks for patches of up to
// 256|512 pixels with up to cap (4 or 12) channel groups of prefetch registers
conv_fn pick_kernel(int nt, int mt, int cap, int pf) {
    if (pf == 0) return pick_nt<4, 0>(nt, mt);
    if (pf == 1 && cap == 4) return pick_nt<4, 1>(nt, mt);
    if (pf == 1 && cap == 12) return pick_nt<12, 1>(nt, mt);
    if (pf == 2 && cap == 4) return pick_nt<4, 2>(nt, mt);
    return nullptr;
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

}  // namespace

static void copy_desc(const i2r_conv_desc* d, ConvK& k) {
    k.in = d->in; k.in2 = d->in2; k.w = d->w; k.bias = d->bias; k.res1 = d->res1; k.res2 = d->res2; k.res_post = d->res_post; k.out = d->out;
    k.n_img = d->n_img; k.in_h = d->in_h; k.in_w = d->in_w; k.in_cs = d->in_cs; k.cin = d->cin;
    k.conv_h = d->conv_h; k.conv_w = d->conv_w; k.out_h = d->out_h; k.out_w = d->out_w; k.out_cs = d->out_cs;
    k.cout = d->cout; k.cout_pad = d->cout_pad; k.stride = d->stride; k.iy0 = d->iy0; k.ix0 = d->ix0;
    k.ntaps = d->ntaps;
    k.out_step = d->out_step; k.out_off_y = d->out_off_y; k.out_off_x = d->out_off_x; k.rep = d->rep; k.relu = d->relu;
    k.dtype = d->dtype;
    k.in16 = d->in_f16; k.out16 = d->out_f16;
    k.algo = d->algo; k.w_fwlog = k.w_pitch = k.w_half = k.w_nfrag = k.w_rcp = k.w_band = 0;
    k.w_m_cblk = k.w_m_img = k.w_m_tx = 0; k.in_bytes = k.w_bytes = k.out_bytes = 0;
    k.m_cblk = k.m_tx = k.m_ty = k.m_pw = k.m_tw = 0; k.wn_log = 0; k.npass = 0;
    k.dbg = 0;
}

// Winograd F(2x2, 3x3) launch geometry (i2r_conv_wino.hip): fragment shape, patch layout in LDS, workgroup count
static int prepare_wino(const i2r_conv_desc* d, int force_mt, ConvK& k, int* nt_out, int* mt_out, int* cap_out, int* pf_out, size_t* lds_out,
                        long long* nblk_out) {
    I2R_CHECK_ARG(d->dtype == 0 && !d->in_f16 && !d->out_f16, "i2r_conv: the Winograd kernels are fp32");
    I2R_CHECK_ARG(d->stride == 1 && d->ntaps == 9 && d->iy0 == -1 && d->ix0 == -1 && d->rep == 1 && d->out_step == 1 && d->out_off_y == 0 &&
                      d->out_off_x == 0 && d->in2 == nullptr,
                  "i2r_conv: algo 1 (Winograd) needs a plain 3x3 stride-1 pad-1 convolution");
    for (int t = 0; t < 9; ++t) I2R_CHECK_ARG(d->dy[t] == t / 3 && d->dx[t] == t % 3, "i2r_conv: algo 1 needs row-major 3x3 taps");
    I2R_CHECK_ARG(d->relu == 0 || d->relu == 1, "i2r_conv: algo 1 has no GELU epilogue");
    I2R_CHECK_ARG((long long)d->n_img * d->in_h * d->in_w * d->in_cs < (1ll << 29) && (long long)d->n_img * d->out_h * d->out_w * d->out_cs < (1ll << 29) &&
                      (long long)16 * d->cin * d->cout_pad < (1ll << 29),
                  "i2r_conv: algo 1 addresses its tensors through buffer descriptors of < 2 GiB");
    I2R_CHECK_ARG(d->conv_h == d->in_h && d->conv_w == d->in_w && d->out_h == d->conv_h && d->out_w == d->conv_w, "i2r_conv: algo 1 geometry");
    const int nfrag = d->cout_pad / 16;
    const int nt = nfrag % 3 == 0 ? 3 : (nfrag % 4 == 0 ? 4 : 0);
    I2R_CHECK_ARG(nt != 0, "i2r_conv: algo 1 needs cout_pad=%d to be a multiple of 48 or 64", d->cout_pad);
    const int mt = force_mt ? force_mt : (d->mt ? d->mt : 1);  // (one fragment per item = 4 waves per SIMD: the measured optimum, and the only form NT = 4 is built for)
    I2R_CHECK_ARG(mt == 1 || mt == 2, "i2r_conv: algo 1 takes mt 1 or 2 (got %d)", mt);
    int fw = 0;  // Winograd tiles across a fragment of 16 (64 output pixels): the shape that covers the map with the fewest fragments
    if (d->tile_w) {
        I2R_CHECK_ARG(d->tile_w == 16 || d->tile_w == 8 || d->tile_w == 4, "i2r_conv: algo 1 fragment width %d (16, 8, 4)", d->tile_w);
        fw = d->tile_w / 2;
    } else {
        long long best = -1;
        for (int cand : {8, 4, 2}) {
            const long long n = (long long)cdiv(d->conv_w, 2 * cand) * cdiv(d->conv_h, 32 / cand);
            if (best < 0 || n < best) { best = n; fw = cand; }
        }
    }
    copy_desc(d, k);
    k.w_fwlog = fw == 8 ? 3 : (fw == 4 ? 2 : 1);
    k.tile_w = 2 * fw; k.tile_h = 32 / fw;
    k.tiles_x = cdiv(d->conv_w, k.tile_w); k.tiles_y = cdiv(d->conv_h, k.tile_h);
    k.n_cblk = nfrag / nt;
    k.ph = k.tile_h + 2; k.pw = k.tile_w + 2;
    // row pitch (16-byte slots): odd columns sit at + half; 2 * pitch = 8 / 4 / 2 (mod 16) for fw = 8 / 4 / 2, so the 16 tiles of a
    // fragment (fw across) gather 16 distinct slots modulo 16 for every (row, column) of their 4x4 patches
    k.w_half = fw + 1;
    k.w_pitch = fw == 8 ? 20 : (fw == 4 ? 10 : 9);
    k.plane = cdiv(k.ph * k.w_pitch, 16) * 16;
    k.w_rcp = (65536 + k.pw - 1) / k.pw;  // exact for pixel indices < 256 and pw in {6, 10, 18}
    k.in_bytes = (unsigned)((long long)d->n_img * d->in_h * d->in_w * d->in_cs * 4);
    k.w_bytes = (unsigned)((long long)16 * d->cin * d->cout_pad * 4);
    k.out_bytes = (unsigned)((long long)d->n_img * d->out_h * d->out_w * d->out_cs * 4);
    // item decode without integer divisions: n / d = mulhi(n, ceil(2^32 / d)) is exact for n < 2^20 and d < 2^11
    auto magic = [](int d) { return d == 1 ? 0u : (unsigned)(((1ull << 32) + d - 1) / d); };
    k.w_m_cblk = magic(k.n_cblk); k.w_m_img = magic(k.tiles_y * k.tiles_x); k.w_m_tx = magic(k.tiles_x);
    k.tap_kw = k.tap_kh = 3;
    k.ck = 16; k.wn = 1;
    const long long nf = (long long)d->n_img * k.tiles_y * k.tiles_x;
    I2R_CHECK_ARG(nf > 0 && nf * k.n_cblk < (1ll << 20) && k.tiles_y * k.tiles_x < 2048 && k.n_cblk < 2048, "i2r_conv: algo 1 grid (%lld fragments)", nf);
    k.w_nfrag = (int)nf;
#ifdef I2R_TUNING
    {
        static const int dbg = getenv("I2R_CONV_DBG") ? atoi(getenv("I2R_CONV_DBG")) : 0;
        k.dbg = dbg;
    }
#endif
    *nt_out = nt; *mt_out = mt; *cap_out = 0; *pf_out = 100;
    *lds_out = i2r_conv_wino_lds(nt, mt, k.plane);
    *nblk_out = (nf + mt - 1) / mt * k.n_cblk;
    return I2R_OK;
}

// validate one descriptor and derive its launch geometry; *nt_out/*mt_out: fragment blocking, *lds_out: LDS bytes
static int prepare(const i2r_conv_desc* d, int force_mt, int force_cap, int force_pf, ConvK& k, int* nt_out, int* mt_out, int* cap_out,
                   int* pf_out, size_t* lds_out, long long* nblk_out) {
    I2R_CHECK_ARG(d && d->in && d->w && d->bias && d->out, "i2r_conv: null pointer");
    I2R_CHECK_ARG(d->cin > 0 && d->cin % 16 == 0 && d->cin <= d->in_cs && d->in_cs % 4 == 0,
                  "i2r_conv: cin=%d must be a multiple of 16 and <= in_cs=%d (in_cs %% 4 == 0)", d->cin, d->in_cs);
    I2R_CHECK_ARG(d->cout > 0 && d->cout_pad % 16 == 0 && d->cout <= d->cout_pad && d->cout <= d->out_cs && d->out_cs % 4 == 0,
                  "i2r_conv: cout=%d cout_pad=%d out_cs=%d", d->cout, d->cout_pad, d->out_cs);
    I2R_CHECK_ARG(d->stride == 1 || d->stride == 2, "i2r_conv: stride %d", d->stride);
    I2R_CHECK_ARG(d->dtype >= 0 && d->dtype <= 2 && (d->dtype == 0 || d->in2 == nullptr), "i2r_conv: dtype %d", d->dtype);
    I2R_CHECK_ARG((d->in_f16 == 0 || d->in_f16 == 1) && (d->out_f16 == 0 || d->out_f16 == 1) && (d->dtype != 0 || (!d->in_f16 && !d->out_f16)),
                  "i2r_conv: 16-bit activation storage (in_f16=%d out_f16=%d) needs a 16-bit operand type (dtype=%d)", d->in_f16, d->out_f16, d->dtype);
    I2R_CHECK_ARG(!d->in_f16 || d->in_cs % 8 == 0, "i2r_conv: 16-bit input needs in_cs %% 8 == 0 (in_cs=%d)", d->in_cs);
    I2R_CHECK_ARG(d->ntaps >= 1 && d->ntaps <= I2R_MAX_TAPS, "i2r_conv: ntaps %d", d->ntaps);
    I2R_CHECK_ARG(d->rep >= 1 && d->out_step >= 1, "i2r_conv: rep/out_step");
    I2R_CHECK_ARG((d->conv_h - 1) * d->out_step + d->out_off_y + d->rep <= d->out_h &&
                      (d->conv_w - 1) * d->out_step + d->out_off_x + d->rep <= d->out_w,
                  "i2r_conv: destination grid exceeds out tensor");
    I2R_CHECK_ARG(d->in2 != (const float*)d->out && d->in != (const float*)d->out, "i2r_conv: out aliases in");
    I2R_CHECK_ARG(d->algo == 0 || d->algo == 1, "i2r_conv: algo %d", d->algo);
    if (d->algo == 1) return prepare_wino(d, force_mt, k, nt_out, mt_out, cap_out, pf_out, lds_out, nblk_out);

    int max_dy = 0, max_dx = 0;
    for (int t = 0; t < d->ntaps; ++t) {
        I2R_CHECK_ARG(d->dy[t] >= 0 && d->dx[t] >= 0, "i2r_conv: negative tap offset");
        if (d->dy[t] > max_dy) max_dy = d->dy[t];
        if (d->dx[t] > max_dx) max_dx = d->dx[t];
    }

    // ---- fragment decomposition of the output channels ----
    const int nfrag = d->cout_pad / 16;
    int nt = 0;
    for (int cand : {3, 4, 5})
        if (nfrag % cand == 0) { nt = cand; break; }
    I2R_CHECK_ARG(nt != 0, "i2r_conv: cout_pad=%d is not a multiple of 48, 64 or 80", d->cout_pad);
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
    int th = d->tile_h, tw = d->tile_w, mt = force_mt ? force_mt : d->mt;
    if (th == 0 || tw == 0) {
        tw = d->conv_w <= 16 ? d->conv_w : (d->conv_w % 16 == 0 ? 16 : (d->conv_w % 12 == 0 ? 12 : 8));
        if (mt == 0) mt = 2;
        th = (wm * mt * 16) / tw;
        if (th < 1) th = 1;
        if (th > d->conv_h) th = d->conv_h;
    }
    if (mt == 0) mt = cdiv(th * tw, wm * 16);
    I2R_CHECK_ARG(mt >= 1 && mt <= 4 && th * tw <= wm * mt * 16, "i2r_conv: tile %dx%d does not fit wm=%d mt=%d", th, tw, wm, mt);

    copy_desc(d, k);
    k.tile_h = th; k.tile_w = tw;
    k.tiles_y = cdiv(d->conv_h, th); k.tiles_x = cdiv(d->conv_w, tw);
    k.n_cblk = nfrag / (nt * wn);
    k.ph = (th - 1) * d->stride + max_dy + 1;
    k.pw = (tw - 1) * d->stride + max_dx + 1;
    I2R_CHECK_ARG(k.ph * k.pw <= kMaxPP * 256, "i2r_conv: patch %dx%d too large", k.ph, k.pw);
#ifdef I2R_TUNING
    static const int plane_pad = getenv("I2R_CONV_PLANE_PAD") ? atoi(getenv("I2R_CONV_PLANE_PAD")) : 0;  // tuning switch (LDS banks)
#else
    constexpr int plane_pad = 0;
#endif
    k.plane = cdiv(k.ph * k.pw, 16) * 16 + plane_pad;
    k.tap_kw = max_dx + 1;
    k.tap_kh = max_dy + 1;
    for (int t = 0; t < d->ntaps; ++t)
        I2R_CHECK_ARG(d->dy[t] == t / k.tap_kw && d->dx[t] == t % k.tap_kw && d->ntaps == (max_dy + 1) * (max_dx + 1),
                      "i2r_conv: taps must form a dense row-major kh x kw grid");
    // staging variant: double-buffered chunks when the patch fits 1-2 pixels per thread (the common case); a grouped
    // launch must agree on one variant (force_pf: -1 = free choice, else force_cap/force_pf are imposed)
    const int npp = cdiv(k.ph * k.pw, 256);
#ifdef I2R_TUNING
    static const int pf_env = getenv("I2R_CONV_PF") ? atoi(getenv("I2R_CONV_PF")) : 1;  // tuning switch: 0 disables prefetch
#else
    constexpr int pf_env = 1;
#endif
    // channel groups = 16-byte LDS slots per patch pixel: 4 fp32 channels, or 8 16-bit channels padded to whole 32-channel MFMA steps
    const int cin_g = d->dtype == 0 ? d->cin / 4 : (d->cin / 8 + 3) / 4 * 4;
    const int g_ch = d->dtype == 0 ? 4 : 8;
    int pf = (pf_env && d->ck == 0 && npp <= 2 && cin_g % 4 == 0) ? npp : 0;
    int cap = 4;
    if (force_pf >= 0) {
        I2R_CHECK_ARG(force_pf == 0 || (npp <= force_pf && d->ck == 0 && cin_g % 4 == 0), "i2r_conv: grouped members disagree on the staging mode");
        pf = force_pf;
    }
    int ck = d->ck;
    size_t lds_bytes;
    if (pf > 0) {
        int ckg = 4;
#ifdef I2R_TUNING
        static const int cap_env = getenv("I2R_CONV_CAP") ? atoi(getenv("I2R_CONV_CAP")) : 12;  // tuning switch: prefetch capacity limit
#else
        constexpr int cap_env = 12;
#endif
        const int cap_max = std::min(cap_env, d->dtype == 0 ? 12 : 8);  // prefetch registers of the kernel variants: fp32 up to 12 groups, 16-bit up to 8
        const int cap_lim = force_pf >= 0 ? force_cap : (pf == 1 ? cap_max : 4);
        for (int cand : {12, 8, 4})
            if (cand <= cap_lim && cin_g % cand == 0 && (size_t)2 * cand * k.plane * 16 <= 40 * 1024) { ckg = cand; break; }
        cap = force_pf >= 0 ? force_cap : (ckg > 4 ? cap_max : 4);
        ck = ckg * g_ch;
        lds_bytes = (size_t)2 * ckg * k.plane * 16;
    } else if (d->dtype != 0) {
        // bf16/f16 body: LDS slots hold 8 channels; stage whole 32-channel steps, <= ~24 KB per pass
        const int g8_pad = (d->cin / 8 + 3) / 4 * 4;
        int fit = (24 * 1024 / (k.plane * 16)) / 4 * 4;
        if (fit < 4) fit = 4;
        const int nchunk = cdiv(g8_pad, fit);
        const int ckg = cdiv(cdiv(g8_pad, nchunk), 4) * 4;
        ck = ckg * 8;
        lds_bytes = (size_t)ckg * k.plane * 16;
    } else {
        if (ck == 0) {
            // small LDS footprint (<= ~20 KB) keeps >= 4 workgroups per CU resident; equal-sized chunks avoid a short tail pass
            const int budget = 20 * 1024;
            int fit = (budget / (k.plane * 16)) * 4;
            fit = fit < 16 ? 16 : fit / 16 * 16;
            const int nchunk = cdiv(d->cin, fit);
            ck = cdiv(cdiv(d->cin, nchunk), 16) * 16;
        }
        I2R_CHECK_ARG(ck % 16 == 0 && ck >= 16, "i2r_conv: ck=%d", ck);
        lds_bytes = (size_t)(ck / 4) * k.plane * 16;
    }
    k.ck = ck;
    I2R_CHECK_ARG(lds_bytes <= 160 * 1024, "i2r_conv: LDS %zu B", lds_bytes);
    k.wn = wn;
    {
        // buffer descriptors of the kernels (in / weights / out): every tensor stays below 2 GiB so that kOOB lies beyond all of them
        const long long esz_in = d->in_f16 ? 2 : 4, esz_out = d->out_f16 ? 2 : 4;
        const long long inb = (long long)d->n_img * d->in_h * d->in_w * d->in_cs * esz_in, outb = (long long)d->n_img * d->out_h * d->out_w * d->out_cs * esz_out;
        const long long wb = d->dtype == 0 ? (long long)d->ntaps * d->cin * d->cout_pad * 4 : (long long)d->ntaps * ((d->cin / 8 + 3) / 4 * 4) * 8 * d->cout_pad * 2;
        I2R_CHECK_ARG(inb < (1ll << 31) && outb < (1ll << 31) && wb < (1ll << 31), "i2r_conv: tensors of 2 GiB and more are not supported (in %lld, out %lld, weights %lld bytes)", inb, outb, wb);
        k.in_bytes = (unsigned)inb; k.out_bytes = (unsigned)outb; k.w_bytes = (unsigned)wb;
        auto magic = [](int d) { return d == 1 ? 0u : (unsigned)(((1ull << 32) + d - 1) / d); };
        k.m_cblk = magic(k.n_cblk); k.m_tx = magic(k.tiles_x); k.m_ty = magic(k.tiles_y); k.m_pw = magic(k.pw); k.m_tw = magic(k.tile_w);
        k.wn_log = wn == 4 ? 2 : (wn == 2 ? 1 : 0);
        // chunks per workgroup: fp32 cin / ck; 16-bit: padded 8-channel groups / groups per chunk (the double-buffered variants use it)
        k.npass = d->dtype == 0 ? d->cin / ck : ((d->cin / 8 + 3) / 4 * 4) / (ck / 8);
    }
#ifdef I2R_TUNING
    {
        static const int dbg = getenv("I2R_CONV_DBG") ? atoi(getenv("I2R_CONV_DBG")) : 0;
        k.dbg = dbg;
    }
#endif
    const long long nblk = (long long)d->n_img * k.tiles_y * k.tiles_x * k.n_cblk;
    I2R_CHECK_ARG(nblk > 0 && nblk < (1ll << 30), "i2r_conv: grid");
    // ranges of the reciprocal divisions in the kernels (div_m): n < 2^32 / d
    I2R_CHECK_ARG(nblk * std::max(k.n_cblk, std::max(k.tiles_x, k.tiles_y)) < (1ll << 32) && (long long)kMaxPP * 256 * k.pw < (1ll << 32),
                  "i2r_conv: grid of %lld workgroups exceeds the index-decode range", nblk);
    *nt_out = nt; *mt_out = mt; *cap_out = cap; *pf_out = pf; *lds_out = lds_bytes; *nblk_out = nblk;
    return I2R_OK;
}

static int resolve(const i2r_conv_desc* const* descs, int32_t n, ConvGroupK& grp, int* nt0_, int* mt0_, int* cap0_, int* pf0_,
                   size_t* lds_, long long* total_, bool table = true) {
    I2R_CHECK_ARG(descs && n >= 1 && n <= kMaxGroups, "i2r_conv_grouped: 1..%d descriptors", kMaxGroups);
    int nt0 = 0, mt0 = 0, pf0 = -1, cap0 = 4;
    size_t lds_max = 0;
    long long total = 0;
    for (int i = 0; i < n; ++i) I2R_CHECK_ARG(descs[i] && descs[i]->algo == descs[0]->algo, "i2r_conv_grouped: members mix algorithms");
    if (n > 1 && descs[0]->algo == 0) {  // members must share the staging variant: the most general one any member needs
        pf0 = 0;
        cap0 = 4;
        bool all_pf = true;
        for (int i = 0; i < n; ++i) {
            ConvK tmp;
            int nt, mt, cap, pf;
            size_t lds;
            long long nblk;
            int rc = prepare(descs[i], mt0, 4, -1, tmp, &nt, &mt, &cap, &pf, &lds, &nblk);
            if (rc) return rc;
            if (i == 0) mt0 = mt;
            if (pf == 0) all_pf = false;
            if (pf > pf0) pf0 = pf;
            if (cap > cap0) cap0 = cap;
        }
        if (!all_pf) pf0 = 0;
        if (pf0 != 1) cap0 = 4;
        mt0 = 0;
    }
    for (int i = 0; i < n; ++i) {
        int nt, mt, cap, pf;
        size_t lds;
        long long nblk;
        int rc = prepare(descs[i], mt0, cap0, pf0, grp.g[i], &nt, &mt, &cap, &pf, &lds, &nblk);
        if (rc) return rc;
        if (i == 0) { nt0 = nt; mt0 = mt; pf0 = pf; cap0 = cap; }
        I2R_CHECK_ARG(nt == nt0 && mt == mt0, "i2r_conv_grouped: descriptor %d has fragment blocking (%d,%d) != (%d,%d)", i, mt, nt, mt0, nt0);
        I2R_CHECK_ARG(descs[i]->dtype == descs[0]->dtype, "i2r_conv_grouped: members mix compute dtypes");
        if (lds > lds_max) lds_max = lds;
        if (descs[i]->algo == 1 && !table) {  // XCD-aware item numbering of the Winograd kernels (i2r_conv_wino.hip): whole rounds of 8 fragment groups
            const long long groups = nblk / grp.g[i].n_cblk;
            grp.g[i].w_band = (int)((groups + 7) / 8);
            nblk = (long long)grp.g[i].w_band * 8 * grp.g[i].n_cblk;
        }
        total += nblk;
        grp.blk_end[i] = (int)total;
    }
    for (int i = n; i < kMaxGroups; ++i) { grp.g[i] = grp.g[0]; grp.blk_end[i] = (int)total; }
    grp.n = n;
    I2R_CHECK_ARG(total < (1ll << 31), "i2r_conv_grouped: grid");
    *nt0_ = nt0; *mt0_ = mt0; *cap0_ = cap0; *pf0_ = pf0; *lds_ = lds_max; *total_ = total;
    return I2R_OK;
}

extern "C" int i2r_conv_grouped(const i2r_conv_desc* const* descs, int32_t n, const int32_t* block_map,
                                int32_t map_len, void* stream) {
    ConvGroupK grp;
    int nt0, mt0, cap0, pf0;
    size_t lds_max;
    long long total;
    int rc = resolve(descs, n, grp, &nt0, &mt0, &cap0, &pf0, &lds_max, &total, block_map != nullptr);
    if (rc) return rc;
    grp.blk_map = block_map;
    I2R_CHECK_ARG(block_map == nullptr || map_len == (int32_t)total, "i2r_conv_grouped: block_map has %d entries, grid has %lld", map_len, total);
    conv_fn fn = descs[0]->algo == 1 ? reinterpret_cast<conv_fn>(i2r_pick_conv_wino(nt0, mt0))
                 : descs[0]->dtype == 0 ? pick_kernel(nt0, mt0, cap0, pf0)
                                      : reinterpret_cast<conv_fn>(descs[0]->dtype == 1 ? i2r_pick_conv_bf16(nt0, mt0, cap0, pf0) : i2r_pick_conv_f16(nt0, mt0, cap0, pf0));
    I2R_CHECK_ARG(fn != nullptr, "i2r_conv: no kernel for nt=%d mt=%d cap=%d pf=%d dtype=%d", nt0, mt0, cap0, pf0, descs[0]->dtype);
    if (lds_max > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
    const unsigned grid = (unsigned)total;
    i2r_launch(fn, dim3(grid), dim3(256), lds_max, (hipStream_t)stream, grp);
    I2R_CHECK_LAUNCH("i2r_conv");
    return I2R_OK;
}

extern "C" int i2r_conv(const i2r_conv_desc* d, void* stream) { return i2r_conv_grouped(&d, 1, nullptr, 0, stream); }

// ---- chain launches (see conv_chain_f32) ----
namespace {
typedef void (*chain_fn)(const ChainK);
chain_fn pick_chain(int nt, int mt, int cap, int pf) {
    // instantiated for the blockings the HRNet branch chains resolve to; anything else keeps the per-layer grouped launches
    if (nt == 3 && cap == 12 && pf == 1) {
        if (mt == 1) return conv_chain_f32<1, 3, 12, 1>;
        if (mt == 2) return conv_chain_f32<2, 3, 12, 1>;
        if (mt == 3) return conv_chain_f32<3, 3, 12, 1>;
        if (mt == 4) return conv_chain_f32<4, 3, 12, 1>;
    }
    return nullptr;
}
}  // namespace

extern "C" int i2r_conv_chain_pack(i2r_conv_chain_args* a, void* host_buf, int64_t host_bytes) {
    I2R_CHECK_ARG(a && a->descs && a->n_layers >= 1 && a->n_layers <= 32 && a->n_members >= 1 && a->n_members <= kMaxGroups,
                  "i2r_conv_chain: 1..32 layers x 1..%d members", kMaxGroups);
    const int L = a->n_layers, G = a->n_members;
    int nt0 = 0, mt0 = 0, cap0 = 0, pf0 = 0;
    size_t lds0 = 0;
    const int64_t need = (int64_t)L * G * (sizeof(ConvK) + sizeof(int));
    a->kdesc_bytes = (int32_t)need;
    ConvK* kd = reinterpret_cast<ConvK*>(host_buf);
    int* fb = host_buf ? reinterpret_cast<int*>(reinterpret_cast<char*>(host_buf) + (size_t)L * G * sizeof(ConvK)) : nullptr;
    I2R_CHECK_ARG(host_buf == nullptr || host_bytes >= need, "i2r_conv_chain_pack: buffer of %lld bytes, need %lld", (long long)host_bytes, (long long)need);
    int nflags = 0;
    for (int l = 0; l < L; ++l) {
        ConvGroupK grp;
        int nt, mt, cap, pf;
        size_t lds;
        long long total;
        int rc = resolve(a->descs + (size_t)l * G, G, grp, &nt, &mt, &cap, &pf, &lds, &total);
        if (rc) return rc;
        if (l == 0) { nt0 = nt; mt0 = mt; cap0 = cap; pf0 = pf; lds0 = lds; }
        I2R_CHECK_ARG(nt == nt0 && mt == mt0 && cap == cap0 && pf == pf0 && lds == lds0, "i2r_conv_chain: layer %d resolves to a different kernel variant", l);
        for (int g = 0; g < G; ++g) {
            const ConvK& k = grp.g[g];
            const i2r_conv_desc* d = a->descs[(size_t)l * G + g];
            I2R_CHECK_ARG(d->stride == 1 && d->rep == 1 && d->out_step == 1 && d->dtype == 0, "i2r_conv_chain: stride-1 fp32 layers only");
            I2R_CHECK_ARG(k.tap_kh <= 3 && k.tap_kw <= 3, "i2r_conv_chain: the dependency window is the 3x3 tile neighbourhood");
            if (l == 0) {
                a->tiles[g][0] = k.tiles_y; a->tiles[g][1] = k.tiles_x; a->tiles[g][2] = k.n_cblk;
                a->tiles[g][3] = k.n_img * k.tiles_y * k.tiles_x * k.n_cblk;
                I2R_CHECK_ARG(k.tile_h >= 1 && k.tile_w >= 1, "i2r_conv_chain: tile");
            } else {
                I2R_CHECK_ARG(a->tiles[g][0] == k.tiles_y && a->tiles[g][1] == k.tiles_x && a->tiles[g][2] == k.n_cblk,
                              "i2r_conv_chain: member %d changes its tiling at layer %d", g, l);
                I2R_CHECK_ARG(d->in == a->descs[(size_t)(l - 1) * G + g]->out, "i2r_conv_chain: layer %d of member %d does not read layer %d's output", l, g, l - 1);
            }
            if (kd) { kd[l * G + g] = k; fb[l * G + g] = nflags; }
            nflags += k.n_img * k.tiles_y * k.tiles_x;
        }
    }
    chain_fn fn = pick_chain(nt0, mt0, cap0, pf0);
    I2R_CHECK_ARG(fn != nullptr, "i2r_conv_chain: no chain kernel for nt=%d mt=%d cap=%d pf=%d", nt0, mt0, cap0, pf0);
    a->n_flags = nflags;
    a->nt = nt0; a->mt = mt0; a->cap = cap0; a->pf = pf0; a->lds_bytes = (int32_t)lds0;
    if (lds0 > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds0);
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(fn), 256, lds0) != hipSuccess) {
        (void)hipGetLastError();
        a->capacity = 0;  // (no device: sizes only)
    } else {
        a->capacity = per_cu * prop.multiProcessorCount;
    }
    return I2R_OK;
}

extern "C" int i2r_conv_chain(const i2r_conv_chain_args* a, void* stream) {
    I2R_CHECK_ARG(a && a->kdesc && a->item_ofs && a->items && a->flags && a->n_blocks >= 1, "i2r_conv_chain: null workspace");
    I2R_CHECK_ARG(a->capacity > 0 && a->n_blocks <= a->capacity, "i2r_conv_chain: %d workgroups exceed the resident capacity %d", a->n_blocks, a->capacity);
    chain_fn fn = pick_chain(a->nt, a->mt, a->cap, a->pf);
    I2R_CHECK_ARG(fn != nullptr, "i2r_conv_chain: run i2r_conv_chain_pack first");
    ChainK c;
    const int LG = a->n_layers * a->n_members;
    c.k = reinterpret_cast<const ConvK*>(a->kdesc);
    c.fbase = reinterpret_cast<const int*>(reinterpret_cast<const char*>(a->kdesc) + (size_t)LG * sizeof(ConvK));
    c.item_ofs = a->item_ofs; c.items = a->items; c.flags = a->flags;
    c.n_layers = a->n_layers; c.n_members = a->n_members; c.n_flags = a->n_flags;
    if (hipMemsetAsync(a->flags, 0, (size_t)(a->n_flags + 17) * sizeof(int), (hipStream_t)stream) != hipSuccess) {
        i2r_set_error("i2r_conv_chain: hipMemsetAsync failed");
        return I2R_E_LAUNCH;
    }
    i2r_launch(fn, dim3((unsigned)a->n_blocks), dim3(256), (size_t)a->lds_bytes, (hipStream_t)stream, c);
    I2R_CHECK_LAUNCH("i2r_conv_chain");
    return I2R_OK;
}

extern "C" int i2r_conv_kernel_name(const i2r_conv_desc* const* descs, int32_t n, char* buf, int32_t buflen) {
    ConvGroupK grp;
    int nt0, mt0, cap0, pf0;
    size_t lds_max;
    long long total;
    int rc = resolve(descs, n, grp, &nt0, &mt0, &cap0, &pf0, &lds_max, &total);
    if (rc) return rc;
    I2R_CHECK_ARG(buf && buflen > 0, "i2r_conv_kernel_name: buffer");
    if (descs[0]->algo == 1)
        snprintf(buf, (size_t)buflen, "conv_wino_f32<%d, %d>", mt0, nt0);
    else if (descs[0]->dtype)
        snprintf(buf, (size_t)buflen, "conv_igemm_lp<%d, %d, %d, %d>/%s", mt0, nt0, cap0, pf0, descs[0]->dtype == 1 ? "bf16" : "f16");
    else
        snprintf(buf, (size_t)buflen, "conv_igemm_f32<%d, %d, %d, %d>", mt0, nt0, cap0, pf0);
    return I2R_OK;
}
