// 3x3 stride-1 convolution as Winograd F(2x2, 3x3) on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32), gfx950.
//
// Why: the fp32 MFMA runs at the fp32 VECTOR rate (157 TFLOP/s, 1/16 of the 16-bit pipe) and the direct implicit-GEMM kernel
// (i2r_conv.hip) is bound by exactly that pipe; its HBM side sits at 9 % of the roof.  F(2x2, 3x3) computes a 2x2 output tile from
// a 4x4 input tile with 16 multiplies per (cin, cout) pair instead of 36:  Y = A^T [ (G g G^T) . (B^T d B) ] A  -- 2.25x fewer
// MFMAs for the same result (exact in exact arithmetic; in fp32 the rounding differs from the direct sum by a few ulp of the
// partial sums, far inside the 1e-3 parity bar -- tests/test_kernels_gpu.py).  The weights U = G g G^T are transformed once at
// pack time in float64 (engine.Packer.conv) and stored "k4" with the 16 Winograd positions in place of the 9 taps.
//
// Decomposition (one 4-wave workgroup):
//  * M = Winograd tiles.  A FRAGMENT is 16 tiles = 64 output pixels laid out FW x (16/FW) tiles (FW = 8, 4 or 2, chosen per map so
//    that fragments cover the map without waste: 64x48 -> 16x4 px, 32x24 -> 8x8 px, 16x12 -> 4x16 px); a work ITEM is MT
//    consecutive fragments of the flattened (crop, fy, fx) list x NT x 16 output channels.  Every fragment stages its own
//    (2 FH + 2) x (2 FW + 2) input patch, so the fragments of an item may belong to different crops.
//  * wave i (0..3) owns ROW i of the 4x4 Winograd positions: positions (i, 0..3) x MT fragments x NT channel fragments =
//    4 MT NT accumulator tiles.  Row i of  B^T d B  needs only two rows of d: the wave reads 2 x 4 patch pixels per tile from the RAW
//    patch in LDS and does the input transform in registers (8 ds_read_b128 + 32 VALU per fragment and 16-channel step, next to
//    48 MT MFMAs) -- the transformed tensor (3.2x the patch) never exists in LDS.
//  * LDS holds the raw patch of a 16-channel chunk, double-buffered, the next chunk fetched into registers under the current
//    chunk's MFMAs (as in i2r_conv.hip).  Columns are de-interleaved by parity (slot = row * pitch + (x & 1) * half + (x >> 1)) and
//    the pitch is chosen per FW so that the stride-2 tile gathers of a 16-lane ds_read_b128 group hit 16 distinct 16-byte slots.
//  * B operands (U, k-permuted k4 packing) stream from L2 straight into registers one position ahead, as in the direct kernel.
//  * epilogue: the column half of the output transform (over j) is in-lane; the row half (over i) crosses the four waves: each
//    wave writes its two partial tiles T_i[b] to LDS as [i][b][tile][cout] and every thread then owns (pixel, 4 channels) pieces:
//    three 16-byte LDS reads, bias / residuals / ReLU, one 16-byte store (a quad of lanes = 64 contiguous bytes of one pixel).
//  * Dispatch: the launch walks a host-made item table [rounds][workgroups] (entry = member << 24 | item).  By default it has ONE
//    round -- one workgroup per item in longest-processing-time order, the hardware dispatcher hands out slots as they free up.
//    The kernel can also run PERSISTENT workgroups (as many as are resident, each walking its column of an LPT-packed table and
//    fetching the next item's first chunk and weight fragments while the current item's partial tiles cross the waves): measured
//    80.5 vs 80.1 us per grouped stage-3 launch at 32 crops and 166.7 vs 154.4 us at 64 -- no gain, the per-item launch stays the
//    default (engine._WINO_BINS).  What the kernel IS sensitive to is occupancy: one fragment per item at 124 registers (4 waves per
//    SIMD) runs 78 us where two fragments per item (204 registers, 2 waves per SIMD) run 89 us, and software-pipelining the input
//    transform one chunk ahead in registers changed nothing at either size.  Hence the opaque copies of the thread index below:
//    without them the compiler hoists the thread-only index math of the epilogue out of the item loop and keeps ~20 registers
//    alive across the K loops (142 registers = 3 waves per SIMD).
#include "i2r_conv.h"

namespace {

constexpr int kWinoPatchMax = 108;  // patch pixels of a fragment: 6 x 18 (FW = 8 or 2), 10 x 10 (FW = 4)

#ifndef I2R_WINO_WAVES  // waves per SIMD the <1, 3> instantiation is compiled for (tuning builds try 3)
#define I2R_WINO_WAVES 4
#endif
template <int MT, int NT>
__global__ __launch_bounds__(256, (MT == 1 ? (NT == 3 ? I2R_WINO_WAVES : 3) : 2)) void conv_wino_f32(const ConvGroupK grp) {
    extern __shared__ __attribute__((aligned(16))) f32x4 lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wi = __builtin_amdgcn_readfirstlane(tid >> 6);  // Winograd row of this wave
    const int li = lane & 15, g = lane >> 4;
    const int nbins = gridDim.x;

    // ---- the item list of this workgroup: column blockIdx.x of the table (entry = member << 24 | item index within the member,
    //      -1 = none); without a table every workgroup has exactly one item, numbered member by member ----
    auto item_at = [&](int r) -> int {
        // (readfirstlane: the table may alias the kernel's stores as far as the compiler knows, so the load is a vector load -- without
        //  it the item, the member index and everything read from grp.g[] would live in vector registers)
        if (grp.blk_map) return r < grp.n_rounds ? __builtin_amdgcn_readfirstlane(grp.blk_map[r * nbins + (int)blockIdx.x]) : -1;
        if (r > 0) return -1;
        int bid = blockIdx.x, gi = 0, start = 0;
#pragma unroll
        for (int i = 0; i < kMaxGroups - 1; ++i)
            if (i + 1 < grp.n && bid >= grp.blk_end[i]) { gi = i + 1; start = grp.blk_end[i]; }
        return (gi << 24) | (bid - start);
    };

    // ---- per-item state.  S: staging tables of the item whose chunks are being fetched;  F: its fragments (two copies: the item being
    //      computed keeps `fc` for its epilogue while the next one is staged from `fn`);  C: operand addressing of the item being computed ----
    struct Frags {
        int img[MT], oy[MT], ox[MT], n_base;
        bool ok[MT];
    };
    constexpr int NIT = (MT * kWinoPatchMax * 4 + 255) / 256;
    int gs = 0;  // member index of the item being staged (grp.g[gs] is read through the kernel-argument segment: scalar loads.  Taking
                 // the ADDRESS of a member would force a private copy of the whole argument struct into scratch)
    const float* in_s = nullptr;
    int cin4_s = 0, cop_s = 0;
    int goff[NIT], lslot[NIT];  // global element offset of (pixel, channel group) or -1 (outside the image: zero), LDS slot or -1 (no item)
    Frags fn;
    const f32x4* wq_s = nullptr;  // its weight pointer (lane part included)
    auto setup = [&](int item) {
        gs = item >> 24;
        const ConvK& p = grp.g[gs];
        in_s = p.in;
        cin4_s = p.cin >> 2;
        cop_s = p.cout_pad;
        const int bid = item & 0xFFFFFF;
        const int cb = bid % p.n_cblk, wg = bid / p.n_cblk;
        const int fwl = p.w_fwlog, FW = 1 << fwl;
        const int PC = p.pw, PP = p.ph * p.pw;
        const int per_img = p.tiles_y * p.tiles_x;
        fn.n_base = __builtin_amdgcn_readfirstlane(cb * NT * 16);  // (the divisions run on the vector ALU: bring the uniform results back to SGPRs)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            int fid = wg * MT + mt;
            fn.ok[mt] = fid < p.w_nfrag;
            if (!fn.ok[mt]) fid = 0;
            const int img = fid / per_img;
            const int rem = fid - img * per_img;
            const int fy = rem / p.tiles_x;
            fn.img[mt] = __builtin_amdgcn_readfirstlane(img);
            fn.oy[mt] = __builtin_amdgcn_readfirstlane(fy * (32 >> fwl));
            fn.ox[mt] = __builtin_amdgcn_readfirstlane((rem - fy * p.tiles_x) * (2 * FW));
        }
        // staging assignment: item = (fragment, patch pixel, channel group of the 16-channel chunk); the four lanes of a quad fetch the
        // 64 contiguous bytes of one pixel (quad rule of the texture addresser, see i2r_conv.hip)
        int tid_s = tid;
        asm volatile("" : "+v"(tid_s));  // (as in the epilogue: no hoisting of the thread-only index math out of the item loop)
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int it = tid_s + k * 256;
            const int cg = it & 3, pix = it >> 2;
            int f = 0, pp = pix;  // (MT <= 2 and PP <= 108: compare / subtract instead of a division; rows through a 16-bit reciprocal)
#pragma unroll
            for (int m = 1; m <= MT; ++m)
                if (pix >= m * PP) { f = m; pp = pix - m * PP; }
            const int row = (pp * p.w_rcp) >> 16, col = pp - row * PC;
            goff[k] = -1;
            lslot[k] = -1;
            if (f < MT) {
                int img = fn.img[0], oy = fn.oy[0], ox = fn.ox[0];
                bool ok = fn.ok[0];
#pragma unroll
                for (int m = 1; m < MT; ++m)
                    if (f == m) { img = fn.img[m]; oy = fn.oy[m]; ox = fn.ox[m]; ok = fn.ok[m]; }
                const int iy = oy - 1 + row, ix = ox - 1 + col;
                lslot[k] = (f * 4 + cg) * p.plane + row * p.w_pitch + (col & 1) * p.w_half + (col >> 1);
                if (ok && iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w) goff[k] = ((img * p.in_h + iy) * p.in_w + ix) * p.in_cs + cg * 4;
            }
        }
        wq_s = reinterpret_cast<const f32x4*>(p.w) + fn.n_base + li;
    };
    f32x4 v[NIT];
    auto stage_load = [&](int c0) {
#pragma unroll
        for (int k = 0; k < NIT; ++k)
            v[k] = goff[k] >= 0 ? *reinterpret_cast<const f32x4*>(in_s + goff[k] + c0) : (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    auto stage_store = [&](f32x4* buf) {
#pragma unroll
        for (int k = 0; k < NIT; ++k)
            if (lslot[k] >= 0) buf[lslot[k]] = v[k];
    };

    // ---- C: A gather / B stream addressing of the item being computed ----
    //      lane (tile li, channel group g) gathers rows ra, rb of its tile's 4x4 patch and forms row i of B^T d B:
    //      i = 0: d0 - d2,  1: d1 + d2,  2: d2 - d1,  3: d1 - d3
    const int ra = wi == 0 ? 0 : (wi == 2 ? 2 : 1);
    const int rb = wi == 0 ? 2 : (wi == 1 ? 2 : (wi == 2 ? 1 : 3));
    const float sg = wi == 1 ? 1.f : -1.f;
    int gc = 0;
    Frags fc;
    int abase = 0, oa = 0, ob = 0, half = 0, plane4 = 0, npass = 0, bufsz = 0, cin4 = 0, cout_pad = 0;
    const f32x4* wq = nullptr;
    auto promote = [&]() {  // the staged item becomes the computed one
        gc = gs;
        fc = fn;
        const ConvK& p = grp.g[gc];
        const int fwl = p.w_fwlog;
        const int ty = li >> fwl, tx = li & ((1 << fwl) - 1);
        abase = g * p.plane + 2 * ty * p.w_pitch + tx;
        oa = ra * p.w_pitch;
        ob = rb * p.w_pitch;
        half = p.w_half;
        plane4 = 4 * p.plane;
        bufsz = MT * plane4;
        npass = p.cin >> 4;
        cin4 = cin4_s;
        cout_pad = cop_s;
        wq = wq_s;
    };
    auto load_a = [&](const f32x4* buf, f32x4 (&a)[MT][4]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const f32x4* q = buf + mt * plane4 + abase;
            f32x4 R[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int oc = (c & 1) * half + (c >> 1);
                const f32x4 xa = q[oa + oc], xb = q[ob + oc];
                R[c] = xa + sg * xb;
            }
            a[mt][0] = R[0] - R[2];
            a[mt][1] = R[1] + R[2];
            a[mt][2] = R[2] - R[1];
            a[mt][3] = R[1] - R[3];
        }
    };
    // B operand: U[pos = 4 i + j][cin / 4][cout_pad][4], lane (cout li, g) takes channels 4g..4g+3 of the 16-channel step
    auto fetch_b = [&](f32x4 (&b)[NT], const f32x4* w, int c4n, int cop, int pass, int j) {
        const f32x4* wp = w + (size_t)((wi * 4 + j) * c4n + pass * 4 + g) * cop;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[nt] = wp[nt * 16];
    };

    int item = item_at(0);
    if (item < 0) return;
    setup(item);
    stage_load(0);
    f32x4 a[MT][4], b0[NT], b1[NT];
    fetch_b(b0, wq_s, cin4_s, cop_s, 0, 0);
    float* const Tl = reinterpret_cast<float*>(lds);
    constexpr int TW = NT * 16;                    // floats per tile row of the exchange buffer
    constexpr int TPL = MT * 16 * TW;              // floats per (i, b) plane
    constexpr int C4 = NT * 4;                     // 16-byte channel pieces per pixel
    constexpr int NOUT = MT * 64 * C4 / 256;       // pieces per thread

    for (int round = 0;; ++round) {
        promote();
        const ConvK& p = grp.g[gc];
        // tuning aid (I2R_CONV_DBG & 8, -DI2R_TUNING builds only): phase time stamps of the item into the buffer passed as res2
        const bool stamp = (I2R_DBG(p) & 8) != 0;
        unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;
        if (stamp) ts0 = __builtin_amdgcn_s_memtime();
        const float* const res2 = stamp ? nullptr : p.res2;
        const int item_c = item;
        stage_store(lds);
        const int nx = item_at(round + 1);
        __syncthreads();
        if (stamp) ts1 = __builtin_amdgcn_s_memtime();

        f32x4 acc[4][MT][NT];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[j][mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#define WINO_MMA(J, B)                                                                                          \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)             \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) acc[J][mt][nt] = mfma16(a[mt][J][s], B[nt][s], acc[J][mt][nt]);
        for (int pass = 0; pass < npass; ++pass) {
            f32x4* cur = lds + (pass & 1) * bufsz;
            f32x4* nxt = lds + ((pass + 1) & 1) * bufsz;
            const bool more = pass + 1 < npass;
            load_a(cur, a);
            if (more) stage_load((pass + 1) * 16);
            fetch_b(b1, wq, cin4, cout_pad, pass, 1);
            WINO_MMA(0, b0)
            fetch_b(b0, wq, cin4, cout_pad, pass, 2);
            WINO_MMA(1, b1)
            fetch_b(b1, wq, cin4, cout_pad, pass, 3);
            WINO_MMA(2, b0)
            if (more) fetch_b(b0, wq, cin4, cout_pad, pass + 1, 0);
            WINO_MMA(3, b1)
            if (more) stage_store(nxt);
            __syncthreads();
        }
#undef WINO_MMA
        if (stamp) ts2 = __builtin_amdgcn_s_memtime();

        // ---- output transform.  Over j in registers:  T[0] = m0 + m1 + m2,  T[1] = m1 - m2 - m3 ----
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const f32x4 t0 = acc[0][mt][nt] + acc[1][mt][nt] + acc[2][mt][nt];
                const f32x4 t1 = acc[1][mt][nt] - acc[2][mt][nt] - acc[3][mt][nt];
                float* q = Tl + (wi * 2) * TPL + (mt * 16 + 4 * g) * TW + nt * 16 + li;  // D layout: rows 4g + r, column li
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    q[r * TW] = t0[r];
                    q[TPL + r * TW] = t1[r];
                }
            }
        if (nx >= 0) {  // the accumulators are free: set the next item up, fetch its first chunk and weight fragments under the exchange
            setup(nx);
            stage_load(0);
            fetch_b(b0, wq_s, cin4_s, cop_s, 0, 0);
        }
        // ---- over i across the waves:  Y[0][b] = T0 + T1 + T2,  Y[1][b] = T1 - T2 - T3;  thread = (pixel, 4 channels) ----
        int ooff[NOUT];  // element offset of the piece in out / res*, -1 = nothing to write
        f32x4 r[NOUT];
        int lrd[NOUT];
        const int fwl = p.w_fwlog;
        int tid_e = tid;  // (opaque copy: keeps the thread-only part of the index math inside the item instead of hoisted out of the item
        asm volatile("" : "+v"(tid_e));  //  loop, where it would hold ~20 registers across the K loops and cost a wave of occupancy)
#pragma unroll
        for (int k = 0; k < NOUT; ++k) {
            const int idx = tid_e + k * 256;
            const int c4 = idx % C4, pix = idx / C4;
            const int tile = pix >> 2, ay = (pix >> 1) & 1, bx = pix & 1;
            const int f = tile >> 4, t = tile & 15;
            int img = fc.img[0], oy = fc.oy[0], ox = fc.ox[0];
            bool ok = fc.ok[0];
#pragma unroll
            for (int m = 1; m < MT; ++m)
                if (f == m) { img = fc.img[m]; oy = fc.oy[m]; ox = fc.ox[m]; ok = fc.ok[m]; }
            oy += 2 * (t >> fwl) + ay;
            ox += 2 * (t & ((1 << fwl) - 1)) + bx;
            const int n = fc.n_base + c4 * 4;
            ok = ok && oy < p.conv_h && ox < p.conv_w && n < p.cout_pad && (n + 4 <= p.cout || n + 4 <= p.out_cs);
            ooff[k] = ok ? ((img * p.out_h + oy) * p.out_w + ox) * p.out_cs + n : -1;
            lrd[k] = (bx + (ay ? 2 : 0)) * TPL + tile * TW + c4 * 4;  // T0[b] (even output rows) or T1[b] (odd rows): the first of its three terms
            r[k] = n < p.cout_pad ? *reinterpret_cast<const f32x4*>(p.bias + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
            if (ok) {
                if (p.res1) r[k] += *reinterpret_cast<const f32x4*>(p.res1 + ooff[k]);
                if (res2) r[k] += *reinterpret_cast<const f32x4*>(res2 + ooff[k]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NOUT; ++k) {
            const float* q = Tl + lrd[k];
            const bool hi = (((tid_e + k * 256) / C4) & 2) != 0;
            const f32x4 u0 = *reinterpret_cast<const f32x4*>(q);             // T0 | T1
            const f32x4 u1 = *reinterpret_cast<const f32x4*>(q + 2 * TPL);   // T1 | T2
            const f32x4 u2 = *reinterpret_cast<const f32x4*>(q + 4 * TPL);   // T2 | T3
            f32x4 y = hi ? (u0 - u1 - u2) : (u0 + u1 + u2);
            if (ooff[k] < 0) continue;
            y += r[k];
            if (p.relu) { y[0] = fmaxf(y[0], 0.f); y[1] = fmaxf(y[1], 0.f); y[2] = fmaxf(y[2], 0.f); y[3] = fmaxf(y[3], 0.f); }
            if (p.res_post) y += *reinterpret_cast<const f32x4*>(p.res_post + ooff[k]);
            const int n = fc.n_base + (int)((tid_e + k * 256) % C4) * 4;
            if (n + 4 > p.cout) {  // channels >= cout are padding: keep them exactly zero
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n + e >= p.cout) y[e] = 0.f;
            }
            *reinterpret_cast<f32x4*>(p.out + ooff[k]) = y;
        }
        if (stamp) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long ts3 = __builtin_amdgcn_s_memtime();
            if (tid == 0) {
                unsigned long long* o = reinterpret_cast<unsigned long long*>(const_cast<float*>(p.res2)) + (size_t)(item_c & 0xFFFFFF) * 4;
                o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = ts3;
            }
        }
        if (nx < 0) break;
        item = nx;
        __syncthreads();  // the exchange buffer has been read: the next item's first chunk may overwrite it
    }
}

}  // namespace

void* i2r_pick_conv_wino(int nt, int mt) {
    conv_fn fn = nullptr;
    if (mt == 2 && nt == 3) fn = conv_wino_f32<2, 3>;
    if (mt == 1 && nt == 3) fn = conv_wino_f32<1, 3>;
    if (mt == 1 && nt == 4) fn = conv_wino_f32<1, 4>;
    return reinterpret_cast<void*>(fn);
}

// LDS bytes of a Winograd workgroup: two raw-patch chunk buffers, re-used by the cross-wave exchange of the output transform
size_t i2r_conv_wino_lds(int nt, int mt, int plane) {
    const size_t stage = (size_t)2 * mt * 4 * plane * 16;
    const size_t xchg = (size_t)4 * 2 * mt * 16 * nt * 16 * 4;
    return stage > xchg ? stage : xchg;
}
