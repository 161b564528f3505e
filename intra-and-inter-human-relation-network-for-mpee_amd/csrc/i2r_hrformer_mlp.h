// Shared between the fused MLP-half kernels of the HRFormer-B transformer block (i2r_hrformer_mlp.hip: every wave accumulates fc2 over
// its own hidden blocks; i2r_hrformer_mlp_wide.hip: fc2 by output-block ownership, for the wide branches): argument block, 16-bit
// packing helpers and the one-transcendental GELU (internal).
#pragma once
#include "i2r_common.h"

struct I2rMlpK {
    const float* x; float* out;
    const float* ln_w; const float* ln_b;
    const f32x4* w1; const float* b1;     // fc1 (+BN1): 32-deep fragments [hidden block][KS k-steps][64 lanes]; bias [hidden_pad]
    const float* wdw; const float* bdw;   // depth-wise 3x3 (+BN2): [9][hidden_pad] tap-major; bias [hidden_pad]
    const f32x4* w2; const float* b2;     // fc2 (+BN3): 32-deep fragments [CB out blocks][hidden block pairs][64 lanes] (slot order); bias [cs]
    int n_img, h, w, c, tiles_y, tiles_x, total;
    float eps;
};

namespace {

template <int DT>
__device__ __forceinline__ uint2 pack4(f32x4 v) {
    if constexpr (DT == 1) {
        typedef __bf16 b16x4 __attribute__((ext_vector_type(4)));
        const b16x4 b = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
        return __builtin_bit_cast(uint2, b);
    } else {
        typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
        const h16x4 h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
        return __builtin_bit_cast(uint2, h);
    }
}
__device__ __forceinline__ f32x4 join8(uint2 lo, uint2 hi) {  // two packed 4-element halves -> one 8-element MFMA operand
    const uint4 v = {lo.x, lo.y, hi.x, hi.y};
    return __builtin_bit_cast(f32x4, v);
}
__device__ __forceinline__ float xsum4(float v) {  // over the 4 lanes that share l & 15
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

using MlpK = I2rMlpK;

// exact-erf GELU (nn.GELU, hrformer.py:1197) with ONE transcendental:  GELU(x) = max(x, 0) - |x| Phi(-|x|), and the normal tail
// Phi(-a) = erfc(a / sqrt 2) / 2 = 2^-Q(a), Q(0) = 1, Q a polynomial in a = min(|x|, 8) evaluated by Horner (minimax fit of the error of
// the GELU VALUE, tools/fit_gelu.py; beyond 8 the tail term is below 1e-14 |x|, so the clamped a also serves as |x| in the product).
// DEG 4: |error| < 9e-6 (hidden activations: two orders below one bf16 / f16 rounding); DEG 5: |error| < 1e-6 (the block output, added to
// the fp32 residual stream).  The VALU retires one wave64 instruction per 4 cycles, packed fp32 ones (v_pk_fma_f32) included, so the
// Horner steps and the final product run on PAIRS: 5.5 (6) issue slots per element: min, exp2, max + 5 (6) packed FMAs per pair.
// The fp32 parity kernels keep libm's erff.
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct GeluC {  // the coefficients as opaque SGPR pairs: literals would make the compiler pick scalar fmaak / fmamk over the packed FMAs
    f32x2 c1, c2, c3, c4, d1, d2, d3, d4, d5, one;
    __device__ __forceinline__ GeluC() {
        auto splat = [](float v) { f32x2 r = {v, v}; asm("" : "+s"(r)); return r; };
        c1 = splat(1.14955728e+00f); c2 = splat(4.64952799e-01f); c3 = splat(4.57202931e-02f); c4 = splat(-4.15856780e-03f);
        d1 = splat(1.15100107e+00f); d2 = splat(4.59593681e-01f); d3 = splat(5.21493605e-02f); d4 = splat(-7.20005350e-03f);
        d5 = splat(4.88322604e-04f); one = splat(1.f);
    }
};
template <int DEG>
__device__ __forceinline__ f32x2 gelu2(f32x2 x, const GeluC& k) {
    // (v_med3_f32: clamp without the NaN-canonicalising v_max the IEEE fminf / fmaxf forms cost)
    const f32x2 a = {__builtin_amdgcn_fmed3f(__builtin_fabsf(x[0]), 0.f, 8.f), __builtin_amdgcn_fmed3f(__builtin_fabsf(x[1]), 0.f, 8.f)};
    f32x2 t;
    if constexpr (DEG == 4) {
        t = a * k.c4 + k.c3;
        t = t * a + k.c2;
        t = t * a + k.c1;
    } else {
        t = a * k.d5 + k.d4;
        t = t * a + k.d3;
        t = t * a + k.d2;
        t = t * a + k.d1;
    }
    const f32x2 q = t * a + k.one;
    const f32x2 e = {__builtin_amdgcn_exp2f(-q[0]), __builtin_amdgcn_exp2f(-q[1])};
    const f32x2 r = {__builtin_amdgcn_fmed3f(x[0], 0.f, 3.0e38f), __builtin_amdgcn_fmed3f(x[1], 0.f, 3.0e38f)};
    return r - a * e;
}
template <int DEG>
__device__ __forceinline__ f32x4 gelu4(f32x4 v, const GeluC& k) {
    const f32x2 lo = gelu2<DEG>(v.xy, k), hi = gelu2<DEG>(v.zw, k);
    return (f32x4){lo[0], lo[1], hi[0], hi[1]};
}

}  // namespace

// the output-block-ownership kernel (i2r_hrformer_mlp_wide.hip); cs in {160, 320}; -> false if not built
bool i2r_mlp_wide_launch(const I2rMlpK& k, int dtype, int cs, long long nblk, hipStream_t stream);
