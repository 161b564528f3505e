// Shared between the two fused attention-half kernels of the HRFormer-B transformer block (i2r_hrformer_lp.hip: one wave per 16-token
// tile; i2r_hrformer_attn_head.hip: one wave per head) -- the kernel argument block and the 16-bit packing helpers (internal).
#pragma once
#include "i2r_common.h"

struct I2rAttnK {
    const float* x; float* out;
    const float* ln_w; const float* ln_b;
    const f32x4* wqkv; const float* bqkv;   // [head][q,k,v][KS k-steps][3 dim blocks][64 lanes] 16-byte fragments; [head][3][48] biases
    const f32x4* wo; const float* bo;       // [CB out blocks][HEADS*3/2 k-steps][64 lanes] (columns in slot order); [cs]
    int n_img, h, w, c, nwy, nwx, pad_top, pad_left, total;
    float eps;
};

template <int DT>
__device__ __forceinline__ uint2 i2r_pack4(f32x4 v) {
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
__device__ __forceinline__ float i2r_xsum4(float v) {  // over the 4 lanes that share l & 15
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}
__device__ __forceinline__ float i2r_xmax4(float v) {
    v = fmaxf(v, __shfl_xor(v, 16));
    v = fmaxf(v, __shfl_xor(v, 32));
    return v;
}

// the head-per-wave kernel (i2r_hrformer_attn_head.hip); (cs, heads) in {(80, 2), (160, 4), (320, 8), (624, 16)}; -> false if not built
bool i2r_attn_head_launch(const I2rAttnK& k, int dtype, int cs, int heads, long long nblk, hipStream_t stream);
