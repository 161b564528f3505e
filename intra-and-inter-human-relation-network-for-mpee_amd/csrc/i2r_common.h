// Shared helpers for the gfx950 kernels of libi2r_hip.so (internal; the public boundary is include/i2r_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>

#include "i2r_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// D = A(16x4) * B(4x16) + C on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32, 32 cycles/SIMD, exact fp32).
// lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]; D: col j = l&15, rows 4*(l>>4)+r.
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// 8 fp32 -> 8 bf16/f16 packed in 16 bytes (v_cvt_pk_*): the low-precision MFMA operand image of one (pixel, 8-channel group)
template <int DT>
__device__ __forceinline__ f32x4 pack8(f32x4 a, f32x4 b) {
    if constexpr (DT == 1) {
        bf16x8 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = (__bf16)a[i]; v[i + 4] = (__bf16)b[i]; }
        return __builtin_bit_cast(f32x4, v);
    } else {
        f16x8 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = (_Float16)a[i]; v[i + 4] = (_Float16)b[i]; }
        return __builtin_bit_cast(f32x4, v);
    }
}
template <int DT>
__device__ __forceinline__ f32x4 mfma32_lp(f32x4 a, f32x4 b, f32x4 c) {  // D = A(16x32) B(32x16) + C, fp32 accumulate
    if constexpr (DT == 1)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// exchange a value with another lane of the same quad (DPP quad_perm; CTRL 0xB1 = [1,0,3,2], 0x4E = [2,3,0,1])
template <int CTRL>
__device__ __forceinline__ float quad_xchg(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), CTRL, 0xF, 0xF, false));
}
// 4x4 transpose inside each quad of consecutive lanes (two DPP butterfly stages, no LDS): with j = lane & 3,
// out(lane j)[r] = in(lane r of the quad)[j]
__device__ __forceinline__ f32x4 quad_transpose(f32x4 v, int j) {
    const bool odd1 = j & 1, odd2 = j & 2;
    const float s0 = quad_xchg<0xB1>(odd1 ? v[0] : v[1]);
    const float s1 = quad_xchg<0xB1>(odd1 ? v[2] : v[3]);
    if (odd1) { v[0] = s0; v[2] = s1; } else { v[1] = s0; v[3] = s1; }
    const float t0 = quad_xchg<0x4E>(odd2 ? v[0] : v[2]);
    const float t1 = quad_xchg<0x4E>(odd2 ? v[1] : v[3]);
    if (odd2) { v[0] = t0; v[1] = t1; } else { v[2] = t0; v[3] = t1; }
    return v;
}

// XCD-banded work-item order for kernels whose items are listed in (image, row, column) order: workgroup b runs on XCD b % 8 (observed
// dispatch rule, MI355X_MICROARCH.md -- used for speed only), so XCD x is handed the CONTIGUOUS eighth [x total / 8, (x + 1) total / 8) of
// the list.  Two consecutive kernels that walk the same map in this order (the attention and MLP halves of a transformer block, whose
// 7x7 windows and 8x6 tiles do not coincide) then touch the same band of pixel rows from the same XCD: what one wrote with plain
// stores is still in that XCD's 4 MB L2 when the next one reads it.  Launch 8 * ceil(total / 8) workgroups; -1 = nothing to do.
__device__ __forceinline__ int xcd_band_item(int b, int total) {
    const int x = b & 7, q = b >> 3;
    const int lo = (int)(((long long)x * total) >> 3), hi = (int)(((long long)(x + 1) * total) >> 3);
    return lo + q < hi ? lo + q : -1;
}

void i2r_set_error(const char* fmt, ...);

// Every kernel launch of the library goes through i2r_launch.  Normally it IS kernel<<<grid, block, shmem, stream>>>(args...).  While
// i2r_run_program_timed replays an op, the calling thread's timing pair is set: the launch then goes through hipExtLaunchKernelGGL with
// the STOP event bound to the dispatch itself (its completion signal carries the kernel's end time: no extra packet) and, if given, the
// START event as a marker in front of it (include/i2r_hip.h: i2r_run_program_timed).  An op that launches several kernels keeps the
// first start and the last stop.
extern thread_local hipEvent_t i2r_tls_t0, i2r_tls_t1;
template <typename... KArgs, typename... Args>
inline void i2r_launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t shmem, hipStream_t stream, Args&&... args) {
    if (i2r_tls_t1) {
        hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)shmem, stream, i2r_tls_t0, i2r_tls_t1, 0, static_cast<KArgs>(args)...);
        i2r_tls_t0 = nullptr;
    } else {
        kernel<<<grid, block, shmem, stream>>>(static_cast<KArgs>(args)...);
    }
}

#define I2R_CHECK_ARG(cond, ...)            \
    do {                                    \
        if (!(cond)) {                      \
            i2r_set_error(__VA_ARGS__);     \
            return I2R_E_ARG;               \
        }                                   \
    } while (0)

#define I2R_CHECK_LAUNCH(what)                                              \
    do {                                                                    \
        hipError_t e__ = hipGetLastError();                                 \
        if (e__ != hipSuccess) {                                            \
            i2r_set_error("%s: %s", what, hipGetErrorString(e__));          \
            return I2R_E_LAUNCH;                                            \
        }                                                                   \
    } while (0)
