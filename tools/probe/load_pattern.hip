// Probe: cost of a 64-lane global_load_dwordx4 as a function of the lane -> address pattern (data L2-resident).
// mode 0: lane-contiguous (1 KB per instruction); mode 1: MFMA-operand gather from a row-major matrix: lane (li, g) reads
// 16 B at row li (stride ROWB bytes), column 16 B * g  -> 16 rows x 64 B;  mode 2: 16 rows x 64 B with 4 consecutive lanes
// on the same row (lane = 4*row + chunk).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(const float* __restrict__ src, float* out, long long* cyc, int iters, int mode, int rowf) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, g = lane >> 4;
    size_t off;
    if (mode == 0) off = (size_t)lane * 4;
    else if (mode == 1) off = (size_t)li * rowf + 4 * g;
    else if (mode == 3) off = (size_t)lane * rowf;
    else off = (size_t)(lane >> 2) * rowf + 4 * (lane & 3);
    const float* base = src + (size_t)((blockIdx.x * 4 + wave) & 15) * 64 * rowf + off;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const float* p = base + (size_t)(it & 7) * 16 * 64 * rowf;
#pragma unroll
        for (int c = 0; c < 6; ++c) acc += *reinterpret_cast<const f32x4*>(p + (mode == 0 ? 256 * c : mode == 3 ? 4 * c : 16 * c));
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}
int main() {
    const int rowf = 96;  // floats per row (the encoder's K rows)
    float *src, *out; long long* cyc;
    const size_t n = (size_t)9 * 16 * 64 * rowf + 4096;
    hipMalloc(&src, n * 4); hipMemset(src, 0, n * 4);
    hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 1024 * 4 * 8);
    for (int blocks = 256; blocks <= 512; blocks *= 2)
        for (int mode = 0; mode < 4; ++mode) {
            const int iters = 2000;
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, src, out, cyc, iters, mode, rowf);
            hipDeviceSynchronize();
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, src, out, cyc, iters, mode, rowf);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long h[8]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
            printf("blocks %d (x4 waves) mode %d: %.3f ms, %.1f ticks per load per wave, %.2f TB/s aggregate\n", blocks, mode, ms, h[0] / (iters * 6.0),
                   (double)blocks * 4 * iters * 6 * 1024 / ms / 1e9);
        }
    return 0;
}
