// Probe: fp32 MFMA 16x16x4 issue cost as a function of the dependent-chain structure (NCH independent accumulators cycled),
// with WPS waves per SIMD.  cycles per MFMA per wave from s_memtime, and aggregate TFLOP/s.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NCH>
__global__ __launch_bounds__(1024) void k(float* out, long long* cyc, int iters) {
    f32x4 acc[NCH];
    for (int i = 0; i < NCH; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 24 / NCH; ++s)
#pragma unroll
            for (int i = 0; i < NCH; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
    for (int i = 0; i < NCH; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}
template <int NCH>
void run(float* out, long long* cyc, int wps) {
    const int iters = 2000, blocks = 256, threads = 256 * wps;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NCH>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NCH>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[16]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double n = (double)iters * 24;
    printf("chains %2d  waves/SIMD %d: %.3f ms  %.1f TF  %.1f ticks/MFMA/wave\n", NCH, wps, ms, blocks * (threads / 64) * n * 2048.0 / ms / 1e9, h[0] / n);
}
int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 16 * 8);
    for (int wps = 1; wps <= 2; ++wps) {
        run<1>(out, cyc, wps); run<2>(out, cyc, wps); run<3>(out, cyc, wps); run<4>(out, cyc, wps); run<6>(out, cyc, wps); run<12>(out, cyc, wps);
    }
    return 0;
}
