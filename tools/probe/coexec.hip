// Probe: can fp32 MFMA waves and fp32 VALU (v_fmac with SGPR operand) waves on the same SIMDs add up?  (MI355X_MICROARCH.md:
// "MFMA and VALU pipes are separate ... run concurrently").  mode 0: all 8 waves MFMA, 1: all VALU, 2: waves 0-3 MFMA + 4-7 VALU
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(float* out, const float* w, int iters, int mode) {
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = mode == 0 || (mode == 2 && wave < 4);
    float r = 0.f;
    if (do_mfma) {
        f32x4 acc[9];
        for (int i = 0; i < 9; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 9; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 9; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        float acc[48];
        for (int j = 0; j < 48; ++j) acc[j] = 0.f;
        float x0 = threadIdx.x * 0.001f, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
        for (int it = 0; it < iters; ++it) {
            const float* ww = w + (it & 7) * 192;   // wave-uniform -> scalar loads
#pragma unroll
            for (int j = 0; j < 48; ++j) {
                acc[j] = fmaf(x0, ww[j * 4 + 0], acc[j]);
                acc[j] = fmaf(x1, ww[j * 4 + 1], acc[j]);
                acc[j] = fmaf(x2, ww[j * 4 + 2], acc[j]);
                acc[j] = fmaf(x3, ww[j * 4 + 3], acc[j]);
            }
        }
        for (int j = 0; j < 48; ++j) r += acc[j];
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
}
int main() {
    float *out, *w;
    hipMalloc(&out, 1024 * 512 * 4);
    hipMalloc(&w, 8 * 192 * 4);
    hipMemset(w, 0, 8 * 192 * 4);
    const int iters = 4000, blocks = 512;
    for (int mode = 0; mode < 3; ++mode) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, out, w, iters, mode);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, out, w, iters, mode);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // flops: mfma wave: iters*36 MFMAs * 2048 flop; valu wave: iters*192 fma * 64 lanes * 2
        double mf = (mode == 0 ? 8.0 : mode == 2 ? 4.0 : 0.0) * blocks * iters * 36.0 * 2048.0;
        double vf = (mode == 1 ? 8.0 : mode == 2 ? 4.0 : 0.0) * blocks * iters * 192.0 * 128.0;
        printf("mode %d: %.3f ms  MFMA %.1f TF + VALU %.1f TF = %.1f TF\n", mode, ms, mf / ms / 1e9, vf / ms / 1e9, (mf + vf) / ms / 1e9);
    }
    return 0;
}
