// Probe: does a second wave on the SIMD hide the softmax (VALU) gap between the two MFMA runs of one attention step?
// Per iteration: 24 dependent MFMAs (S^T), ~40 VALU incl. 4 v_exp (softmax), 24 MFMAs on 6 accumulators (O^T).  No memory.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(1024) void k(float* out, long long* cyc, int iters, int variant) {
    f32x4 o[6], ka[6], va[6], q[6];
    for (int i = 0; i < 6; ++i) {
        o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < 4; ++r) { ka[i][r] = threadIdx.x * 1e-4f + i; va[i][r] = 1.f + r; q[i][r] = 0.01f * r; }
    }
    float m_run = -1e30f, l_run = 0.f;
    long long t0 = __builtin_amdgcn_s_memtime();
    f32x4 st_next = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (variant == 1) {
        for (int c = 0; c < 6; ++c)
            for (int s = 0; s < 4; ++s) st_next = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[c][s], q[c][s], st_next, 0, 0, 0);
    }
    for (int it = 0; it < iters; ++it) {
        f32x4 st;
        if (variant == 0) {
            st = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 6; ++c)
#pragma unroll
                for (int s = 0; s < 4; ++s) st = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[c][s], q[c][s], st, 0, 0, 0);
        } else {
            st = st_next;  // scores of THIS step were produced during the previous step; issue the next step's now
            st_next = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 6; ++c)
#pragma unroll
                for (int s = 0; s < 4; ++s) st_next = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[c][s] + (float)it, q[c][s], st_next, 0, 0, 0);
        }
        const float mx = fmaxf(fmaxf(st[0], st[1]), fmaxf(st[2], st[3]));
        if (__any(mx > m_run + 10.f)) {
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
            for (int nt = 0; nt < 6; ++nt) o[nt] *= alpha;
            m_run = m_new;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { st[r] = __builtin_amdgcn_exp2f(st[r] - m_run); l_run += st[r]; }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int nt = 0; nt < 6; ++nt) o[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(va[nt][s], st[s], o[nt], 0, 0, 0);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float r = l_run + st_next[0];
    for (int i = 0; i < 6; ++i) r += o[i][0] + o[i][1] + o[i][2] + o[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}
int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 16 * 8);
    const int iters = 2000;
    for (int variant = 0; variant < 2; ++variant)
        for (int wps = 1; wps <= 2; ++wps) {
            hipLaunchKernelGGL(k, dim3(256), dim3(256 * wps), 0, 0, out, cyc, iters, variant);
            hipDeviceSynchronize();
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(256 * wps), 0, 0, out, cyc, iters, variant);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%.3f ms, %.1f TF MFMA | ", ms, 256.0 * 4 * wps * iters * 48 * 2048 / ms / 1e9);
            long long h[8]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
            printf("variant %d (%s) waves/SIMD %d: %.0f ticks per step per wave (MFMA alone = 1536), SIMD MFMA busy %.0f%%\n", variant,
                   variant ? "S of the next step issued before this step's softmax" : "plain", wps, (double)h[0] / iters,
                   100.0 * 1536 * wps / ((double)h[0] / iters));
        }
    return 0;
}
