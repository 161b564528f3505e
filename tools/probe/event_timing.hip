// Probe: what do HIP timing events measure around a kernel of KNOWN duration, and at what cost?  (backs bench.py's in-situ timing)
//   A  hipEventRecord pairs around each launch           B  hipExtLaunchKernelGGL(start_i, stop_i)
//   C  hipExtLaunchKernelGGL(nullptr, stop_i): stop_i - stop_{i-1}     D  no per-launch events (one pair around all launches)
// build: hipcc --offload-arch=gfx950 -O2 tools/probe/event_timing.hip -o tools/probe/event_timing
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <vector>
__global__ void spin(long long ticks, long long* out) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (out && threadIdx.x == 0 && blockIdx.x == 0) *out = wall_clock64() - t0;
}
int main() {
    const int N = 40;
    hipStream_t s;
    hipStreamCreate(&s);
    long long* d;
    hipMalloc(&d, 8);
    for (long long us : {20LL, 50LL}) {
        const long long ticks = us * 100;  // wall_clock64: 100 MHz
        std::vector<hipEvent_t> a(N), b(N);
        for (int i = 0; i < N; ++i) { hipEventCreate(&a[i]); hipEventCreate(&b[i]); }
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {  // D
            hipEventRecord(e0, s);
            for (int i = 0; i < N; ++i) hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s, ticks, d);
            hipEventRecord(e1, s);
            hipStreamSynchronize(s);
        }
        float msD; hipEventElapsedTime(&msD, e0, e1);
        float tot;
        for (int rep = 0; rep < 2; ++rep) {  // A
            hipEventRecord(e0, s);
            for (int i = 0; i < N; ++i) { hipEventRecord(a[i], s); hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s, ticks, d); hipEventRecord(b[i], s); }
            hipEventRecord(e1, s);
            hipStreamSynchronize(s);
        }
        double sA = 0; for (int i = 0; i < N; ++i) { float ms; hipEventElapsedTime(&ms, a[i], b[i]); sA += ms; }
        hipEventElapsedTime(&tot, e0, e1);
        printf("kernel %lld us | D no events: %.2f us per launch | A record pairs: %.2f us (wall per launch %.2f)", us, msD / N * 1e3, sA / N * 1e3, tot / N * 1e3);
        for (int rep = 0; rep < 2; ++rep) {  // B
            hipEventRecord(e0, s);
            for (int i = 0; i < N; ++i) hipExtLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s, a[i], b[i], 0, ticks, d);
            hipEventRecord(e1, s);
            hipStreamSynchronize(s);
        }
        double sB = 0; int bad = 0; for (int i = 0; i < N; ++i) { float ms = -1; if (hipEventElapsedTime(&ms, a[i], b[i]) != hipSuccess) ++bad; sB += ms; }
        hipEventElapsedTime(&tot, e0, e1);
        printf(" | B ext start/stop: %.2f us (wall per launch %.2f, errors %d)", sB / N * 1e3, tot / N * 1e3, bad);
        for (int rep = 0; rep < 2; ++rep) {  // C
            hipEventRecord(e0, s);
            for (int i = 0; i < N; ++i) hipExtLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s, nullptr, b[i], 0, ticks, d);
            hipEventRecord(e1, s);
            hipStreamSynchronize(s);
        }
        double sC = 0; bad = 0; for (int i = 1; i < N; ++i) { float ms = -1; if (hipEventElapsedTime(&ms, b[i - 1], b[i]) != hipSuccess) ++bad; sC += ms; }
        hipEventElapsedTime(&tot, e0, e1);
        float self = -1; hipError_t se = hipEventElapsedTime(&self, b[3], b[3]);
        printf(" | C ext stop only, stop_i - stop_{i-1}: %.2f us (wall per launch %.2f, errors %d; elapsed(stop,stop) = %.3f rc %d)\n", sC / (N - 1) * 1e3, tot / N * 1e3, bad, self * 1e3, (int)se);
    }
    return 0;
}
