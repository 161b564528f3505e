// Probe: latency of a cross-stream dependency (hipEventRecord on stream A -> hipStreamWaitEvent on stream B) against an in-stream one,
// per event flag.  Kernels stamp wall_clock64 (100 MHz) at start and end; latency = start(consumer) - end(producer).
// build: hipcc --offload-arch=gfx950 -O2 tools/probe/xstream_latency.hip -o tools/probe/xstream_latency
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
__global__ void work(long long ticks, long long* stamp) {
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) stamp[0] = t0;
    while (wall_clock64() - t0 < ticks) {}
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x == 0) stamp[1] = wall_clock64();
}
int main() {
    hipStream_t a, b;
    hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    long long* st;
    hipHostMalloc(&st, 4 * sizeof(long long) * 64);
    const int N = 30;
    struct { const char* name; unsigned flags; } kinds[] = {
        {"default", hipEventDefault}, {"disableTiming", hipEventDisableTiming},
        {"disableTiming|releaseToDevice", hipEventDisableTiming | hipEventReleaseToDevice},
        {"disableTiming|releaseToSystem", hipEventDisableTiming | hipEventReleaseToSystem},
        {"disableTiming|disableSystemFence", hipEventDisableTiming | hipEventDisableSystemFence}};
    for (int grid : {1, 256}) {
        // in-stream reference: two kernels back to back on stream a
        std::vector<double> v;
        for (int i = 0; i < N; ++i) {
            hipLaunchKernelGGL(work, dim3(grid), dim3(64), 0, a, 2000LL, st);
            hipLaunchKernelGGL(work, dim3(grid), dim3(64), 0, a, 500LL, st + 2);
            hipStreamSynchronize(a);
            v.push_back((st[2] - st[1]) / 100.0);
        }
        std::sort(v.begin(), v.end());
        printf("grid %3d | same stream: median %.1f us (min %.1f)\n", grid, v[N / 2], v[0]);
        for (auto& k : kinds) {
            hipEvent_t e;
            if (hipEventCreateWithFlags(&e, k.flags) != hipSuccess) { printf("  %s: create failed\n", k.name); continue; }
            v.clear();
            for (int i = 0; i < N; ++i) {
                hipLaunchKernelGGL(work, dim3(grid), dim3(64), 0, a, 2000LL, st);   // 20 us producer
                hipEventRecord(e, a);
                hipStreamWaitEvent(b, e, 0);
                hipLaunchKernelGGL(work, dim3(grid), dim3(64), 0, b, 500LL, st + 2);  // consumer
                hipStreamSynchronize(b);
                hipStreamSynchronize(a);
                v.push_back((st[2] - st[1]) / 100.0);
            }
            std::sort(v.begin(), v.end());
            printf("grid %3d | cross stream, event %-36s: median %.1f us (min %.1f, max %.1f)\n", grid, k.name, v[N / 2], v[0], v[N - 1]);
            // with the consumer stream kept busy until the dependency (a long kernel in front of the wait): is the wait cheaper when b is not idle?
            v.clear();
            for (int i = 0; i < N; ++i) {
                hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, b, 1000LL, st + 4);       // b busy for 10 us
                hipLaunchKernelGGL(work, dim3(grid), dim3(64), 0, a, 2000LL, st);
                hipEventRecord(e, a);
                hipStreamWaitEvent(b, e, 0);
                hipLaunchKernelGGL(work, dim3(grid), dim3(64), 0, b, 500LL, st + 2);
                hipStreamSynchronize(b);
                hipStreamSynchronize(a);
                v.push_back((st[2] - st[1]) / 100.0);
            }
            std::sort(v.begin(), v.end());
            printf("grid %3d |   ... consumer stream busy before the wait             : median %.1f us (min %.1f, max %.1f)\n", grid, v[N / 2], v[0], v[N - 1]);
            hipEventDestroy(e);
        }
    }
    return 0;
}
