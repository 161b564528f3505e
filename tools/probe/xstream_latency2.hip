// Probe 2: what a hipStreamWaitEvent costs the WAITING stream when the event completed long ago, and chains of several waits.
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
    hipStream_t s[4];
    for (auto& x : s) (void)hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
    long long* st;
    (void)hipHostMalloc(&st, 16 * sizeof(long long));
    const int N = 30;
    for (unsigned flags : {(unsigned)hipEventDisableTiming, (unsigned)(hipEventDisableTiming | hipEventDisableSystemFence)}) {
        hipEvent_t e[3];
        for (auto& x : e) (void)hipEventCreateWithFlags(&x, flags);
        for (int nwait = 0; nwait <= 3; ++nwait) {
            std::vector<double> v;
            for (int i = 0; i < N; ++i) {
                // producers: short kernels on streams 1..3, recorded; consumer stream 0: 40 us kernel, then nwait waits on LONG-complete events, then a kernel
                for (int k = 0; k < nwait; ++k) { hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, s[k + 1], 200LL, st + 8); (void)hipEventRecord(e[k], s[k + 1]); }
                hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, s[0], 4000LL, st);
                for (int k = 0; k < nwait; ++k) (void)hipStreamWaitEvent(s[0], e[k], 0);
                hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, s[0], 200LL, st + 2);
                (void)hipDeviceSynchronize();
                v.push_back((st[2] - st[1]) / 100.0);
            }
            std::sort(v.begin(), v.end());
            printf("flags %#x: %d satisfied wait(s) between two kernels of one stream: gap median %.1f us (min %.1f max %.1f)\n", flags, nwait, v[N / 2], v[0], v[N - 1]);
        }
        // all-to-all among 3 streams (the engine's XSYNC): each records, each waits for the two others; stream 2 finishes last
        std::vector<double> v0, v2;
        for (int i = 0; i < N; ++i) {
            hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, s[0], 1000LL, st + 8);
            hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, s[1], 1500LL, st + 10);
            hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, s[2], 4000LL, st);
            for (int k = 0; k < 3; ++k) (void)hipEventRecord(e[k], s[k]);
            for (int d = 0; d < 3; ++d) for (int k = 0; k < 3; ++k) if (k != d) (void)hipStreamWaitEvent(s[d], e[k], 0);
            hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, s[0], 200LL, st + 2);
            hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, s[1], 200LL, st + 12);
            hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, s[2], 200LL, st + 4);
            (void)hipDeviceSynchronize();
            v0.push_back((st[2] - st[1]) / 100.0);
            v2.push_back((st[4] - st[1]) / 100.0);
        }
        std::sort(v0.begin(), v0.end()); std::sort(v2.begin(), v2.end());
        printf("flags %#x: 3-lane all-to-all, last finisher = lane 2: lane 0 restarts %.1f us after it (min %.1f), lane 2 itself %.1f us (min %.1f)\n", flags, v0[N / 2], v0[0], v2[N / 2], v2[0]);
    }
    return 0;
}
