// Probe 3: cross-stream dependency through stream memory operations (hipStreamWriteValue32 / hipStreamWaitValue32) against events.
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
    hipStream_t s[3];
    for (auto& x : s) (void)hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
    long long* st;
    (void)hipHostMalloc(&st, 16 * sizeof(long long));
    uint32_t* flag;
    hipError_t e = hipExtMallocWithFlags((void**)&flag, 64, hipMallocSignalMemory);
    printf("signal memory alloc rc %d\n", (int)e);
    if (e != hipSuccess) return 0;
    (void)hipMemset(flag, 0, 64);
    const int N = 30;
    std::vector<double> v;
    for (int i = 1; i <= N; ++i) {  // pair: producer on s[1], consumer on s[0]
        hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, s[1], 2000LL, st);
        hipError_t e1 = hipStreamWriteValue32(s[1], flag, (uint32_t)i, 0);
        hipError_t e2 = hipStreamWaitValue32(s[0], flag, (uint32_t)i, hipStreamWaitValueGte, 0xffffffffu);
        hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, s[0], 500LL, st + 2);
        (void)hipDeviceSynchronize();
        if (i == 1) printf("write rc %d wait rc %d\n", (int)e1, (int)e2);
        v.push_back((st[2] - st[1]) / 100.0);
    }
    std::sort(v.begin(), v.end());
    printf("write/wait value, one dependency: consumer starts %.1f us after the producer ends (min %.1f max %.1f)\n", v[N / 2], v[0], v[N - 1]);
    // 3-lane all-to-all
    std::vector<double> v0, v2;
    for (int i = 1; i <= N; ++i) {
        const uint32_t val = 1000 + i;
        hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, s[0], 1000LL, st + 8);
        hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, s[1], 1500LL, st + 10);
        hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, s[2], 4000LL, st);
        for (int k = 0; k < 3; ++k) (void)hipStreamWriteValue32(s[k], flag + 4 * k, val, 0);
        for (int d = 0; d < 3; ++d) for (int k = 0; k < 3; ++k) if (k != d) (void)hipStreamWaitValue32(s[d], flag + 4 * k, val, hipStreamWaitValueGte, 0xffffffffu);
        hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, s[0], 200LL, st + 2);
        hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, s[1], 200LL, st + 12);
        hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, s[2], 200LL, st + 4);
        (void)hipDeviceSynchronize();
        v0.push_back((st[2] - st[1]) / 100.0);
        v2.push_back((st[4] - st[1]) / 100.0);
    }
    std::sort(v0.begin(), v0.end()); std::sort(v2.begin(), v2.end());
    printf("write/wait value, 3-lane all-to-all: lane 0 restarts %.1f us after the last finisher (min %.1f), the last finisher itself %.1f us (min %.1f)\n", v0[N / 2], v0[0], v2[N / 2], v2[0]);
    return 0;
}
