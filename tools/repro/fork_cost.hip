// What a fork (main queue -> side queue dependency) costs the MAIN queue on this box, per mechanism:
//   0  no fork at all (baseline cadence of the main chain)
//   1  hipEventRecord (default event: system-scope fence) + hipStreamWaitEvent
//   2  hipEventRecord (hipEventDisableSystemFence)        + hipStreamWaitEvent
//   3  a one-thread kernel on the main stream writes a sequence number, the side stream waits with hipStreamWaitValue32
//   4  hipStreamWriteValue32 on the main stream,                         the side stream waits with hipStreamWaitValue32
// Main chain: N kernels that stream `mb` MB each (dependent, one stream); every `every` kernels a fork releases ONE
// small kernel on the side stream.  Timed with events around the main chain while the host runs ahead (spin first).
// Build: hipcc --offload-arch=gfx950 -O2 -o /tmp/fork_cost tools/repro/fork_cost.hip ; run: /tmp/fork_cost [mb] [N] [every]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void stream_kernel(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = a[i];
        v.x += 1.f;
        b[i] = v;
    }
}
__global__ void side_kernel(float* p) { if (threadIdx.x == 0) p[blockIdx.x] += 1.f; }
__global__ void flag_kernel(volatile unsigned* f, unsigned v) { *f = v; }
__global__ void spin_kernel(long long cycles) {
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
}

int main(int argc, char** argv) {
    const double mb = argc > 1 ? atof(argv[1]) : 22.0;
    const int N = argc > 2 ? atoi(argv[2]) : 240;
    const int every = argc > 3 ? atoi(argv[3]) : 6;
    const size_t n4 = (size_t)(mb * 1e6 / 16);
    float4 *a, *b;
    float* sidebuf;
    CK(hipMalloc(&a, n4 * 16 + 16));
    CK(hipMalloc(&b, n4 * 16 + 16));
    CK(hipMalloc(&sidebuf, 4096));
    CK(hipMemset(a, 0, n4 * 16 + 16));
    CK(hipMemset(sidebuf, 0, 4096));
    int can_wait = 0;
    CK(hipDeviceGetAttribute(&can_wait, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can_wait);
    unsigned* flag = nullptr;
    if (can_wait) {
        hipError_t e = hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory);
        if (e != hipSuccess) { printf("signal memory: %s\n", hipGetErrorString(e)); can_wait = 0; (void)hipGetLastError(); }
        else CK(hipMemset(flag, 0, 8));
    }
    hipStream_t ms, ss;
    CK(hipStreamCreateWithFlags(&ms, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&ss, hipStreamNonBlocking));
    std::vector<hipEvent_t> ev_f(N), ev_nf(N);
    for (int i = 0; i < N; ++i) {
        CK(hipEventCreateWithFlags(&ev_f[i], hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&ev_nf[i], hipEventDisableTiming | hipEventDisableSystemFence));
    }
    hipEvent_t t0, t1, sj;
    CK(hipEventCreate(&t0));
    CK(hipEventCreate(&t1));
    CK(hipEventCreateWithFlags(&sj, hipEventDisableTiming));
    const int grid = 2048;
    unsigned seq = 0;
    if (can_wait) {
        // an already satisfied wait first: if this does not come back the mechanism is unusable here
        hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(1), 0, ms, flag, 1u);
        CK(hipStreamSynchronize(ms));
        hipError_t e = hipStreamWaitValue32(ss, flag, 1u, hipStreamWaitValueGte, 0xFFFFFFFFu);
        printf("hipStreamWaitValue32 (satisfied) -> %s\n", hipGetErrorString(e));
        if (e != hipSuccess) { can_wait = 0; (void)hipGetLastError(); }
        else {
            hipLaunchKernelGGL(side_kernel, dim3(1), dim3(64), 0, ss, sidebuf);
            CK(hipStreamSynchronize(ss));
            printf("satisfied wait + kernel completed\n");
        }
        seq = 1;
    }
    for (int mode = 0; mode <= 4; ++mode) {
        if (mode >= 3 && !can_wait) { printf("mode %d skipped (no stream wait-value)\n", mode); continue; }
        double best = 1e30, sum = 0;
        const int reps = 5;
        for (int r = 0; r < reps + 1; ++r) {
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(1), 0, ms, 300000LL);      // ~3 ms at 100 MHz: the host runs ahead
            CK(hipEventRecord(t0, ms));
            for (int i = 0; i < N; ++i) {
                hipLaunchKernelGGL(stream_kernel, dim3(grid), dim3(256), 0, ms, (i & 1) ? b : a, (i & 1) ? a : b, n4);
                if (mode && (i % every) == every - 1) {
                    if (mode == 1 || mode == 2) {
                        hipEvent_t e = mode == 1 ? ev_f[i] : ev_nf[i];
                        CK(hipEventRecord(e, ms));
                        CK(hipStreamWaitEvent(ss, e, 0));
                    } else if (mode == 3) {
                        ++seq;
                        hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(1), 0, ms, flag, seq);
                        CK(hipStreamWaitValue32(ss, flag, seq, hipStreamWaitValueGte, 0xFFFFFFFFu));
                    } else {
                        ++seq;
                        CK(hipStreamWriteValue32(ms, flag, seq, 0));
                        CK(hipStreamWaitValue32(ss, flag, seq, hipStreamWaitValueGte, 0xFFFFFFFFu));
                    }
                    hipLaunchKernelGGL(side_kernel, dim3(8), dim3(64), 0, ss, sidebuf);
                }
            }
            CK(hipEventRecord(t1, ms));
            CK(hipEventRecord(sj, ss));
            CK(hipStreamWaitEvent(ms, sj, 0));
            CK(hipStreamSynchronize(ms));
            CK(hipStreamSynchronize(ss));
            float ms_ = 0;
            CK(hipEventElapsedTime(&ms_, t0, t1));
            if (r) { sum += ms_; if (ms_ < best) best = ms_; }
        }
        printf("mode %d: main chain of %d x %.0f MB kernels, fork every %d: mean %.1f us, best %.1f us  (%.2f us per kernel)\n",
               mode, N, mb, every, sum / reps * 1e3, best * 1e3, best * 1e3 / N);
        fflush(stdout);
    }
    printf("done\n");
    return 0;
}
