// Stand-alone reproducer for the round-3 finding "destroying a captured HIP graph and capturing another one crashes" (DESIGN section 2):
// capture -> instantiate -> launch -> destroy -> capture ..., no msau code, no torch.  Two streams inside the capture (fork / join by
// events), as the training step's backward has.    hipcc --offload-arch=gfx950 -O2 graph_destroy.hip -o graph_destroy && ./graph_destroy 60
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAIL %s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 2; } } while (0)
__global__ void axpy(float* y, const float* x, float a, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) y[i] = a * x[i] + y[i]; }
int main(int argc, char** argv) {
    const int cycles = argc > 1 ? atoi(argv[1]) : 40, n = 1 << 20, nk = argc > 2 ? atoi(argv[2]) : 300;
    float *x, *y, *z;
    CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y, n * 4)); CK(hipMalloc(&z, n * 4));
    CK(hipMemset(x, 0, n * 4)); CK(hipMemset(y, 0, n * 4)); CK(hipMemset(z, 0, n * 4));
    hipStream_t s, side;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    // a POOL of events, one per fork, reused by every sweep -- captured or eager -- as msau_run_ops_overlap does (argv[3] = 1: one event)
    const int pool_n = (argc > 3 && atoi(argv[3]) == 1) ? 1 : 64;
    hipEvent_t pool[64], join;
    for (int i = 0; i < pool_n; ++i) CK(hipEventCreateWithFlags(&pool[i], hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    auto sweep = [&]() -> int {
        int used = 0;
        for (int k = 0; k < nk; ++k) {
            hipLaunchKernelGGL(axpy, dim3(n / 256), dim3(256), 0, s, y, x, 1.0f, n);
            if (k % 6 == 5) {                                   // side-stream launches behind a fork, as the weight gradients
                hipEvent_t fork = pool[used++ % pool_n];
                CK(hipEventRecord(fork, s)); CK(hipStreamWaitEvent(side, fork, 0));
                for (int j = 0; j < 4; ++j) hipLaunchKernelGGL(axpy, dim3(n / 256), dim3(256), 0, side, z, x, 2.0f, n);
            }
        }
        CK(hipEventRecord(join, side)); CK(hipStreamWaitEvent(s, join, 0));
        return 0;
    };
    for (int c = 0; c < cycles; ++c) {
        hipGraph_t g; hipGraphExec_t ge;
        if (sweep()) return 2;                                  // an eager sweep (the warm-up before a capture) with the same events
        CK(hipStreamSynchronize(s));
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        if (sweep()) return 2;
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
        if (c % 10 == 9) { printf("cycle %d ok\n", c + 1); fflush(stdout); }
    }
    printf("PASS %d capture / destroy cycles of %d-kernel graphs\n", cycles, nk);
    return 0;
}
