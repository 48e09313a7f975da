// How much earlier does the host see a kernel's own write to pinned memory than hipStreamQuery reports completion?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void k_work(int iters, double* out, volatile int* flag, int seq) {
    double a = threadIdx.x;
    for (int i = 0; i < iters; ++i)
        a = fma(a, 1.0000001, 0.5);
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
    __threadfence_system();
    if (threadIdx.x == 0 && blockIdx.x == 0)
        *flag = seq;
}
int main() {
    double* out;
    int* flag;
    CK(hipMalloc(&out, 8 * 64 * 4));
    CK(hipHostMalloc(&flag, 64));
    *flag = 0;
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    using clk = std::chrono::steady_clock;
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            double tot = 0;
            const int n = 2000;
            for (int k = 1; k <= n; ++k) {
                const int seq = mode * 100000 + rep * 10000 + k;
                const auto t0 = clk::now();
                hipLaunchKernelGGL(k_work, dim3(4), dim3(64), 0, st, 2000, out, (volatile int*)flag, seq);
                if (mode == 0) {
                    while (hipStreamQuery(st) == hipErrorNotReady) {
                    }
                } else {
                    while (*(volatile int*)flag != seq) {
                    }
                }
                tot += std::chrono::duration<double, std::micro>(clk::now() - t0).count();
                if (mode == 1)
                    while (hipStreamQuery(st) == hipErrorNotReady) {
                    } // drain so the next launch starts from the same state
            }
            printf("%s: %.2f us launch -> host sees completion\n", mode == 0 ? "hipStreamQuery spin " : "pinned doorbell spin", tot / n);
        }
    return 0;
}
