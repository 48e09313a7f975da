// Does a resident "keep-alive" wave (sleeping on a second stream) shorten the host round trip
// launch -> kernel runs -> host sees its doorbell, when the main stream would otherwise go idle between launches?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__device__ long long g_ts[2];
__global__ void k_work(int iters, double* out, volatile int* flag, int seq) {
    if (threadIdx.x == 0 && blockIdx.x == 0) g_ts[0] = wall_clock64();
    double a = threadIdx.x;
    for (int i = 0; i < iters; ++i)
        a = fma(a, 1.0000001, 0.5);
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
    if (threadIdx.x == 0 && blockIdx.x == 0) g_ts[1] = wall_clock64();
    __threadfence_system();
    if (threadIdx.x == 0 && blockIdx.x == 0)
        *flag = seq;
}
__global__ void k_keepalive(volatile int* stop, long long max_ticks) {
    const long long t0 = wall_clock64();
    while (!*stop && wall_clock64() - t0 < max_ticks) // bounded: 100 MHz ticks
        __builtin_amdgcn_s_sleep(127);
}
int main() {
    double* out;
    int *flag, *stop;
    CK(hipMalloc(&out, 8 * 64 * 4));
    CK(hipHostMalloc(&flag, 64));
    CK(hipHostMalloc(&stop, 64));
    *flag = 0;
    hipStream_t st, bg;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&bg, hipStreamNonBlocking));
    using clk = std::chrono::steady_clock;
    int seq = 0;
    for (int mode = 0; mode < 2; ++mode) {
        *stop = 0;
        if (mode == 1)
            hipLaunchKernelGGL(k_keepalive, dim3(1), dim3(64), 0, bg, (volatile int*)stop, 300000000LL); // <= 3 s
        for (int rep = 0; rep < 3; ++rep) {
            double tot = 0;
            const int n = 2000;
            for (int k = 1; k <= n; ++k) {
                ++seq;
                const auto t0 = clk::now();
                hipLaunchKernelGGL(k_work, dim3(4), dim3(64), 0, st, 2000, out, (volatile int*)flag, seq);
                while (*(volatile int*)flag != seq) {
                }
                tot += std::chrono::duration<double, std::micro>(clk::now() - t0).count();
                while (hipStreamQuery(st) == hipErrorNotReady) {
                }
                // ~5 us of host "work" between launches, as in the filter
                const auto t1 = clk::now();
                while (std::chrono::duration<double, std::micro>(clk::now() - t1).count() < 5.0) {
                }
            }
            printf("%s: %.2f us launch -> doorbell visible\n", mode == 0 ? "idle between launches " : "keep-alive wave resident", tot / n);
        }
        *stop = 1;
        CK(hipStreamSynchronize(bg));
        long long ts[2];
        CK(hipMemcpyFromSymbol(ts, HIP_SYMBOL(g_ts), 16));
        printf("   kernel body (device clock): %.2f us\n", (ts[1] - ts[0]) / 100.0);
    }
    return 0;
}
