// Cross-workgroup flag latency on one MI355X: two workgroups of one grid exchange a flag (agent-scope release/acquire)
// plus an 8 KB tile, N round trips; printed per hop. Every spin is bounded. Partner index selects same / other XCD
// (workgroups are distributed round-robin over the 8 XCDs by workgroup id).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ bool wait_ge(int* f, int target) {
    for (int it = 0; it < 2000000; ++it) {
        if (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= target)
            return true;
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}
__global__ void __launch_bounds__(256) k_pingpong(int partner, int n, int payload, int* flags, double* buf, long long* cyc, double* out) {
    const int me = blockIdx.x;
    if (me != 0 && me != partner)
        return;
    __shared__ int ok;
    const int tid = threadIdx.x;
    double acc = 0;
    long long t0 = 0;
    if (tid == 0) {
        ok = 1;
        t0 = wall_clock64();
    }
    for (int k = 1; k <= n; ++k) {
        if (me == 0) {
            if (payload)
                for (int e = tid; e < 1024; e += 256)
                    buf[e] = k + e;
            __threadfence();
            __syncthreads();
            if (tid == 0)
                __hip_atomic_store(&flags[0], k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            if (tid == 0 && !wait_ge(&flags[1], k))
                ok = 0;
            __syncthreads();
            if (payload)
                for (int e = tid; e < 1024; e += 256)
                    acc += buf[1024 + e];
        } else {
            if (tid == 0 && !wait_ge(&flags[0], k))
                ok = 0;
            __syncthreads();
            if (payload)
                for (int e = tid; e < 1024; e += 256) {
                    acc += buf[e];
                    buf[1024 + e] = 2 * k + e;
                }
            __threadfence();
            __syncthreads();
            if (tid == 0)
                __hip_atomic_store(&flags[1], k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!ok)
            break;
    }
    if (tid == 0 && me == 0) {
        cyc[0] = wall_clock64() - t0;
        cyc[1] = ok;
    }
    out[me * 256 + tid] = acc;
}

int main() {
    int* flags;
    double *buf, *out;
    long long* cyc;
    CK(hipMalloc(&flags, 64));
    CK(hipMalloc(&buf, 2048 * 8));
    CK(hipMalloc(&out, 256 * 256 * 8));
    CK(hipMalloc(&cyc, 16));
    const int n = 2000;
    for (int payload = 0; payload < 2; ++payload)
        for (int partner : {8, 1, 4, 128, 255}) {
            CK(hipMemset(flags, 0, 64));
            hipLaunchKernelGGL(k_pingpong, dim3(256), dim3(256), 0, 0, partner, n, payload, flags, buf, cyc, out);
            CK(hipDeviceSynchronize());
            long long hc[2];
            CK(hipMemcpy(hc, cyc, 16, hipMemcpyDeviceToHost));
            printf("payload %d partner wg %3d (XCD %d): ok=%lld  %.3f us per hop (one-way flag%s)\n", payload, partner, partner % 8, hc[1], hc[0] / 100.0 / n / 2,
                   payload ? " + 8 KB tile" : "");
        }
    return 0;
}
