// One-way hand-off latency between two workgroups of one grid as a function of their XCD placement and of the store / load flavour:
//   protocol M ("memory"): payload + flag as agent-scope relaxed atomics (global_store / global_load ... sc1): the look-ahead kernel's hand-off
//   protocol L ("L2"):     payload + flag as PLAIN stores (line stays in the XCD's L2) ordered by s_waitcnt vmcnt(0), read with sc1 loads
//                          (bypass the CU's L1, served by the L2): only valid when producer and consumer share an XCD
// 8 KB tile + flag, ping-pong, verified word by word. Placement from HW_REG_XCC_ID (block b is observed on XCD b % 8).
// hipcc --offload-arch=gfx950 -O2 xcd_hop.hip -o xcd_hop && ./xcd_hop
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11)) & 0xf; } // HW_REG_XCC_ID, bits 3:0
template <int P> __device__ __forceinline__ void st(double* p, double v) {
    if (P == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
template <int P> __device__ __forceinline__ void sti(int* p, int v) {
    if (P == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
__device__ __forceinline__ double ldd(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } // sc1: L2-served
__device__ __forceinline__ int ldi(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <int P>
__global__ void __launch_bounds__(256) k_hop(int a, int b, int n, int* flags, double* buf, long long* cyc, int* info) {
    const int me = blockIdx.x, tid = threadIdx.x;
    if (me != a && me != b) return;
    __shared__ int ok;
    if (tid == 0) { ok = 1; info[me == a ? 0 : 1] = xcc_id(); }
    __syncthreads();
    long long t0 = wall_clock64();
    double* mine = buf + (me == a ? 0 : 1024);
    const double* theirs = buf + (me == a ? 1024 : 0);
    int* fmine = flags + (me == a ? 0 : 64);
    const int* ftheirs = flags + (me == a ? 64 : 0);
    for (int k = 1; k <= n; ++k) {
        if (me == a) {
            for (int e = tid; e < 1024; e += 256) st<P>(mine + e, k + e);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) sti<P>(fmine, k);
        }
        if (tid == 0) { int it = 0; while (ldi(ftheirs) < k && ++it < 40000000) {} if (it >= 40000000) ok = 0; }
        __syncthreads();
        for (int e = tid; e < 1024; e += 256) if (ldd(theirs + e) != (me == a ? 2.0 * k + e : k + e)) ok = 0;
        if (me == b) {
            for (int e = tid; e < 1024; e += 256) st<P>(mine + e, 2.0 * k + e);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) sti<P>(fmine, k);
        }
        __syncthreads();
    }
    if (tid == 0 && me == a) { cyc[0] = wall_clock64() - t0; info[2] = ok; }
    if (tid == 0 && me == b) info[3] = ok;
}
int main() {
    int* flags; double* buf; long long* cyc; int* info;
    CK(hipMalloc(&flags, 1024)); CK(hipMalloc(&buf, 2048 * 8)); CK(hipMalloc(&cyc, 8)); CK(hipMalloc(&info, 16));
    const int n = 2000;
    for (int proto = 0; proto < 2; ++proto)
        for (int b : {8, 16, 1, 3}) { // 8, 16: same XCD as block 0 (b % 8 == 0); 1, 3: other XCDs
            if (proto == 1 && (b % 8) != 0) continue; // the L2 protocol is only valid inside one XCD
            CK(hipMemset(flags, 0, 1024)); CK(hipMemset(info, 0, 16));
            if (proto == 0) hipLaunchKernelGGL(k_hop<0>, dim3(64), dim3(256), 0, 0, 0, b, n, flags, buf, cyc, info);
            else hipLaunchKernelGGL(k_hop<1>, dim3(64), dim3(256), 0, 0, 0, b, n, flags, buf, cyc, info);
            CK(hipDeviceSynchronize());
            long long c; int h[4]; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(h, info, 16, hipMemcpyDeviceToHost));
            printf("protocol %s  blocks 0 <-> %2d  XCDs %d / %d  ok %d %d  %.2f us per hop (8 KB tile + flag)\n", proto ? "L2 (plain stores, sc1 loads)" : "memory (sc1 stores, sc1 loads)", b, h[0], h[1], h[2],
                   h[3], c * 0.01 / (2.0 * n));
        }
    return 0;
}
