// Round 5: hand-off latency between two workgroups, by XCD placement and by the cache-policy bits of every access involved.
// One hop = producer stores an 8 KB tile, waits for the acknowledgement (s_waitcnt vmcnt(0)), raises a flag; consumer polls the flag, loads the tile, checks it.
//   ST  : data stores   0 plain | 1 sc1 (write-through to the device coherence point: what eqf_lookahead.hpp's la_st does)
//   FL  : flag poll     0 sc0   | 1 sc1 | 2 sc0 sc1
//   LD  : data loads    0 plain | 1 sc0 | 2 sc1
// Ping-pong between block 0 and block b, every word verified, n round trips. Output: us per hop.
// hipcc --offload-arch=gfx950 -O2 xcd_hop2.hip -o xcd_hop2_bin && ./xcd_hop2_bin
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11)) & 0xf; }
template <int ST> __device__ __forceinline__ void st_d(double* p, double v) {
    if (ST == 0) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
template <int ST> __device__ __forceinline__ void st_i(int* p, int v) {
    if (ST == 0) asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
template <int FL> __device__ __forceinline__ int ld_flag(const int* p) {
    int v;
    if (FL == 0) asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (FL == 1) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int LD> __device__ __forceinline__ void ld4(const double* p, double (&v)[4]) { // 4 doubles, 256 apart (tile of 1024, 256 threads)
    if (LD == 0)
        asm volatile("global_load_dwordx2 %0, %4, off\n\tglobal_load_dwordx2 %1, %4, off offset:2048\n\tglobal_load_dwordx2 %2, %5, off\n\tglobal_load_dwordx2 %3, %5, off offset:2048\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(p), "v"(p + 512) : "memory");
    else if (LD == 1)
        asm volatile("global_load_dwordx2 %0, %4, off sc0\n\tglobal_load_dwordx2 %1, %4, off offset:2048 sc0\n\tglobal_load_dwordx2 %2, %5, off sc0\n\tglobal_load_dwordx2 %3, %5, off offset:2048 sc0\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(p), "v"(p + 512) : "memory");
    else
        asm volatile("global_load_dwordx2 %0, %4, off sc1\n\tglobal_load_dwordx2 %1, %4, off offset:2048 sc1\n\tglobal_load_dwordx2 %2, %5, off sc1\n\tglobal_load_dwordx2 %3, %5, off offset:2048 sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(p), "v"(p + 512) : "memory");
}
// The tile moves every round trip (slot k & 15 of a 16-slot ring) so that a consumer never re-reads a line it has cached from an earlier trip unless the
// ring wraps (then the cached copy would be 16 trips old: a stale hit shows up as ok = 0).
template <int ST, int FL, int LD>
__global__ void __launch_bounds__(256) k_hop(int a, int b, int n, int* flags, double* buf, long long* cyc, int* info, long long* split) {
    const int me = blockIdx.x, tid = threadIdx.x;
    if (me != a && me != b) return;
    __shared__ int ok;
    if (tid == 0) { ok = 1; info[me == a ? 0 : 1] = xcc_id(); }
    __syncthreads();
    long long t0 = wall_clock64();
    long long t_ack = 0, t_poll = 0, t_load = 0;
    int* fmine = flags + (me == a ? 0 : 64);
    const int* ftheirs = flags + (me == a ? 64 : 0);
    for (int k = 1; k <= n; ++k) {
        double* mine = buf + (me == a ? 0 : 16 * 1024) + 1024 * (k & 15);
        const double* theirs = buf + (me == a ? 16 * 1024 : 0) + 1024 * (k & 15);
        if (me == a) {
            long long s0 = wall_clock64();
            for (int e = tid; e < 1024; e += 256) st_d<ST>(mine + e, k + e);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) { st_i<ST>(fmine, k); t_ack += wall_clock64() - s0; }
        }
        long long s1 = wall_clock64();
        if (tid == 0) { int it = 0; while (ld_flag<FL>(ftheirs) < k && ++it < 20000) {} if (it >= 20000) ok = -1; } // ~20 ms: the flag never became visible
        __syncthreads();
        if (ok < 0) { // give up (and release the partner, whose polls would otherwise run out one by one)
            if (tid == 0) st_i<1>(fmine, 0x7fffffff);
            break;
        }
        long long s2 = wall_clock64();
        double v[4];
        ld4<LD>(theirs + tid, v);
        for (int q = 0; q < 4; ++q) if (v[q] != (me == a ? 2.0 * k : 1.0 * k) + (tid + 256 * q)) ok = 0;
        if (tid == 0) { t_poll += s2 - s1; t_load += wall_clock64() - s2; }
        if (me == b) {
            for (int e = tid; e < 1024; e += 256) st_d<ST>(mine + e, 2.0 * k + e);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) st_i<ST>(fmine, k);
        }
        __syncthreads();
    }
    if (tid == 0 && me == a) { cyc[0] = wall_clock64() - t0; info[2] = ok; split[0] = t_ack; split[1] = t_poll; split[2] = t_load; }
    if (tid == 0 && me == b) info[3] = ok;
}
template <int ST, int FL, int LD> int run(int b, int* flags, double* buf, long long* cyc, int* info, long long* split) {
    const int n = 2000;
    CK(hipMemset(flags, 0, 1024)); CK(hipMemset(info, 0, 16)); CK(hipMemset(buf, 0, 32 * 1024 * 8));
    hipLaunchKernelGGL((k_hop<ST, FL, LD>), dim3(64), dim3(256), 0, 0, 0, b, n, flags, buf, cyc, info, split);
    CK(hipDeviceSynchronize());
    long long c, sp[3]; int h[4];
    CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(h, info, 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(sp, split, 24, hipMemcpyDeviceToHost));
    static const char* stn[] = {"plain", "sc1"}; static const char* fln[] = {"sc0", "sc1", "sc0sc1"}; static const char* ldn[] = {"plain", "sc0", "sc1"};
    fflush(stdout);
    printf("stores %-5s flag-poll %-6s loads %-5s | blocks 0 <-> %2d XCDs %d/%d ok %d %d | %.2f us per hop  (store+ack+flag %.2f, poll wait %.2f incl. the other side's turn, tile load %.2f)\n", stn[ST], fln[FL], ldn[LD], b, h[0], h[1], h[2], h[3],
           c * 0.01 / (2.0 * n), sp[0] * 0.01 / n, sp[1] * 0.01 / n, sp[2] * 0.01 / n);
    return 0;
}
int main() {
    int* flags; double* buf; long long* cyc; int* info; long long* split;
    CK(hipMalloc(&flags, 1024)); CK(hipMalloc(&buf, 32 * 1024 * 8)); CK(hipMalloc(&cyc, 8)); CK(hipMalloc(&info, 16)); CK(hipMalloc(&split, 24));
    for (int b : {8, 3}) { // 8: same XCD as block 0; 3: another XCD
        printf("--- block 0 <-> block %d\n", b);
        run<1, 1, 0>(b, flags, buf, cyc, info, split); // the look-ahead kernel's protocol today
        run<1, 1, 2>(b, flags, buf, cyc, info, split);
        run<1, 0, 0>(b, flags, buf, cyc, info, split);
        run<1, 0, 1>(b, flags, buf, cyc, info, split);
        run<0, 1, 2>(b, flags, buf, cyc, info, split);
        run<0, 1, 0>(b, flags, buf, cyc, info, split);
        run<0, 0, 1>(b, flags, buf, cyc, info, split);
        run<0, 0, 0>(b, flags, buf, cyc, info, split);
        run<0, 2, 1>(b, flags, buf, cyc, info, split);
    }
    return 0;
}
