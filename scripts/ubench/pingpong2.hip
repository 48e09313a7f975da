// Cross-workgroup hand-off variants on one MI355X (gfx950), for the look-ahead factorisation kernel:
//   mode 0  plain stores + __threadfence() (agent release: L2 write-back) + acquire flag load   [= pingpong.hip]
//   mode 1  payload and flag as RELAXED agent-scope atomics (write-through stores / cache-bypassing loads), ordered by
//           s_waitcnt vmcnt(0) + workgroup barrier only: no L2 write-back, no cache invalidate
//   mode 2  mode 1 with a tight poll (no s_sleep)
// Two workgroups exchange a flag (+ an 8 KB tile) N times; the printed figure is the one-way latency per hop.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE> __device__ __forceinline__ bool wait_ge(int* f, int target) {
    for (int it = 0; it < 4000000; ++it) {
        const int v = MODE == 0 ? __hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) : __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v >= target)
            return true;
        if (MODE != 2)
            __builtin_amdgcn_s_sleep(1);
    }
    return false;
}
template <int MODE> __device__ __forceinline__ void put(double* p, double v) {
    if (MODE == 0)
        *p = v;
    else
        __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int MODE> __device__ __forceinline__ double get(const double* p) {
    return MODE == 0 ? *p : __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int MODE> __device__ __forceinline__ void publish(int* flag, int k, int tid) {
    if (MODE == 0)
        __threadfence();
    else
        __builtin_amdgcn_s_waitcnt(0); // vmcnt(0) expcnt(0) lgkmcnt(0): this wave's write-through stores are complete
    __syncthreads();
    if (tid == 0) {
        if (MODE == 0)
            __hip_atomic_store(flag, k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        else
            __hip_atomic_store(flag, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
template <int MODE>
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
    __syncthreads();
    for (int k = 1; k <= n; ++k) {
        if (me == 0) {
            if (payload)
                for (int e = tid; e < 1024; e += 256)
                    put<MODE>(&buf[e], k + e);
            publish<MODE>(&flags[0], k, tid);
            if (tid == 0 && !wait_ge<MODE>(&flags[32], k))
                ok = 0;
            __syncthreads();
            if (payload)
                for (int e = tid; e < 1024; e += 256) {
                    const double v = get<MODE>(&buf[1024 + e]);
                    if (v != 2 * k + e)
                        ok = 0; // stale payload
                    acc += v;
                }
        } else {
            if (tid == 0 && !wait_ge<MODE>(&flags[0], k))
                ok = 0;
            __syncthreads();
            if (payload)
                for (int e = tid; e < 1024; e += 256) {
                    const double v = get<MODE>(&buf[e]);
                    if (v != k + e)
                        ok = 0;
                    acc += v;
                    put<MODE>(&buf[1024 + e], 2 * k + e);
                }
            publish<MODE>(&flags[32], k, tid);
        }
        __syncthreads();
        if (!ok)
            break;
    }
    if (tid == 0 && me == 0) {
        cyc[0] = wall_clock64() - t0;
        cyc[1] = ok;
    }
    out[me * 256 + tid] = acc;
}
// broadcast: workgroup 0 publishes an 8 KB tile, R reader workgroups (ids 1..R) pick it up and acknowledge; per round = publish -> all acks seen
template <int MODE>
__global__ void __launch_bounds__(256) k_bcast(int R, int n, int* flags, double* buf, long long* cyc, double* out) {
    const int me = blockIdx.x;
    if (me > R)
        return;
    __shared__ int ok;
    const int tid = threadIdx.x;
    double acc = 0;
    long long t0 = 0;
    if (tid == 0) {
        ok = 1;
        t0 = wall_clock64();
    }
    __syncthreads();
    for (int k = 1; k <= n; ++k) {
        if (me == 0) {
            for (int e = tid; e < 1024; e += 256)
                put<MODE>(&buf[e], k + e);
            publish<MODE>(&flags[0], k, tid);
            if (tid < R && !wait_ge<MODE>(&flags[32 * (tid + 1)], k))
                ok = 0;
            __syncthreads();
        } else {
            if (tid == 0 && !wait_ge<MODE>(&flags[0], k))
                ok = 0;
            __syncthreads();
            for (int e = tid; e < 1024; e += 256) {
                const double v = get<MODE>(&buf[e]);
                if (v != k + e)
                    ok = 0;
                acc += v;
            }
            publish<MODE>(&flags[32 * me], k, tid);
        }
        __syncthreads();
        if (!ok)
            break;
    }
    if (tid == 0 && me == 0) {
        cyc[0] = wall_clock64() - t0;
        cyc[1] = ok;
    }
    out[me * 256 + tid] = acc;
}


// mode 3: no separate flag. Every payload element travels as one 16-byte (value, sequence) pair written by ONE global_store_dwordx4 sc1
// and read by ONE global_load_dwordx4 sc1 (a 16-byte aligned access never straddles a 32-byte sector); a lane polls lane 0's first
// pair, then loads its own pairs and re-polls the ones that still carry the old sequence number. The hop is one store -> load latency.
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st16(void* p, double v, int seq) {
    v4i x;
    x.x = __double2loint(v);
    x.y = __double2hiint(v);
    x.z = seq;
    x.w = ~seq;
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(x) : "memory");
}
__device__ __forceinline__ v4i ld16(const void* p) {
    v4i r;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"(p) : "memory");
    return r;
}
__device__ __forceinline__ void ld16x4(const char* base, int e0, v4i (&r)[4]) { // four loads in flight, one wait
    const char* p0 = base + 16 * (size_t)e0;
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\t"
                 "global_load_dwordx4 %1, %5, off sc1\n\t"
                 "global_load_dwordx4 %2, %6, off sc1\n\t"
                 "global_load_dwordx4 %3, %7, off sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3])
                 : "v"(p0), "v"(p0 + 4096), "v"(p0 + 8192), "v"(p0 + 12288)
                 : "memory");
}
__global__ void __launch_bounds__(256) k_pingpong16(int partner, int n, char* buf, long long* cyc, double* out) {
    const int me = blockIdx.x;
    if (me != 0 && me != partner)
        return;
    const int tid = threadIdx.x;
    double acc = 0;
    long long t0 = 0;
    int ok = 1;
    if (tid == 0)
        t0 = wall_clock64();
    char* mine = buf + (me == 0 ? 0 : 16 * 1024);
    const char* theirs = buf + (me == 0 ? 16 * 1024 : 0);
    for (int k = 1; k <= n && ok; ++k) {
        if (me == 0)
            for (int e = tid; e < 1024; e += 256)
                st16(mine + 16 * (size_t)e, k + e, k);
        {
            v4i r[4];
            int it = 0;
            for (; it < 4000000; ++it) {
                ld16x4(theirs, tid, r);
                bool all = true;
                for (int q = 0; q < 4; ++q)
                    all = all && r[q].z == k && r[q].w == ~k;
                if (all)
                    break;
            }
            if (it == 4000000)
                ok = 0;
            for (int q = 0; q < 4; ++q) {
                const double v = __hiloint2double(r[q].y, r[q].x);
                const int e = tid + 256 * q;
                if (v != (me == 0 ? 2 * k + e : k + e))
                    ok = 0;
                acc += v;
            }
        }
        if (me != 0)
            for (int e = tid; e < 1024; e += 256)
                st16(mine + 16 * (size_t)e, 2 * k + e, k);
    }
    ok = __syncthreads_and(ok);
    if (tid == 0 && me == 0) {
        cyc[0] = wall_clock64() - t0;
        cyc[1] = ok;
    }
    out[me * 256 + tid] = acc;
}

// tearing test for mode 3: one writer, R readers polling the SAME 8 KB tile (the factorisation kernel's pattern), values depend on the round;
// a word with the right sequence number and a wrong value would be a torn 16-byte access. Acks travel the same way.
__global__ void __launch_bounds__(256) k_bcast16(int R, int n, char* buf, long long* cyc, unsigned long long* bad) {
    const int me = blockIdx.x;
    if (me > R)
        return;
    const int tid = threadIdx.x;
    long long t0 = 0;
    if (tid == 0)
        t0 = wall_clock64();
    char* tile = buf;                       // 16 KB
    char* acks = buf + 16 * 1024;           // one word per reader
    unsigned long long nbad = 0;
    for (int k = 1; k <= n; ++k) {
        if (me == 0) {
            for (int e = tid; e < 1024; e += 256)
                st16(tile + 16 * (size_t)e, 1e6 * k + e, k);
            if (tid < R) {
                v4i r;
                for (int it = 0; it < 40000000; ++it) {
                    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"(acks + 16 * (size_t)tid) : "memory");
                    if (r.z == k && r.w == ~k)
                        break;
                }
            }
            __syncthreads();
        } else {
            v4i r[4];
            for (int it = 0; it < 40000000; ++it) {
                ld16x4(tile, tid, r);
                bool all = true;
                for (int q = 0; q < 4; ++q)
                    all = all && r[q].z == k && r[q].w == ~k;
                if (all)
                    break;
            }
            for (int q = 0; q < 4; ++q)
                if (__hiloint2double(r[q].y, r[q].x) != 1e6 * k + (tid + 256 * q))
                    ++nbad;
            __syncthreads();
            if (tid == 0)
                st16(acks + 16 * (size_t)(me - 1), 0.0, k);
        }
    }
    if (nbad)
        atomicAdd(bad, nbad);
    if (tid == 0 && me == 0)
        cyc[0] = wall_clock64() - t0;
}

template <int MODE> int run(int* flags, double* buf, long long* cyc, double* out) {
    const int n = 2000;
    for (int payload = 0; payload < 2; ++payload)
        for (int partner : {8, 1, 4, 255}) {
            CK(hipMemset(flags, 0, 4096 * 4));
            hipLaunchKernelGGL(k_pingpong<MODE>, dim3(256), dim3(256), 0, 0, partner, n, payload, flags, buf, cyc, out);
            CK(hipDeviceSynchronize());
            long long hc[2];
            CK(hipMemcpy(hc, cyc, 16, hipMemcpyDeviceToHost));
            printf("mode %d payload %d partner wg %3d (XCD %d): ok=%lld  %.3f us per hop\n", MODE, payload, partner, partner % 8, hc[1], hc[0] / 100.0 / n / 2);
        }
    for (int R : {8, 32}) {
        CK(hipMemset(flags, 0, 4096 * 4));
        hipLaunchKernelGGL(k_bcast<MODE>, dim3(64), dim3(256), 0, 0, R, n, flags, buf, cyc, out);
        CK(hipDeviceSynchronize());
        long long hc[2];
        CK(hipMemcpy(hc, cyc, 16, hipMemcpyDeviceToHost));
        printf("mode %d broadcast 8 KB to %2d workgroups + acks: ok=%lld  %.3f us per round (two hops)\n", MODE, R, hc[1], hc[0] / 100.0 / n);
    }
    return 0;
}
int main() {
    int* flags;
    double *buf, *out;
    long long* cyc;
    CK(hipMalloc(&flags, 4096 * 4));
    CK(hipMalloc(&buf, 2048 * 8));
    CK(hipMalloc(&out, 256 * 256 * 8));
    CK(hipMalloc(&cyc, 16));
    if (run<0>(flags, buf, cyc, out) || run<1>(flags, buf, cyc, out) || run<2>(flags, buf, cyc, out))
        return 1;
    char* buf16;
    CK(hipMalloc(&buf16, 32 * 1024));
    for (int partner : {8, 1, 4, 255}) {
        CK(hipMemset(buf16, 0, 32 * 1024));
        const int n = 20000;
        hipLaunchKernelGGL(k_pingpong16, dim3(256), dim3(256), 0, 0, partner, n, buf16, cyc, out);
        CK(hipDeviceSynchronize());
        long long hc[2];
        CK(hipMemcpy(hc, cyc, 16, hipMemcpyDeviceToHost));
        printf("mode 3 (16-byte value+sequence pairs, no flag) 8 KB tile, partner wg %3d (XCD %d): ok=%lld  %.3f us per hop\n", partner, partner % 8, hc[1], hc[0] / 100.0 / n / 2);
    }
    unsigned long long* bad;
    CK(hipMalloc(&bad, 8));
    for (int R : {1, 8, 32}) {
        CK(hipMemset(buf16, 0, 32 * 1024));
        CK(hipMemset(bad, 0, 8));
        const int n = 50000;
        hipLaunchKernelGGL(k_bcast16, dim3(64), dim3(256), 0, 0, R, n, buf16, cyc, bad);
        CK(hipDeviceSynchronize());
        long long hc[2];
        unsigned long long hb;
        CK(hipMemcpy(hc, cyc, 16, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
        printf("mode 3 broadcast to %2d readers, %d rounds: %.3f us per round (two hops), words with the right sequence and a wrong value: %llu of %llu\n", R, n, hc[0] / 100.0 / n, hb,
               (unsigned long long)R * n * 1024ull);
    }
    return 0;
}
