// Broadcast of one 16-lane row of a wave to all four rows (what the 16x16 elimination needs for its multiplier column):
// ds_bpermute (LDS crossbar) against the gfx950 pair v_permlane32_swap + v_permlane16_swap (VALU). Semantics check + dependent-chain latency.
// build: hipcc -O3 --offload-arch=gfx950 -o rowcast_bin rowcast.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
template <int Q> __device__ __forceinline__ unsigned rowcast32(unsigned a) {
    const v2u r = __builtin_amdgcn_permlane32_swap(a, a, false, false); // r.x = [lo, lo], r.y = [hi, hi] (halves of 32 lanes)
    const unsigned h = (Q & 2) ? r.y : r.x;                               // rows [q0, q1, q0, q1] of the wanted half
    const v2u s = __builtin_amdgcn_permlane16_swap(h, h, false, false);  // s.x = [q0 x4], s.y = [q1 x4]
    return (Q & 1) ? s.y : s.x;
}
template <int Q> __device__ __forceinline__ double rowcast(double v) {
    return __hiloint2double((int)rowcast32<Q>((unsigned)__double2hiint(v)), (int)rowcast32<Q>((unsigned)__double2loint(v)));
}
__device__ __forceinline__ double fetch_lane(double v, int byte_addr) {
    return __hiloint2double(__builtin_amdgcn_ds_bpermute(byte_addr, __double2hiint(v)), __builtin_amdgcn_ds_bpermute(byte_addr, __double2loint(v)));
}
__global__ void k_sem(double* out, const double* in) {
    const double a = in[threadIdx.x];
    out[threadIdx.x] = rowcast<0>(a);
    out[64 + threadIdx.x] = rowcast<1>(a);
    out[128 + threadIdx.x] = rowcast<2>(a);
    out[192 + threadIdx.x] = rowcast<3>(a);
}
template <int MODE> __global__ void k_lat(double* out, const double* in, int iters, long long* ticks) {
    double a = in[threadIdx.x];
    const int r = threadIdx.x & 15;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        // chain: broadcast row (i & 3 fixed per unrolled slot), one fma on it
        if (MODE == 0) {
            a = fma(fetch_lane(a, 4 * (r + 16 * 0)), 0.999, 1e-9);
            a = fma(fetch_lane(a, 4 * (r + 16 * 1)), 0.999, 1e-9);
            a = fma(fetch_lane(a, 4 * (r + 16 * 2)), 0.999, 1e-9);
            a = fma(fetch_lane(a, 4 * (r + 16 * 3)), 0.999, 1e-9);
        } else {
            a = fma(rowcast<0>(a), 0.999, 1e-9);
            a = fma(rowcast<1>(a), 0.999, 1e-9);
            a = fma(rowcast<2>(a), 0.999, 1e-9);
            a = fma(rowcast<3>(a), 0.999, 1e-9);
        }
    }
    const long long t1 = clock64();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0)
        *ticks = t1 - t0;
}
int main() {
    double h[64], o[256], *di, *dout;
    long long* dt, ht;
    for (int i = 0; i < 64; ++i) h[i] = i;
    (void)hipMalloc(&di, 512); (void)hipMalloc(&dout, 2048); (void)hipMalloc(&dt, 8);
    (void)hipMemcpy(di, h, 512, hipMemcpyHostToDevice);
    k_sem<<<1, 64>>>(dout, di);
    (void)hipMemcpy(o, dout, 2048, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int q = 0; q < 4; ++q)
        for (int i = 0; i < 64; ++i)
            if (o[64 * q + i] != 16 * q + (i & 15)) ++bad;
    printf("rowcast semantics: %s\n", bad ? "WRONG" : "ok");
    if (bad) { for (int q = 0; q < 4; ++q) { for (int i = 0; i < 64; ++i) printf("%g ", o[64 * q + i]); printf("\n"); } }
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) k_lat<0><<<1, 64>>>(dout, di, 1000, dt); else k_lat<1><<<1, 64>>>(dout, di, 1000, dt);
            (void)hipDeviceSynchronize();
        }
        (void)hipMemcpy(&ht, dt, 8, hipMemcpyDeviceToHost);
        printf("%s: %.1f clock64 ticks per broadcast+fma (s_memtime 100 MHz: x 24 for core cycles at 2.4 GHz)\n", mode ? "permlane swaps" : "ds_bpermute   ", ht / 4000.0);
    }
    return bad;
}
