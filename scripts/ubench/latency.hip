// Dependent-chain latencies of the instructions the register-resident pivot elimination is made of (one wave, shader cycles
// via s_memtime-free clock64() deltas of a 1000-iteration loop, 4 chained ops per iteration).
// build: hipcc -O3 --offload-arch=gfx950 -o latency_bin latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));
template <int J> __device__ __forceinline__ double row_bcast(double v) {
    double old;
    asm volatile("" : "=v"(old));
    return __builtin_amdgcn_update_dpp(old, v, 0x150 + J, 0xf, 0xf, false);
}
template <int L> __device__ __forceinline__ double read_lane(double v) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), L), __builtin_amdgcn_readlane(__double2loint(v), L));
}
__device__ __forceinline__ double swap32lo(double v) { // both halves of the wave receive the lower half's value
    const v2u h = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
    const v2u l = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
    return __hiloint2double((int)h.x, (int)l.x);
}
__device__ __forceinline__ double swap16lo(double v) {
    const v2u h = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
    const v2u l = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
    return __hiloint2double((int)h.x, (int)l.x);
}
template <int MODE> __global__ void k_lat(double* out, const double* in, int iters, long long* ticks) {
    double a = in[threadIdx.x];
    d4 acc = {a, a, a, a};
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE == 0) a = fma(a, 0.999, 1e-9);                               // v_fma_f64
            if (MODE == 1) a = __builtin_amdgcn_rcp(a) + 0.5;                      // v_rcp_f64 + v_add_f64
            if (MODE == 2) a = fma(row_bcast<5>(a), 0.999, 1e-9);                  // v_mov_b64_dpp + v_fma_f64
            if (MODE == 3) a = fma(read_lane<7>(a), 0.999, 1e-9);                  // 2 v_readlane + v_fma_f64 (SGPR operand)
            if (MODE == 4) a = fma(swap32lo(a), 0.999, 1e-9);                      // 2 v_permlane32_swap + fma
            if (MODE == 5) a = fma(swap16lo(swap32lo(a)), 0.999, 1e-9);            // 4 swaps + fma (row all-gather depth)
            if (MODE == 6) a = __builtin_amdgcn_rsq(a) + 0.5;                      // v_rsq_f64 + add
            if (MODE == 7) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc, 0, 0, 0); a = acc[0] * 1e-3; } // mfma -> valu -> mfma operand
            if (MODE == 8) a = a * 0.999;                                          // v_mul_f64
            if (MODE == 9) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(0.001, 0.002, acc, 0, 0, 0); }  // mfma accumulate chain
        }
    }
    const long long t1 = clock64();
    out[threadIdx.x] = a + acc[0] + acc[1] + acc[2] + acc[3];
    if (threadIdx.x == 0)
        *ticks = t1 - t0;
}
template <int MODE> static void run(const char* name, double* dout, double* di, long long* dt) {
    long long ht;
    for (int rep = 0; rep < 2; ++rep) {
        k_lat<MODE><<<1, 64>>>(dout, di, 1000, dt);
        (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(&ht, dt, 8, hipMemcpyDeviceToHost);
    printf("%-48s %.1f clock64 ticks per link\n", name, ht / 4000.0);
}
int main() {
    double h[64], *di, *dout;
    long long* dt;
    for (int i = 0; i < 64; ++i) h[i] = 1.0 + i * 1e-3;
    (void)hipMalloc(&di, 512); (void)hipMalloc(&dout, 2048); (void)hipMalloc(&dt, 8);
    (void)hipMemcpy(di, h, 512, hipMemcpyHostToDevice);
    run<0>("v_fma_f64", dout, di, dt);
    run<8>("v_mul_f64", dout, di, dt);
    run<1>("v_rcp_f64 + v_add_f64", dout, di, dt);
    run<6>("v_rsq_f64 + v_add_f64", dout, di, dt);
    run<2>("v_mov_b64_dpp row_newbcast + v_fma_f64", dout, di, dt);
    run<3>("2 v_readlane + v_fma_f64", dout, di, dt);
    run<4>("2 v_permlane32_swap + v_fma_f64", dout, di, dt);
    run<5>("2 permlane32 + 2 permlane16 swaps + v_fma_f64", dout, di, dt);
    run<7>("mfma_f64_16x16x4 -> v_mul_f64 -> mfma operand", dout, di, dt);
    run<9>("mfma_f64_16x16x4 accumulate chain", dout, di, dt);
    return 0;
}
