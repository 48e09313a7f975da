// Single-wave ISSUE cost (shader cycles per instruction, 8 independent streams so that no instruction waits for an operand).
// build: hipcc -O3 --offload-arch=gfx950 -o issue_bin issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int J> __device__ __forceinline__ double row_bcast(double v) {
    double old;
    asm volatile("" : "=v"(old));
    return __builtin_amdgcn_update_dpp(old, v, 0x150 + J, 0xf, 0xf, false);
}
template <int MODE> __global__ void k_issue(double* out, const double* in, int iters, long long* ticks) {
    double a[8];
    for (int i = 0; i < 8; ++i) a[i] = in[threadIdx.x] + i;
    const bool odd = threadIdx.x & 1;
    const int addr = 4 * ((threadIdx.x + 16) & 63);
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) a[i] = fma(a[i], 0.999, 1e-9);
            if (MODE == 1) a[i] = a[i] * 0.999;
            if (MODE == 2) a[i] = row_bcast<3>(a[i]);
            if (MODE == 3) { int lo = __double2loint(a[i]); lo = __builtin_amdgcn_update_dpp(lo, lo, 0xA0, 0xf, 0xf, false); a[i] = __hiloint2double(__double2hiint(a[i]), lo); }
            if (MODE == 4) { int lo = __double2loint(a[i]); asm volatile("v_cndmask_b32 %0, 0, %0, vcc" : "+v"(lo)); a[i] = __hiloint2double(__double2hiint(a[i]), lo); }
            if (MODE == 5) { int lo = __double2loint(a[i]); lo = __builtin_amdgcn_ds_bpermute(addr, lo); a[i] = __hiloint2double(__double2hiint(a[i]), lo); }
            if (MODE == 6) { int lo = __double2loint(a[i]); lo = __builtin_amdgcn_readlane(lo, 5); a[i] = __hiloint2double(__double2hiint(a[i]), lo); }
            if (MODE == 7) { float f = __int_as_float(__double2loint(a[i])); f = fmaf(f, 0.999f, 1e-9f); a[i] = __hiloint2double(__double2hiint(a[i]), __float_as_int(f)); }
            if (MODE == 8) a[i] = __builtin_amdgcn_rcp(a[i]);
        }
    }
    const long long t1 = clock64();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) *ticks = t1 - t0;
}
typedef double d4 __attribute__((ext_vector_type(4)));
// independent fp64 MFMAs of ONE wave (NACC accumulators round robin), optionally with VALU FMAs in between
template <int NACC, int NVALU> __global__ void k_mfma_issue(double* out, const double* in, int iters, long long* ticks) {
    d4 acc[NACC];
    double v[4];
    const double x = in[threadIdx.x];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{x, x, x, x};
    for (int i = 0; i < 4; ++i) v[i] = x + i;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, 0.001, acc[i], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NVALU; ++j) v[j & 3] = fma(v[j & 3], 0.999, 1e-9);
        }
    }
    const long long t1 = clock64();
    double s = v[0] + v[1] + v[2] + v[3];
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) *ticks = (t1 - t0) / NACC;
}
template <int NACC, int NVALU> static void run_mfma(double* dout, double* di, long long* dt) {
    long long ht;
    for (int rep = 0; rep < 2; ++rep) { k_mfma_issue<NACC, NVALU><<<1, 64>>>(dout, di, 500, dt); (void)hipDeviceSynchronize(); }
    (void)hipMemcpy(&ht, dt, 8, hipMemcpyDeviceToHost);
    printf("mfma_f64_16x16x4, %d independent accumulators, %d VALU FMAs after each: %.1f cycles per MFMA\n", NACC, NVALU, ht / 500.0);
}
template <int MODE> static void run(const char* name, double* dout, double* di, long long* dt) {
    long long ht;
    for (int rep = 0; rep < 2; ++rep) { k_issue<MODE><<<1, 64>>>(dout, di, 500, dt); (void)hipDeviceSynchronize(); }
    (void)hipMemcpy(&ht, dt, 8, hipMemcpyDeviceToHost);
    printf("%-40s %.2f cycles per instruction (one wave, 8 independent streams)\n", name, ht / 4000.0);
}
int main() {
    double h[64], *di, *dout; long long* dt;
    for (int i = 0; i < 64; ++i) h[i] = 1.0 + i * 1e-3;
    (void)hipMalloc(&di, 512); (void)hipMalloc(&dout, 2048); (void)hipMalloc(&dt, 8);
    (void)hipMemcpy(di, h, 512, hipMemcpyHostToDevice);
    run<0>("v_fma_f64", dout, di, dt);
    run<1>("v_mul_f64", dout, di, dt);
    run<2>("v_mov_b64_dpp row_newbcast", dout, di, dt);
    run<3>("v_mov_b32_dpp quad_perm", dout, di, dt);
    run<4>("v_cndmask_b32", dout, di, dt);
    run<5>("ds_bpermute_b32", dout, di, dt);
    run<6>("v_readlane_b32 (+ v_mov back)", dout, di, dt);
    run<7>("v_fma_f32", dout, di, dt);
    run<8>("v_rcp_f64", dout, di, dt);
    run_mfma<1, 0>(dout, di, dt);
    run_mfma<2, 0>(dout, di, dt);
    run_mfma<4, 0>(dout, di, dt);
    run_mfma<4, 4>(dout, di, dt);
    run_mfma<4, 8>(dout, di, dt);
    run_mfma<2, 8>(dout, di, dt);
    return 0;
}
