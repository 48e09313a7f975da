// Cycles of one 16x16 elimination (ldl16_inverse_wave) and of the whole 32x32 tile inverse on one wave, operands already on chip
// (the tile ubench in ubench.hip includes the global loads of the tile).
// build: hipcc -O3 --offload-arch=gfx950 -I../../include -o ldl16_probe_bin ldl16_probe.hip   (-DKERNELS_HPP=... to time another version of the header)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#ifndef KERNELS_HPP
#define KERNELS_HPP "../../eqvio_amd/csrc/eqf_kernels.hpp"
#endif
#include KERNELS_HPP
using namespace eqf;
#define FENCE() do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
__global__ void __launch_bounds__(256) k_probe(const double* D, double* Linv, double* Linv32, int* flags, long long* cyc) {
    __shared__ double sX[16 * 17];
    __shared__ double sD[32 * 33];
    __shared__ double swork[LDL_SBUF];
    const int lane = threadIdx.x & 63, r = lane & 15, cq = lane >> 4;
    {
        const int rr = threadIdx.x & 31, g = threadIdx.x >> 5;
        for (int k = 0; k < 4; ++k) { const int c = g + 8 * k; sD[rr + c * 33] = D[rr + 32 * c]; }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        double a[4], o[4];
        for (int k = 0; k < 4; ++k) a[k] = sD[max(r, cq + 4 * k) + 33 * min(r, cq + 4 * k)];
        FENCE();
        const long long w0 = clock64();
        FENCE();
        ldl16_inverse_wave(a, o, flags, true, sX);
        FENCE();
        const long long w1 = clock64();
        FENCE();
        for (int k = 0; k < 4; ++k) Linv[r + 16 * (cq + 4 * k)] = o[k];
        if (lane == 0) cyc[0] = w1 - w0;
    }
    __syncthreads();
    FENCE();
    const long long t0 = clock64();
    FENCE();
    ldl_inverse_tile_put(sD, 33, 32, [&](int rr, int cc, double v) { sD[rr + 33 * cc] = v; }, flags, swork);
    FENCE();
    const long long t1 = clock64();
    FENCE();
    if (threadIdx.x == 0) cyc[1] = t1 - t0;
    __syncthreads();
    {
        const int rr = threadIdx.x & 31, g = threadIdx.x >> 5;
        for (int k = 0; k < 4; ++k) { const int c = g + 8 * k; Linv32[rr + 32 * c] = sD[rr + c * 33]; }
    }
}
int main(int argc, char** argv) {
    double *dD, *dL, *dL32; int* df; long long* dc;
    (void)hipMalloc(&dD, 8192); (void)hipMalloc(&dL, 2048); (void)hipMalloc(&dL32, 8192); (void)hipMalloc(&df, 16); (void)hipMalloc(&dc, 64);
    (void)hipMemset(df, 0, 16);
    std::vector<double> h(1024);
    const int kind = argc > 1 ? atoi(argv[1]) : 0;
    if (kind == 0) {
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) h[i + 32 * j] = (i == j ? 40.0 : 0.0) + 1.0 / (1 + abs(i - j));
    } else { // B B^T + ridge, B 32 x 6: condition ~ 1 / ridge ; kind 2: the last 10 rows / columns identity padding
        double B[32][6]; unsigned long long st = 88172645463325252ull;
        auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0 - 0.5; };
        for (auto& row : B) for (auto& v : row) v = rnd() * 30.0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double sacc = (i == j) ? 1e-5 : 0.0; for (int k = 0; k < 6; ++k) sacc += B[i][k] * B[j][k]; h[i + 32 * j] = sacc; }
        if (kind == 2) for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) if (i >= 22 || j >= 22) h[i + 32 * j] = (i == j) ? 1.0 : 0.0;
    }
    (void)hipMemcpy(dD, h.data(), 8192, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) {
        k_probe<<<1, 256>>>(dD, dL, dL32, df, dc);
        (void)hipDeviceSynchronize();
        long long c[2]; (void)hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost);
        printf("ldl16_inverse_wave %lld cycles ; ldl_inverse_tile (32x32, LDS to LDS) %lld cycles\n", c[0], c[1]);
    }
    std::vector<double> hl(1024); (void)hipMemcpy(hl.data(), dL32, 8192, hipMemcpyDeviceToHost);
    double worst = 0; static long double t[32][32];
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { long double s = 0; for (int k = 0; k < 32; ++k) s += (long double)hl[i + 32 * k] * h[k + 32 * j]; t[i][j] = s; }
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { long double s = 0; for (int k = 0; k < 32; ++k) s += t[i][k] * hl[j + 32 * k]; worst = std::max(worst, (double)fabsl(s - (i == j))); }
    int hf[4]; (void)hipMemcpy(hf, df, 16, hipMemcpyDeviceToHost);
    double up = 0; for (int i = 0; i < 32; ++i) for (int j = i + 1; j < 32; ++j) up = std::max(up, fabs(hl[i + 32 * j]));
    printf("check: max |Linv A Linv^T - I| = %.2e, max |upper| %.2e, flag %d\n", worst, up, hf[0]);
    return 0;
}
