// Micro-benchmarks that feed DESIGN.md: fp64 MFMA vs VALU peak, pivot-chain cost, barrier/LDS round trips, launch gap.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include "../../eqvio_amd/csrc/eqf_kernels.hpp"
using namespace eqf;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_valu_peak(int iters, double* out) {
    double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double x = 1.0000001, y = 1e-9;
    for (int i = 0; i < iters; ++i) {
        a0 = fma(a0, x, y); a1 = fma(a1, x, y); a2 = fma(a2, x, y); a3 = fma(a3, x, y);
        a4 = fma(a4, x, y); a5 = fma(a5, x, y); a6 = fma(a6, x, y); a7 = fma(a7, x, y);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
__global__ void __launch_bounds__(256) k_ldl_cycles(int w, int ldz, const double* Z, double* Linv, int* flags, long long* cyc) {
    __shared__ double sD[32 * 33];
    __shared__ double swork[LDL_SBUF];
    const int r = threadIdx.x & 31, g = threadIdx.x >> 5;
    for (int k = 0; k < 4; ++k) { const int c = g + 8 * k; sD[r + c * 33] = (r < w && c < w && r >= c) ? Z[r + (size_t)c * ldz] : ((r == c) ? 1.0 : 0.0); }
    __syncthreads();
    const long long t0 = clock64();
    const long long w0 = wall_clock64();
    ldl_inverse_tile(sD, 33, w, Linv, flags, swork);
    __syncthreads();
    if (threadIdx.x == 0) { cyc[0] = clock64() - t0; cyc[1] = wall_clock64() - w0; }
}
__global__ void __launch_bounds__(256) k_barrier_cycles(int n, long long* cyc, double* out) {
    __shared__ double s[256];
    double v = threadIdx.x;
    __syncthreads();
    const long long t0 = clock64();
    for (int i = 0; i < n; ++i) { s[threadIdx.x] = v; __syncthreads(); v = s[(threadIdx.x + 33) & 255] + 1.0; }
    if (threadIdx.x == 0) cyc[0] = clock64() - t0;
    out[threadIdx.x] = v;
}
__global__ void __launch_bounds__(64) k_rcp_chain(int n, long long* cyc, double* out) {
    double v = 1.5 + threadIdx.x * 1e-3;
    const long long t0 = clock64();
    for (int i = 0; i < n; ++i) v = fast_rcp(v) + 0.7;
    if (threadIdx.x == 0) cyc[0] = clock64() - t0;
    out[threadIdx.x] = v;
}
__global__ void k_empty() {}

int main() {
    double* d_out; long long* d_cyc; int* d_flags; double *d_Z, *d_L;
    CK(hipMalloc(&d_out, sizeof(double) * 2048 * 256)); CK(hipMalloc(&d_cyc, 64)); CK(hipMalloc(&d_flags, 16)); CK(hipMemset(d_flags, 0, 16)); CK(hipMalloc(&d_Z, 8 * 1024)); CK(hipMalloc(&d_L, 8 * 1024));
    std::vector<double> hz(1024, 0.0);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) hz[i + 32 * j] = (i == j ? 40.0 : 0.0) + 1.0 / (1 + abs(i - j));
    CK(hipMemcpy(d_Z, hz.data(), 8192, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    // MFMA vs VALU fp64 peak
    for (int which = 0; which < 2; ++which) {
        const int nblk = 2048, iters = 4096; double best = 0;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(k_mfma_peak, dim3(nblk), dim3(256), 0, 0, iters, d_out, (unsigned long long*)nullptr);
            else hipLaunchKernelGGL(k_valu_peak, dim3(nblk), dim3(256), 0, 0, iters, d_out);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            const double flops = which == 0 ? (double)nblk * 4 * iters * 4 * 2048.0 : (double)nblk * 256 * iters * 8 * 2.0;
            best = std::max(best, flops / (ms * 1e-3) / 1e12);
        }
        printf("%s fp64 peak: %.1f TFLOP/s\n", which == 0 ? "MFMA 16x16x4" : "VALU v_fma_f64", best);
    }
    long long hc[8];
    // pivot chain
    for (int rep = 0; rep < 6; ++rep) {
        const int wv[6] = {32, 32, 16, 8, 4, 2};
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_ldl_cycles, dim3(1), dim3(256), 0, 0, wv[rep], 32, d_Z, d_L, d_flags, d_cyc);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        CK(hipMemcpy(hc, d_cyc, 16, hipMemcpyDeviceToHost));
        if (rep < 2) { // Linv A Linv^T = I ?
            std::vector<double> hl(1024); CK(hipMemcpy(hl.data(), d_L, 8192, hipMemcpyDeviceToHost));
            double worst = 0; long double t[32][32];
            for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { long double s = 0; for (int k = 0; k < 32; ++k) s += (long double)hl[i + 32 * k] * hz[k + 32 * j]; t[i][j] = s; }
            for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { long double s = 0; for (int k = 0; k < 32; ++k) s += t[i][k] * hl[j + 32 * k]; worst = std::max(worst, (double)fabsl(s - (i == j))); }
            double up = 0; for (int i = 0; i < 32; ++i) for (int j = i + 1; j < 32; ++j) up = std::max(up, fabs(hl[i + 32 * j]));
            int hf; CK(hipMemcpy(&hf, d_flags, 4, hipMemcpyDeviceToHost));
            printf("  check: max |Linv A Linv^T - I| = %.2e, max |upper| = %.2e, flag %d\n", worst, up, hf);
        }
        printf("ldl_inverse_tile(w=%d): %lld shader cycles, %lld wall ticks (100MHz) = %.2f us ; event span %.2f us -> %.0f cycles/pivot, clock ~%.2f GHz\n", wv[rep], hc[0], hc[1], hc[1] / 100.0, ms * 1e3, hc[0] / 32.0, hc[0] / (hc[1] * 10.0));
    }
    hipLaunchKernelGGL(k_barrier_cycles, dim3(1), dim3(256), 0, 0, 1000, d_cyc, d_out); CK(hipMemcpy(hc, d_cyc, 8, hipMemcpyDeviceToHost));
    printf("LDS write + barrier + LDS read round trip (4 waves): %.0f cycles\n", hc[0] / 1000.0);
    hipLaunchKernelGGL(k_rcp_chain, dim3(1), dim3(64), 0, 0, 1000, d_cyc, d_out); CK(hipMemcpy(hc, d_cyc, 8, hipMemcpyDeviceToHost));
    printf("fast_rcp + add dependent chain: %.0f cycles\n", hc[0] / 1000.0);
    // launch gap: 200 empty kernels back to back
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("empty kernel back-to-back: %.2f us per launch\n", ms * 1e3 / 200);
    return 0;
}
