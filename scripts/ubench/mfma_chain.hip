// fp64 MFMA 16x16x4 on ONE wave: cycles per instruction for a dependent chain (each accumulates into the previous result), for 2 / 4 interleaved independent
// chains, and with the A operand coming from LDS (ds_read_b64 in front of every MFMA) as in the look-ahead owner's tail.
// hipcc --offload-arch=gfx950 -O2 mfma_chain.hip -o mfma_chain_bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int CH, bool LDS> __global__ void __launch_bounds__(64) k(double* out, long long* cyc, int n) {
    __shared__ double sm[64 * 8];
    const int lane = threadIdx.x;
    for (int i = 0; i < 8; ++i) sm[lane + 64 * i] = 1e-3 * (lane + i);
    __syncthreads();
    d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    double a = 1.0 + 1e-6 * lane, b = 1.0 - 1e-6 * lane;
    long long t0 = clock64();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int st = 0; st < 8; ++st) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const double av = LDS ? sm[lane + 64 * ((st + c) & 7)] : a;
                acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b, acc[c], 0, 0, 0);
            }
        }
    }
    long long t1 = clock64();
    double s = 0;
    for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[lane] = s;
    if (lane == 0) cyc[0] = t1 - t0;
}
template <int CH, bool LDS> void run(double* out, long long* cyc) {
    const int n = 2000;
    hipLaunchKernelGGL((k<CH, LDS>), dim3(1), dim3(64), 0, 0, out, cyc, n);
    (void)hipDeviceSynchronize();
    long long c;
    (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%d chain(s), A operand from %s: %.1f shader-clock cycles per MFMA (clock64 counts at 100 MHz x ... see ratio) raw %lld for %d MFMAs\n", CH, LDS ? "LDS " : "regs", (double)c / (n * 8.0 * CH), c, n * 8 * CH);
}
int main() {
    double* out; long long* cyc;
    (void)hipMalloc(&out, 64 * 8); (void)hipMalloc(&cyc, 8);
    run<1, false>(out, cyc); run<2, false>(out, cyc); run<4, false>(out, cyc);
    run<1, true>(out, cyc); run<2, true>(out, cyc); run<4, true>(out, cyc);
    return 0;
}
