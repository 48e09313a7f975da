// Accuracy of v_rcp_f64 / v_rsq_f64 followed by 0, 1, 2 Newton steps, in units of the last place against the correctly rounded result.
// build: hipcc -O3 --offload-arch=gfx950 -o rcp_precision_bin rcp_precision.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
__global__ void k(const double* x, double* o, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double d = x[i];
    double r = __builtin_amdgcn_rcp(d);
    o[i] = r;
    double e = fma(-d, r, 1.0); r = fma(r, e, r);
    o[n + i] = r;
    e = fma(-d, r, 1.0); r = fma(r, e, r);
    o[2 * n + i] = r;
    double y = __builtin_amdgcn_rsq(d);
    o[3 * n + i] = y;
    const double h = 0.5 * d;
    y = y * fma(-h * y, y, 1.5);
    o[4 * n + i] = y;
    y = y * fma(-h * y, y, 1.5);
    o[5 * n + i] = y;
}
int main() {
    const int n = 1 << 20;
    std::vector<double> hx(n), ho(6 * n);
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> u(-30, 30);
    for (auto& v : hx) v = std::exp2(u(g)) * (1.0 + (g() >> 11) * 0x1p-53);
    double *dx, *dout;
    (void)hipMalloc(&dx, n * 8); (void)hipMalloc(&dout, 6 * n * 8);
    (void)hipMemcpy(dx, hx.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, dout, n);
    (void)hipMemcpy(ho.data(), dout, 6 * n * 8, hipMemcpyDeviceToHost);
    const char* names[6] = {"v_rcp_f64", "v_rcp_f64 + 1 Newton", "v_rcp_f64 + 2 Newton", "v_rsq_f64", "v_rsq_f64 + 1 Newton", "v_rsq_f64 + 2 Newton"};
    for (int m = 0; m < 6; ++m) {
        double worst = 0;
        for (int i = 0; i < n; ++i) {
            const long double ref = m < 3 ? 1.0L / hx[i] : 1.0L / sqrtl((long double)hx[i]);
            const double ulp = std::ldexp(1.0, std::ilogb((double)ref) - 52);
            worst = std::max(worst, (double)(fabsl(ho[(size_t)m * n + i] - ref) / ulp));
        }
        printf("%-24s max error %.3g ulp\n", names[m], worst);
    }
    return 0;
}
