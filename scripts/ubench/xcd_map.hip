// Which XCD does block b of a grid run on? HW_REG_XCC_ID of every block, for a sequence of launches whose grid sizes are not multiples of 8,
// with small and with CU-filling workgroups, on one stream and alternating between two streams.
// hipcc --offload-arch=gfx950 -O2 xcd_map.hip -o xcd_map_bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11)) & 0xf; }
template <int LDSB> __global__ void __launch_bounds__(512) k_map(int* out, int spin) {
    __shared__ char pad[LDSB];
    if (threadIdx.x == 0) {
        pad[0] = 1;
        out[blockIdx.x] = xcc_id() + (pad[0] ? 0 : 1);
        long long t0 = wall_clock64();
        while (wall_clock64() - t0 < spin) {}
    }
}
int main() {
    int* d; hipMalloc(&d, 4096 * 4);
    hipStream_t s[2]; hipStreamCreate(&s[0]); hipStreamCreate(&s[1]);
    std::vector<int> h(4096);
    const int grids[] = {64, 13, 200, 41, 66, 67, 200, 200, 5, 200};
    for (int big = 0; big < 2; ++big)
        for (int two = 0; two < 2; ++two) {
            printf("=== %s workgroups, %s\n", big ? "512-thread 64 KB-LDS" : "512-thread small", two ? "alternating between two streams" : "one stream");
            int li = 0;
            for (int g : grids) {
                hipStream_t st = s[two ? (li & 1) : 0];
                if (big) hipLaunchKernelGGL(k_map<65536>, dim3(g), dim3(512), 0, st, d, 300);
                else hipLaunchKernelGGL(k_map<16>, dim3(g), dim3(512), 0, st, d, 300);
                hipStreamSynchronize(st);
                hipMemcpy(h.data(), d, g * 4, hipMemcpyDeviceToHost);
                int off = h[0], bad = 0;
                for (int b = 0; b < g; ++b) if (h[b] != ((b + off) & 7)) ++bad;
                printf("grid %3d: block 0 on XCD %d; blocks not on XCD (b + %d) %% 8: %d;  first 16:", g, off, off, bad);
                for (int b = 0; b < 16 && b < g; ++b) printf(" %d", h[b]);
                printf("\n");
                ++li;
            }
        }
    // two kernels running at the same time on two streams: does the interleaving disturb the pattern inside a grid?
    printf("=== two 200-block grids in flight together (two streams), 64 KB LDS, 30 us per block\n");
    int* d2; hipMalloc(&d2, 4096 * 4);
    for (int rep = 0; rep < 4; ++rep) {
        hipLaunchKernelGGL(k_map<65536>, dim3(200), dim3(512), 0, s[0], d, 3000);
        hipLaunchKernelGGL(k_map<65536>, dim3(200), dim3(512), 0, s[1], d2, 3000);
        hipDeviceSynchronize();
        for (int which = 0; which < 2; ++which) {
            hipMemcpy(h.data(), which ? d2 : d, 200 * 4, hipMemcpyDeviceToHost);
            int off = h[0], bad = 0;
            for (int b = 0; b < 200; ++b) if (h[b] != ((b + off) & 7)) ++bad;
            printf("rep %d grid %c: block 0 on XCD %d, off-pattern blocks %d\n", rep, which ? 'B' : 'A', off, bad);
        }
    }
    return 0;
}
