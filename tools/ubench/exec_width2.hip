// exec_width.hip found: a VALU instruction takes ~3x as long when at most 8 lanes are enabled.  Is that a steady-state
// effect only?  Runs of S instructions under a 4-lane mask alternate with runs of S instructions under the full mask.
//   hipcc -O3 --offload-arch=gfx950 -o exec_width2 exec_width2.hip && ./exec_width2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) k(double* out, unsigned long long* t, int outer, int runs,
                                                                                  unsigned long long mask) {
    extern __shared__ double pad[];
    const int lane = threadIdx.x;
    if (outer < 0) pad[lane] = 1.0;
    double v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = 1.0 + 1e-3 * (lane + j);
    const double a = 1.0000001, b = 1e-9;
    const bool sparse = (mask >> lane) & 1ull;
    const unsigned long long c0 = clock64();
    for (int o = 0; o < outer; ++o) {
        if (sparse) {
            for (int i = 0; i < runs; ++i) {
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = __builtin_fma(v[j], a, b);
            }
        }
        for (int i = 0; i < runs; ++i) {
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = __builtin_fma(v[j], a, b);
        }
    }
    const unsigned long long c1 = clock64();
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += v[j];
    out[(size_t)blockIdx.x * 64 + lane] = s;
    if (lane == 0) t[blockIdx.x] = c1 - c0;
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int blocks = p.multiProcessorCount * 4, lds = 40 * 1024;
    double* out; unsigned long long* t;
    CK(hipMalloc(&out, sizeof(double) * blocks * 64));
    CK(hipMalloc(&t, sizeof(unsigned long long) * blocks));
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    printf("runs of S fp64 FMAs under a mask alternate with runs of S under the full mask; cycles per instruction of the masked runs\n"
           "(total - S x 4.81 for the full runs)\n");
    for (unsigned long long mask : {0xFull, 0xFFFFull}) {
        printf("mask of %d lanes:", mask == 0xFull ? 4 : 16);
        for (int runs : {1, 2, 4, 8, 16, 64, 256, 4096}) {
            const int S = runs * 16, outer = (1 << 22) / S;
            for (int rep = 0; rep < 2; ++rep) {
                k<<<blocks, 64, lds>>>(out, t, outer, runs, mask);
                CK(hipDeviceSynchronize());
            }
            std::vector<unsigned long long> h(blocks);
            CK(hipMemcpy(h.data(), t, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost));
            std::sort(h.begin(), h.end());
            const double per_pair = (double)h[blocks / 2] / ((double)outer * S);
            printf("  S=%d: %.2f", S, per_pair - 4.81);
        }
        printf("\n");
    }
    return 0;
}
