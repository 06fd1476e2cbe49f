// Does the time of a VALU instruction depend on HOW MANY lanes the exec mask enables?  One wavefront per SIMD (LDS-forced),
// a chain-free stream of 64 instructions per iteration, `active` lanes enabled (contiguous from lane 0, or strided).
//   hipcc -O3 --offload-arch=gfx950 -o exec_width exec_width.hip && ./exec_width
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int OP>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) op_kernel(double* out, unsigned long long* t, int iters,
                                                                                          unsigned long long mask) {
    extern __shared__ double pad[];
    const int lane = threadIdx.x;
    if (iters < 0) pad[lane] = 1.0;
    double v[16];
    float f[16];
    int n[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { v[k] = 1.0 + 1e-3 * (lane + k); f[k] = 1.0f + 1e-3f * (lane + k); n[k] = lane + k; }
    const double a = 1.0000001, b = 1e-9;
    const unsigned long long c0 = clock64();
    if ((mask >> lane) & 1ull) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    if (OP == 0) v[k] = __builtin_fma(v[k], a, b);
                    if (OP == 1) f[k] = __builtin_fmaf(f[k], 1.0000001f, 1e-9f);
                    if (OP == 2) n[k] = n[k] * 3 + 1;
                    if (OP == 3) v[k] = v[k] + b;
                    if (OP == 4) asm volatile("v_mov_b64_e32 %0, %1" : "=v"(v[k]) : "v"(v[(k + 1) & 15]));
                }
            }
        }
    }
    const unsigned long long c1 = clock64();
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += v[k] + f[k] + n[k];
    out[(size_t)blockIdx.x * 64 + lane] = s;
    if (lane == 0) t[blockIdx.x] = c1 - c0;
}

template <int OP>
int run(const char* name, int cus, int iters) {
    const int blocks = cus * 4;
    double* out; unsigned long long* t;
    CK(hipMalloc(&out, sizeof(double) * blocks * 64));
    CK(hipMalloc(&t, sizeof(unsigned long long) * blocks));
    const int lds = 40 * 1024;
    CK(hipFuncSetAttribute((const void*)op_kernel<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    printf("%-14s cycles per instruction (median wavefront) with lanes enabled:", name);
    const unsigned long long masks[] = {~0ull, 0x7FFFFFFFFFFFFFFFull, 0xFFFFFFFFull, 0x1FFFFull, 0xFFFFull, 0x7FFFull, 0xFFFull, 0x3FFull, 0x1FFull, 0xFFull, 0xFull, 0x1ull,
                                        0x0001000100010001ull, 0x0101010101010101ull, 0x1111111111111111ull, 0x5555555555555555ull, 0x00FF00FF00FF00FFull,
                                        0xFFFF00000000FFFFull, 0x1ull << 63};
    const char* names[] = {"64", "63", "32", "17", "16", "15", "12", "10", "9", "8", "4", "1", "4 (1 per row)", "8 (2 per row)", "16 (4 per row)", "32 (8 per row)",
                           "32 (8 per row, contiguous)", "32 (rows 0, 3)", "1 (lane 63)"};
    const int nm = sizeof(masks) / sizeof(masks[0]);
    for (int m = 0; m < nm; ++m) {
        for (int rep = 0; rep < 2; ++rep) {
            op_kernel<OP><<<blocks, 64, lds>>>(out, t, iters, masks[m]);
            CK(hipDeviceSynchronize());
        }
        std::vector<unsigned long long> h(blocks);
        CK(hipMemcpy(h.data(), t, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        printf("  %s: %.2f", names[m], (double)h[blocks / 2] / ((double)iters * 64));
    }
    printf("\n");
    CK(hipFree(out)); CK(hipFree(t));
    return 0;
}

#include <algorithm>
int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount, iters = 100000;
    if (run<0>("v_fma_f64", cus, iters)) return 1;
    if (run<1>("v_fma_f32", cus, iters)) return 1;
    if (run<2>("v_mad_u32", cus, iters)) return 1;
    if (run<4>("v_mov_b64", cus, iters)) return 1;
    return 0;
}
