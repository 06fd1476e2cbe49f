// fp64 VALU issue-rate microbenchmark: independent FMA / ADD / MUL chains, wave64.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int OP>
__global__ void __launch_bounds__(256) k(double* out, int iters, double a, double b) {
    double v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = a + i + threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (OP == 0) v[i] = fma(v[i], a, b);
            if (OP == 1) v[i] = v[i] + b;
            if (OP == 2) v[i] = v[i] * a;
            if (OP == 3) v[i] = __builtin_fmaf((float)v[i], (float)a, (float)b);
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    double* d; hipMalloc(&d, sizeof(double) * 256 * 8192);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[] = {"fma_f64", "add_f64", "mul_f64", "fma_f32(cvt)"};
    for (int op = 0; op < 3; ++op)
        for (int blocks_per_cu : {1, 2, 4}) {
            const int iters = 20000, grid = 256 * blocks_per_cu;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (op == 0) k<0><<<grid, 256>>>(d, iters, 1.0000001, 1e-9);
                if (op == 1) k<1><<<grid, 256>>>(d, iters, 1.0000001, 1e-9);
                if (op == 2) k<2><<<grid, 256>>>(d, iters, 1.0000001, 1e-9);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep == 1) {
                    const double inst = (double)grid * 4 /*waves*/ * iters * 16;
                    const double per_simd_inst = inst / 1024.0;
                    printf("%s waves/SIMD=%d  %.3f ms  %.2f Ginst/s/SIMD-> cycles/inst @2.4GHz = %.2f   TFLOP/s(2flop)=%.1f\n",
                           names[op], blocks_per_cu, ms, per_simd_inst / ms / 1e6, 2.4e9 * ms * 1e-3 / per_simd_inst,
                           inst * 64 * 2 / ms / 1e9);
                }
            }
        }
    return 0;
}
