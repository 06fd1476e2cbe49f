// fp64 FMA issue rate by operand kind (VGPR vs SGPR/constant sources), wave64, gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void __launch_bounds__(256) k(double* out, int iters, double a, double b) {
    double v[8], w[8], u[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = a + i + threadIdx.x * 1e-9; w[i] = 1.0 + 1e-9 * (i + threadIdx.x); u[i] = 1e-9 * i * threadIdx.x; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) v[i] = fma(v[i], a, b);             // 1 VGPR + 2 SGPR
            if (MODE == 1) v[i] = fma(v[i], w[i], b);          // 2 VGPR + 1 SGPR
            if (MODE == 2) v[i] = fma(v[i], w[i], u[i]);       // 3 VGPR (accumulate into first)
            if (MODE == 3) v[i] = fma(w[i], u[i], v[i]);       // 3 VGPR, fmac form
            if (MODE == 4) v[i] = fma(w[i], u[(i + 3) & 7], v[i]);
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i] + w[i] + u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    double* d; (void)hipMalloc(&d, sizeof(double) * 256 * 8192);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const char* names[] = {"fma v,s,s", "fma v,v,s", "fma v,v,v", "fmac v+=v*v", "fmac mixed"};
    for (int mode = 0; mode < 5; ++mode)
        for (int bpc : {1, 2, 4}) {
            const int iters = 20000, grid = 256 * bpc;
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipEventRecord(e0);
                switch (mode) {
                    case 0: k<0><<<grid, 256>>>(d, iters, 1.0000001, 1e-9); break;
                    case 1: k<1><<<grid, 256>>>(d, iters, 1.0000001, 1e-9); break;
                    case 2: k<2><<<grid, 256>>>(d, iters, 1.0000001, 1e-9); break;
                    case 3: k<3><<<grid, 256>>>(d, iters, 1.0000001, 1e-9); break;
                    case 4: k<4><<<grid, 256>>>(d, iters, 1.0000001, 1e-9); break;
                }
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&ms, e0, e1);
            }
            const double inst = (double)grid * 4 * iters * 8, per_simd = inst / 1024.0;
            printf("%-12s waves/SIMD=%d  %.3f ms  ns/inst/SIMD=%.3f  TFLOP/s=%.1f\n", names[mode], bpc, ms,
                   ms * 1e6 / per_simd, inst * 128 / ms / 1e9);
        }
    return 0;
}
