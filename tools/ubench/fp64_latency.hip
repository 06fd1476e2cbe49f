// Dependent-chain latency of v_fma_f64 on gfx950: C independent chains per wave, W waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int C>
__global__ void __launch_bounds__(256) k(double* out, int iters, double a, double b) {
    double v[C];
#pragma unroll
    for (int i = 0; i < C; ++i) v[i] = a + i + threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < C; ++i) v[i] = fma(v[i], a, b);
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < C; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int C>
void run(double* d) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int bpc : {1, 2}) {
        const int iters = 20000, grid = 256 * bpc;
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            k<C><<<grid, 256>>>(d, iters, 1.0000001, 1e-9);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        const double steps = (double)iters * 8;           // dependent steps per chain
        printf("chains/wave=%d waves/SIMD=%d: %.2f ns per dependent step, %.2f ns per instruction per SIMD\n", C, bpc,
               ms * 1e6 / steps, ms * 1e6 / (steps * C * bpc));
    }
}
int main() {
    double* d; (void)hipMalloc(&d, sizeof(double) * 256 * 4096);
    run<1>(d); run<2>(d); run<4>(d); run<8>(d);
    return 0;
}
