// What clock does the chip hold under fp64 VALU load?  W wavefronts per SIMD on every SIMD run 16 independent fp64 FMA chains
// with `active` of 64 lanes enabled; per launch: wall time (events), shader cycles (s_memtime) and 100 MHz ticks
// (s_memrealtime) of wavefront 0 -> clock = cycles / real time, FMA wave-instructions per second per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -o clock_power clock_power.hip && ./clock_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int W>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(W, W))) fma_kernel(double* out, unsigned long long* t,
                                                                                            int iters, int active, int sleep) {
    extern __shared__ double pad[];          // (160 KB / (4 W) per block: exactly W wavefronts on every SIMD)
    const int lane = threadIdx.x;
    if (iters < 0) pad[lane] = 1.0;
    double v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = 1.0 + 1e-3 * (lane + k);
    const double a = 1.0000001, b = 1e-9;
    const unsigned long long c0 = clock64(), r0 = wall_clock64();
    if (lane < active) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
                for (int k = 0; k < 16; ++k) v[k] = __builtin_fma(v[k], a, b);
            }
            if (sleep) __builtin_amdgcn_s_sleep(8);
        }
    }
    const unsigned long long c1 = clock64(), r1 = wall_clock64();
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += v[k];
    out[(size_t)blockIdx.x * 64 + lane] = s;
    if (lane == 0) { t[2 * blockIdx.x] = c1 - c0; t[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int W>
int run(int cus, int iters, int active, int sleep) {
    const int blocks = cus * 4 * W;
    double* out; unsigned long long* t;
    CK(hipMalloc(&out, sizeof(double) * blocks * 64));
    CK(hipMalloc(&t, sizeof(unsigned long long) * 2 * blocks));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    const int lds = 160 * 1024 / (4 * W);
    CK(hipFuncSetAttribute((const void*)fma_kernel<W>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    for (int rep = 0; rep < 3; ++rep) {          // the third launch is reported (clocks settled)
        CK(hipEventRecord(e0));
        fma_kernel<W><<<blocks, 64, lds>>>(out, t, iters, active, sleep);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    std::vector<unsigned long long> h(2 * blocks);
    CK(hipMemcpy(h.data(), t, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost));
    double cyc = 0, real = 0;
    for (int b = 0; b < blocks; ++b) { cyc += h[2 * b]; real += h[2 * b + 1]; }
    cyc /= blocks; real /= blocks;
    const double insts = (double)iters * 64;                 // FMA wave-instructions per wavefront
    const double clock_ghz = cyc / (real * 10.0);            // 100 MHz ticks -> ns
    printf("waves/SIMD %d active lanes %2d sleep %d: launch %.3f ms, %.0f cycles and %.1f us per wavefront -> clock %.3f GHz, "
           "%.2f cycles per FMA per wavefront, %.3f G FMA wave-instructions/s per SIMD, %.1f TFLOP/s (active lanes)\n",
           W, active, sleep, ms, cyc, real / 100.0, clock_ghz, cyc / insts, insts * W / (ms * 1e6),
           insts * W / (ms * 1e6) * 1e9 * (cus * 4.0) * 2.0 * active / 1e12);
    CK(hipFree(out)); CK(hipFree(t));
    return 0;
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("%s, %d CUs, nominal clock %d MHz\n", p.gcnArchName, cus, p.clockRate / 1000);
    const int iters = 1000000;
    for (int active : {64, 16, 4})
        for (int sleep : {0, 1}) {
            if (run<1>(cus, iters, active, sleep)) return 1;
            if (run<2>(cus, iters / 2, active, sleep)) return 1;
            if (run<4>(cus, iters / 4, active, sleep)) return 1;
        }
    return 0;
}
