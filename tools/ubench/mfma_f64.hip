// v_mfma_f64_16x16x4_f64 on gfx950: operand layout check, cycles per instruction (independent / dependent
// accumulators), and whether fp64 VALU FMAs issued between MFMAs run in their shadow (one and two wavefronts per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void layout_kernel(const double* A, const double* B, double* C) {   // A[16][4], B[4][16] row-major
    const int l = threadIdx.x;
    const double a = A[(l & 15) * 4 + (l >> 4)];
    const double b = B[(l >> 4) * 16 + (l & 15)];
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}

template <int NACC, int NFMA, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) rate_kernel(unsigned long long* cyc, int iters, double seed, double* sink) {
    const int l = threadIdx.x & 63;
    double a = seed + l, b = seed * 0.5 + l;
    d4 acc[NACC > 0 ? NACC : 1];
    for (int k = 0; k < (NACC > 0 ? NACC : 1); ++k) acc[k] = d4{0, 0, 0, 0};
    double f[8];
    for (int k = 0; k < 8; ++k) f[k] = seed + k;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < (NACC > 0 ? NACC : 1); ++k) {
            if constexpr (NACC > 0) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[k], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < NFMA; ++m) f[m & 7] = __builtin_fma(f[m & 7], 1.0000001, 1e-9);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int k = 0; k < NACC; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    for (int k = 0; k < 8; ++k) s += f[k];
    if (s == 12345.678) sink[0] = s;
    if (l == 0) cyc[blockIdx.x * WAVES + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NACC, int NFMA, int WAVES>
void run(const char* what, int blocks_per_cu) {
    const int nb = 256 * blocks_per_cu, iters = 2000;
    unsigned long long* d; double* sink;
    hipMalloc(&d, sizeof(unsigned long long) * nb * WAVES); hipMalloc(&sink, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    rate_kernel<NACC, NFMA, WAVES><<<nb, 64 * WAVES>>>(d, 10, 1.0, sink);
    hipEventRecord(e0);
    rate_kernel<NACC, NFMA, WAVES><<<nb, 64 * WAVES>>>(d, iters, 1.0, sink);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(nb * WAVES);
    hipMemcpy(h.data(), d, sizeof(unsigned long long) * nb * WAVES, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double cyc = (double)h[h.size() / 2];
    const int per = (NACC > 0 ? NACC : 1);
    printf("%-58s waves/block %d blocks/CU %d: %8.1f ticks per group [%d mfma + %d fma] ; kernel %.3f ms -> %.2f ns per group per wave\n",
           what, WAVES, blocks_per_cu, cyc / iters / per, NACC > 0 ? 1 : 0, NFMA, ms, ms * 1e6 / iters / per);
    hipFree(d); hipFree(sink);
}

int main() {
    // layout
    std::vector<double> A(64), B(64), C(256), R(256, 0.0);
    for (int i = 0; i < 64; ++i) { A[i] = 1 + 0.37 * i; B[i] = 2 - 0.11 * i * i; }
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) { double s = 0; for (int k = 0; k < 4; ++k) s = fma(A[m * 4 + k], B[k * 16 + n], s); R[m * 16 + n] = s; }
    double *dA, *dB, *dC; hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dC, 2048);
    hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
    layout_kernel<<<1, 64>>>(dA, dB, dC);
    hipMemcpy(C.data(), dC, 2048, hipMemcpyDeviceToHost);
    double e = 0; int exact = 0; for (int i = 0; i < 256; ++i) { e = std::max(e, fabs(C[i] - R[i])); exact += C[i] == R[i]; }
    printf("layout: A lane l = A[l&15][l>>4], B lane l = B[l>>4][l&15], C reg r = C[(l>>4)+4r][l&15]: max |err| %.3g, %d/256 bit-equal to a k-ordered fma chain\n", e, exact);
    run<8, 0, 1>("8 independent accumulators, no VALU", 4);
    run<4, 0, 1>("4 independent accumulators, no VALU", 4);
    run<2, 0, 1>("2 independent accumulators, no VALU", 4);
    run<1, 0, 1>("1 accumulator (dependent chain), no VALU", 4);
    run<0, 8, 1>("8 fp64 FMAs only", 4);
    run<0, 16, 1>("16 fp64 FMAs only", 4);
    run<8, 4, 1>("8 acc, 4 fp64 FMAs behind each MFMA", 4);
    run<8, 8, 1>("8 acc, 8 fp64 FMAs behind each MFMA", 4);
    run<8, 16, 1>("8 acc, 16 fp64 FMAs behind each MFMA", 4);
    run<8, 32, 1>("8 acc, 32 fp64 FMAs behind each MFMA", 4);
    run<1, 16, 1>("1 acc (dependent), 16 fp64 FMAs behind each MFMA", 4);
    run<8, 0, 2>("two waves per SIMD: 8 acc, no VALU", 4);
    run<8, 16, 2>("two waves per SIMD: 8 acc, 16 FMAs behind each", 4);
    run<0, 16, 2>("two waves per SIMD: 16 FMAs only", 4);
    return 0;
}
