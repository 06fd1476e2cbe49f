// Scalar-load latency seen by ONE wavefront per SIMD: s_load_dwordxN + s_waitcnt lgkmcnt(0) back to back, a 2 KB table
// that every wavefront reads (the arma series), by alignment of the access.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define K(NAME, BODY)                                                                                                   \
    __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) NAME(const double* tab,              \
                                                                                          unsigned long long* cyc, int iters) { \
        extern __shared__ double pad[];                                                                                  \
        if (iters < 0) pad[threadIdx.x] = 1.0;                                                                           \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                      \
        for (int it = 0; it < iters; ++it)                                                                               \
            asm volatile(BODY ::"s"(tab)                                                                                 \
                         : "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", \
                           "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", \
                           "s64", "s65", "s66", "s67");                                                                  \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                      \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                                 \
    }
K(x16_al, "s_load_dwordx16 s[36:51], %0, 0x40\n\ts_waitcnt lgkmcnt(0)\n\ts_load_dwordx16 s[36:51], %0, 0x80\n\ts_waitcnt lgkmcnt(0)")
K(x16_un, "s_load_dwordx16 s[36:51], %0, 0x48\n\ts_waitcnt lgkmcnt(0)\n\ts_load_dwordx16 s[36:51], %0, 0x88\n\ts_waitcnt lgkmcnt(0)")
K(x8_al, "s_load_dwordx8 s[36:43], %0, 0x40\n\ts_waitcnt lgkmcnt(0)\n\ts_load_dwordx8 s[36:43], %0, 0x80\n\ts_waitcnt lgkmcnt(0)")
K(x8_un, "s_load_dwordx8 s[36:43], %0, 0x38\n\ts_waitcnt lgkmcnt(0)\n\ts_load_dwordx8 s[36:43], %0, 0x78\n\ts_waitcnt lgkmcnt(0)")
K(x4_al, "s_load_dwordx4 s[36:39], %0, 0x40\n\ts_waitcnt lgkmcnt(0)\n\ts_load_dwordx4 s[36:39], %0, 0x80\n\ts_waitcnt lgkmcnt(0)")
K(x2_al, "s_load_dwordx2 s[36:37], %0, 0x40\n\ts_waitcnt lgkmcnt(0)\n\ts_load_dwordx2 s[36:37], %0, 0x80\n\ts_waitcnt lgkmcnt(0)")
K(x16x2_al, "s_load_dwordx16 s[36:51], %0, 0x40\n\ts_load_dwordx16 s[52:67], %0, 0x80\n\ts_waitcnt lgkmcnt(0)\n\ts_load_dwordx16 s[36:51], %0, 0xc0\n\ts_load_dwordx16 s[52:67], %0, 0x100\n\ts_waitcnt lgkmcnt(0)")
K(x16x2_un, "s_load_dwordx16 s[36:51], %0, 0x48\n\ts_load_dwordx16 s[52:67], %0, 0x88\n\ts_waitcnt lgkmcnt(0)\n\ts_load_dwordx16 s[36:51], %0, 0xc8\n\ts_load_dwordx16 s[52:67], %0, 0x108\n\ts_waitcnt lgkmcnt(0)")
// a load with 70 independent fp64 instructions (280 cycles) between issue and wait: what is left of the latency
#define F10 "v_fma_f64 v[10:11], v[10:11], v[2:3], v[4:5]\n\tv_fma_f64 v[12:13], v[12:13], v[2:3], v[4:5]\n\tv_fma_f64 v[14:15], v[14:15], v[2:3], v[4:5]\n\tv_fma_f64 v[16:17], v[16:17], v[2:3], v[4:5]\n\tv_fma_f64 v[18:19], v[18:19], v[2:3], v[4:5]\n\tv_fma_f64 v[10:11], v[10:11], v[2:3], v[4:5]\n\tv_fma_f64 v[12:13], v[12:13], v[2:3], v[4:5]\n\tv_fma_f64 v[14:15], v[14:15], v[2:3], v[4:5]\n\tv_fma_f64 v[16:17], v[16:17], v[2:3], v[4:5]\n\tv_fma_f64 v[18:19], v[18:19], v[2:3], v[4:5]\n\t"
#define F70 F10 F10 F10 F10 F10 F10 F10
int main() {
    std::vector<double> t(4096, 0.5);
    double* d; unsigned long long* dc;
    (void)hipMalloc(&d, 4096 * 8); (void)hipMalloc(&dc, 1024 * 8);
    (void)hipMemcpy(d, t.data(), 4096 * 8, hipMemcpyHostToDevice);
    const int iters = 20000; const size_t lds = 36 * 1024;
    std::vector<unsigned long long> c(1024);
#define RUN(NAME, NLOAD)                                                                                                  \
    (void)hipFuncSetAttribute((const void*)NAME, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                   \
    for (int rep = 0; rep < 3; ++rep) NAME<<<1024, 64, lds>>>(d, dc, iters);                                              \
    (void)hipDeviceSynchronize(); (void)hipMemcpy(c.data(), dc, 1024 * 8, hipMemcpyDeviceToHost); std::sort(c.begin(), c.end()); \
    printf("%-10s %.1f cycles per load-and-wait round (median wave; max %.1f)\n", #NAME, (double)c[512] / iters / 2, (double)c[1023] / iters / 2);
    RUN(x16_al, 1) RUN(x16_un, 1) RUN(x8_al, 1) RUN(x8_un, 1) RUN(x4_al, 1) RUN(x2_al, 1) RUN(x16x2_al, 2) RUN(x16x2_un, 2)
    return 0;
}
