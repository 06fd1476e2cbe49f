// Cycles per PrmwcdLaneModel evaluation (one lane per particle: value + gradient of 64 particles per wavefront) in a dependent
// chain at one wavefront per SIMD -- the design rows by scalar loads (LK = false) or by LDS broadcast reads (LK = true) --
// beside the 8-lanes-per-particle FAST functor (8 particles per wavefront).  tools/ubench/lane_eval [iters]
#include "../../smcnuts_amd/csrc/smcn_models.hpp"
#include "../../smcnuts_amd/csrc/smcn_models_variants.hpp"
#include <cstdio>
#include <vector>
#include <algorithm>
using namespace smcn;

template <class Model>
__global__ void __launch_bounds__(256, 1) k(const double* mdata, const double* x0, double* out, unsigned long long* cyc, int iters) {
    extern __shared__ double lds_[];
    constexpr int G = Model::G, DL = Model::DL;
    const int lane = threadIdx.x & 63, lg = lane & (G - 1);
    Model m;
    m.init(mdata, lg, lds_);
    const int p = (blockIdx.x * 256 + threadIdx.x) / G;
    double x[DL];
    for (int i = 0; i < DL; ++i) { const int c = lg + G * i; x[i] = c < m.dim() ? x0[p * 16 + c] : 0.0; }
    double acc = 0.0;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        double lpri, llik, gp[DL], gl[DL];
        m.eval(x, lpri, llik, gp, gl);
        acc += lpri + llik;
        for (int i = 0; i < DL; ++i) x[i] += 1e-9 * (gp[i] + gl[i]);     // the next evaluation depends on this one
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lg == 0) out[p] = acc;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <class Model>
void run(const char* what, int iters) {
    const int C = 11, N = 65536, nobs = 100;
    std::vector<double> md(4 + nobs + nobs * C), x(N * 16, 0.0);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) / 16777216.0; };
    md[0] = nobs; md[1] = C + 1; md[2] = C; md[3] = 0.5;
    for (int i = 0; i < nobs; ++i) md[4 + i] = (double)(int)(6 * rnd());
    for (int i = 0; i < nobs * C; ++i) md[4 + nobs + i] = exp(-3.0 * rnd());
    for (int i = 0; i < N; ++i) for (int c = 0; c < 13; ++c) x[i * 16 + c] = 0.4 * (rnd() - 0.5);
    // the padded row table behind the data (smcn_ctx_create)
    const int RS = (C + 2) & ~1;
    const size_t len = md.size();
    md.resize((len + 15) / 16 * 16, 0.0);
    for (int i = 0; i < nobs + 2; ++i)
        for (int j = 0; j < RS; ++j)
            md.push_back(i < nobs ? (j < C ? md[4 + nobs + (size_t)i * C + j] : (j == RS - 1 ? md[4 + i] : 0.0)) : 0.0);
    md.resize(md.size() + 32, 0.0);
    double *dmd, *dx, *dout; unsigned long long* dc;
    (void)hipMalloc(&dmd, md.size() * 8); (void)hipMalloc(&dx, x.size() * 8); (void)hipMalloc(&dout, N * 8); (void)hipMalloc(&dc, 4096 * 8);
    (void)hipMemcpy(dmd, md.data(), md.size() * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(dx, x.data(), x.size() * 8, hipMemcpyHostToDevice);
    const int blocks = 256;
    size_t lds = sizeof(double) * Model::SHARED;
    if (lds < 90 * 1024) lds = 90 * 1024;      // keep a second block off the CU
    (void)hipFuncSetAttribute((const void*)k<Model>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 10; ++rep) k<Model><<<blocks, 256, lds>>>(dmd, dx, dout, dc, iters);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    k<Model><<<blocks, 256, lds>>>(dmd, dx, dout, dc, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(blocks * 4);
    (void)hipMemcpy(c.data(), dc, blocks * 4 * 8, hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end());
    std::vector<double> o(8);
    (void)hipMemcpy(o.data(), dout, 64, hipMemcpyDeviceToHost);
    const int per_wave = 64 / Model::G;
    printf("%-46s %8.1f ns per evaluation of %2d particles (%6.0f s_memtime ticks, median wave) = %6.1f ns per particle; checksum %.12g\n", what,
           ms * 1e6 / iters, per_wave, (double)c[c.size() / 2] / iters, ms * 1e6 / iters / per_wave, o[0]);
    (void)hipFree(dmd); (void)hipFree(dx); (void)hipFree(dout); (void)hipFree(dc);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200;
    run<PrmwcdDistModel<8, 100, 11, 2, 4, true>>("8 lanes per particle, FAST loop", iters);
    run<PrmwcdLaneModel<100, 11, 1, false>>("1 lane per particle, rows by scalar loads", iters);
    run<PrmwcdLaneModel<100, 11, 1, true>>("1 lane per particle, rows by LDS broadcast", iters);
    return 0;
}
