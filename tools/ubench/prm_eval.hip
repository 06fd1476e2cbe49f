// Cycles per PrmwcdDistModel evaluation (value + gradient, 8 lanes per particle: 8 particles per wavefront) in a dependent
// chain, alone on a SIMD and with a second wavefront -- the LATENCY of config 4's leaf evaluation (the
// launch lasts as long as its longest tree, i.e. 2 047 x the latency of one lock-step leaf) and its split into the
// per-observation part and the fixed part (gather, reductions, prior): nobs = 100 against nobs = 8.
//   tools/ubench/prm_eval [iters]
#include "../../smcnuts_amd/csrc/smcn_models.hpp"
#include <cstdio>
#include <vector>
#include <algorithm>
using namespace smcn;

template <class Model>
__global__ void __launch_bounds__(256, Model::MIN_WAVES) k(const double* mdata, const double* x0, double* out, unsigned long long* cyc, int iters) {
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
void run(const char* what, int nobs, int waves_per_simd, int iters) {
    const int C = 11, N = 65536;
    std::vector<double> md(4 + nobs + nobs * C), x(N * 16, 0.0);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) / 16777216.0; };
    md[0] = nobs; md[1] = C + 1; md[2] = C; md[3] = 0.5;
    for (int i = 0; i < nobs; ++i) md[4 + i] = (double)(int)(6 * rnd());
    for (int i = 0; i < nobs * C; ++i) md[4 + nobs + i] = exp(-3.0 * rnd());
    for (int i = 0; i < N; ++i) for (int c = 0; c < 13; ++c) x[i * 16 + c] = 0.4 * (rnd() - 0.5);
    double *dmd, *dx, *dout; unsigned long long* dc;
    (void)hipMalloc(&dmd, md.size() * 8); (void)hipMalloc(&dx, x.size() * 8); (void)hipMalloc(&dout, N * 8); (void)hipMalloc(&dc, 4096 * 8);
    (void)hipMemcpy(dmd, md.data(), md.size() * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(dx, x.data(), x.size() * 8, hipMemcpyHostToDevice);
    const int blocks = 256 * waves_per_simd;            // 4 wavefronts per block: 1 or 2 blocks per CU
    size_t lds = sizeof(double) * Model::SHARED;
    if (waves_per_simd == 1 && lds < 90 * 1024) lds = 90 * 1024;      // keep a second block off the CU
    (void)hipFuncSetAttribute((const void*)k<Model>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 20; ++rep) k<Model><<<blocks, 256, lds>>>(dmd, dx, dout, dc, iters);
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
    printf("%-34s nobs %3d, %d wavefront(s) per SIMD: %8.1f ns per evaluation (%.0f s_memtime ticks, median wave), checksum %.12g\n", what, nobs,
           waves_per_simd, ms * 1e6 / iters, (double)c[c.size() / 2] / iters, o[0] + o[3]);
    (void)hipFree(dmd); (void)hipFree(dx); (void)hipFree(dout); (void)hipFree(dc);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 300;
    auto all = [&](auto tag, const char* name) {
        using M = decltype(tag);
        run<M>(name, 100, 1, iters); run<M>(name, 100, 2, iters); run<M>(name, 8, 1, iters); run<M>(name, 8, 2, iters);
    };
    all(PrmwcdDistModel<8, 100, 11, 2, 4>{}, "product <8,100,11,RED=2>");
    all(PrmwcdDistModel<8, 100, 11, 2, 4, true>{}, "RED=2, FAST observation loop");
    all(PrmwcdDistModel<8, 100, 11, 0, 4>{}, "RED=0 (LDS scratch)");
#ifdef PRM_EVAL_VARIANTS
    PRM_EVAL_VARIANTS
#endif
    return 0;
}
