// Cycles per ArmaLaneModel evaluation (T = 200) with ONE wavefront per SIMD, as in nuts3_kernel:
// V0 = one lane per particle (64 evaluations per call), V1 = recur_wide<16> (4 particles per call), V2 = recur_wide<4> (16),
// V3 = recur_wide<64> (1), V4 = recur_wide<32> (2).
//   tools/ubench/arma_eval [iters]
#include "../../smcnuts_amd/csrc/smcn_nuts3.hpp"
#include <cstdio>
#include <vector>
#include <algorithm>
using namespace smcn;
using d2 = double __attribute__((ext_vector_type(2)));

template <int V>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1)))
k(const double* mdata, const double* x0, double* out, unsigned long long* cyc, int iters) {
    extern __shared__ d2 lds_[];
    ArmaLaneModel m;
    m.init(mdata);
    const int lane = threadIdx.x;
    double* const Yl = reinterpret_cast<double*>(lds_ + 36 * 64);
    d2* const XCH = reinterpret_cast<d2*>(Yl + ArmaLaneModel::YMAX + ArmaLaneModel::YPAD);
    for (int i = lane; i < m.T + ArmaLaneModel::YPAD; i += 64) Yl[i] = i < m.T ? mdata[1 + i] : 0.0;
    wave_exchange_fence();
    const int p = blockIdx.x * 64 + lane;
    double x[4];
    for (int c = 0; c < 4; ++c) x[c] = x0[p * 4 + c];
    constexpr int A = V == 1 ? 16 : V == 3 ? 64 : V == 4 ? 32 : 4;
    const bool act = V == 0 ? true : (lane % A) == ((5 * (lane / A) + 3) % A);
    double acc = 0.0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        double lpri = 0, llik = 0, gp[4] = {0, 0, 0, 0}, gl[4] = {0, 0, 0, 0}, ss = 0, gm = 0, gb = 0, gt = 0;
        if constexpr (V == 0) m.recur(x, ss, gm, gb, gt);
        else m.template recur_wide<A>(x, act, __ballot(act), Yl, XCH, lane, ss, gm, gb, gt);
        if (act) m.finish(x, ss, gm, gb, gt, lpri, llik, gp, gl);
        acc += lpri + llik + gp[0] + gp[1] + gp[2] + gp[3] + gl[0] + gl[1] + gl[2] + gl[3];
        x[0] += 1e-13 * gl[0]; x[1] += 1e-13 * gl[1]; x[2] += 1e-13 * gl[2]; x[3] += 1e-16 * gl[3];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[p] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 400;
    const int T = 200, N = 65536;
    std::vector<double> md(1 + T + 64, 0.0), x(N * 4);
    md[0] = T;
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) / 16777216.0; };
    for (int t = 0; t < T; ++t) md[1 + t] = 0.3 * (rnd() - 0.5);
    for (int i = 0; i < N; ++i) { x[4 * i] = 0.01 * rnd(); x[4 * i + 1] = 0.9 + 0.05 * rnd(); x[4 * i + 2] = -0.1 * rnd(); x[4 * i + 3] = -1.8 + 0.1 * rnd(); }
    double *dmd, *dx, *dout; unsigned long long* dc;
    (void)hipMalloc(&dmd, md.size() * 8); (void)hipMalloc(&dx, x.size() * 8); (void)hipMalloc(&dout, N * 8); (void)hipMalloc(&dc, 1024 * 8);
    (void)hipMemcpy(dmd, md.data(), md.size() * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(dx, x.data(), x.size() * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const size_t lds = n3_lds_bytes<ArmaLaneModel>(3, 3);   // the product's: four blocks per CU = one wavefront per SIMD
    auto launch = [&](int v) {
        if (v == 0) k<0><<<1024, 64, lds>>>(dmd, dx, dout, dc, iters);
        if (v == 1) k<1><<<1024, 64, lds>>>(dmd, dx, dout, dc, iters);
        if (v == 2) k<2><<<1024, 64, lds>>>(dmd, dx, dout, dc, iters);
        if (v == 3) k<3><<<1024, 64, lds>>>(dmd, dx, dout, dc, iters);
        if (v == 4) k<4><<<1024, 64, lds>>>(dmd, dx, dout, dc, iters);
    };
    (void)hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)k<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int v = 0; v < 5; ++v) {
        for (int rep = 0; rep < 40; ++rep) launch(v);   // clocks ramp over the first ~100 ms
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        launch(v);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> c(1024);
        (void)hipMemcpy(c.data(), dc, 1024 * 8, hipMemcpyDeviceToHost);
        std::sort(c.begin(), c.end());
        printf("V%d (%s): %.3f ms, %.0f ns per call; s_memtime ticks per call: median %.0f, max %.0f; ticks/ns %.3f\n", v,
               v == 0 ? "one lane per particle, 64 per call" : v == 1 ? "16 lanes per particle, 4 per call" : v == 2 ? "4 lanes per particle, 16 per call" : v == 3 ? "64 lanes per particle, 1 per call" : "32 lanes per particle, 2 per call",
               ms, ms * 1e6 / iters, (double)c[512] / iters, (double)c[1023] / iters, (double)c[512] / (ms * 1e6));
    }
    return 0;
}
