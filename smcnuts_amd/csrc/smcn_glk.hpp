// SPDX-License-Identifier: MIT
// The D x D algebra of the Gaussian-approximation L-kernel ON THE DEVICE (smcnuts/lkernel/gaussian_lkernel.py:45-82):
// between the moment sums and the per-particle conditional log-density the reference runs, on 2D x 2D matrices,
//     pinv(c_xx),  B = c_rx pinv(c_xx),  cov = c_rr - c_rx pinv(c_xx) c_xr + 1e-6 I,  multivariate_normal(cov).logpdf
// (np.linalg.pinv: SVD with a 1e-15 cut-off; scipy's _PSD: eigh with a 1e6 eps cut-off, allow_singular = False).  Where
// both matrices are comfortably positive definite -- every population that is not degenerate -- those are an inverse and
// a log-determinant + Mahalanobis form, which one wavefront gets from two Cholesky factorisations in a few microseconds:
//     c_xx = L L^T,  pinv = L^-T L^-1;   cov = Lc Lc^T,  U = Lc^-T  (|U^T d|^2 = d^T cov^-1 d),  log det = 2 sum log Lc_ii.
// The kernel PROVES that it is in that regime (condition estimates from trace x Frobenius norm of the inverse, an upper
// bound of the 2-norm condition number) and otherwise reports a status; the host then runs the reference's own calls,
// cut-offs and exceptions included (lkernel/gaussian_lkernel.py).  Agreement with the NumPy path: the L values to
// ~cond x 1e-16 (tests/test_gpu_parity.py::test_gaussian_lkernel_algebra_on_the_device).
#pragma once
#include "smcn_device.hpp"

namespace smcn {

constexpr int kGlkMaxD = 32;
constexpr double kGlkMaxCond = 1e8;   // beyond this estimate the host path decides (scipy raises at 4.5e9, pinv cuts at 1e15)

// mean of X = [-r', x'] from the un-shifted sums: mu[e] = sums[e] / n  (np.mean)
__global__ void __launch_bounds__(64) glk_mean_kernel(const double* sums, int E, double n, double* mu) {
    for (int e = threadIdx.x; e < E; e += 64) mu[e] = sums[e] / n;
}

// moment sums of several shards, added in rank order (what every rank does with the all-gathered rows, so that all ranks
// hold the same bits): dst[q] = src[0][q] + src[1][q] + ...
__global__ void __launch_bounds__(256) glk_combine_kernel(const double* src, int world, int nq, double* dst) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    double v = src[q];
    for (int w = 1; w < world; ++w) v += src[(size_t)w * nq + q];
    dst[q] = v;
}

// one wavefront; LDS: 6 D x D matrices.  sums2 = [E singles (unused), upper triangle of the centred E x E products],
// mu = [mu_r (D), mu_x (D)].  out = [mu_x (D), mu_r (D), B (D x D), U (D x D), c0, status, cond_xx, cond_cov].
__global__ void __launch_bounds__(64) glk_algebra_kernel(const double* sums2, const double* mu, int D, double n, double* out) {
    extern __shared__ double gsh[];
    const int E = 2 * D, lane = threadIdx.x;
    double* Crr = gsh;               // c_rr, later cov, later Lc
    double* Crx = Crr + D * D;       // c_rx
    double* Cxx = Crx + D * D;       // c_xx, later L
    double* Inv = Cxx + D * D;       // L^-1, later Lc^-1
    double* Pin = Inv + D * D;       // pinv(c_xx)
    double* Bm = Pin + D * D;        // B
    auto sync = []() { wave_exchange_fence(); };
    const double inv_n1 = 1.0 / (n - 1.0);                          // np.cov: / (N - 1)
    auto tri = [&](int i, int j) {                                   // (i <= j) of the E x E upper triangle, row-major
        return E + i * E - (i * (i - 1)) / 2 + (j - i);
    };
    for (int t = lane; t < D * D; t += 64) {
        const int i = t / D, j = t - i * D;
        const int a = i < j ? i : j, b = i < j ? j : i;
        Crr[t] = sums2[tri(a, b)] * inv_n1;
        Cxx[t] = sums2[tri(D + a, D + b)] * inv_n1;
        Crx[t] = sums2[tri(i, D + j)] * inv_n1;                      // r rows, x columns (i < D + j always)
    }
    sync();
    double status = 0.0;
    // Cholesky in place (lower triangle of M), then the inverse of the factor into Inv; returns trace(M) before
    auto cholesky_inverse = [&](double* M) -> double {
        double tr = 0.0;
        for (int k = 0; k < D; ++k) tr += M[k * D + k];
        for (int k = 0; k < D; ++k) {
            const double p = M[k * D + k];
            if (!(p > 0.0) || !finite_d(p)) status = 1.0;            // not positive definite
            const double piv = sqrt(p > 0.0 ? p : 1.0);
            sync();
            if (lane == 0) M[k * D + k] = piv;
            for (int i = k + 1 + lane; i < D; i += 64) M[i * D + k] = M[i * D + k] / piv;
            sync();
            for (int i = k + 1 + lane; i < D; i += 64) {             // trailing update, one row per lane
                const double lik = M[i * D + k];
                for (int j = k + 1; j <= i; ++j) M[i * D + j] = fma(-lik, M[j * D + k], M[i * D + j]);
            }
            sync();
        }
        // Inv = M^-1 (lower): column c by forward substitution, one column per lane
        for (int c = lane; c < D; c += 64) {
            for (int i = 0; i < D; ++i) {
                double v = (i == c) ? 1.0 : 0.0;
                for (int k = c; k < i; ++k) v = fma(-M[i * D + k], Inv[k * D + c], v);
                Inv[i * D + c] = (i < c) ? 0.0 : v / M[i * D + i];
            }
        }
        sync();
        return tr;
    };
    auto frob_of_inverse = [&]() -> double {                         // || Inv^T Inv ||_F >= 1 / lambda_min
        double f = 0.0;
        for (int t = lane; t < D * D; t += 64) {
            const int i = t / D, j = t - i * D;
            double v = 0.0;
            for (int k = (i > j ? i : j); k < D; ++k) v = fma(Inv[k * D + i], Inv[k * D + j], v);
            f = fma(v, v, f);
        }
        return sqrt(wave_sum_bpermute(f));
    };
    // ---- pinv(c_xx) ----------------------------------------------------------------------------
    const double tr_xx = cholesky_inverse(Cxx);
    const double cond_xx = tr_xx * frob_of_inverse();
    for (int t = lane; t < D * D; t += 64) {                          // Pin = L^-T L^-1
        const int i = t / D, j = t - i * D;
        double v = 0.0;
        for (int k = (i > j ? i : j); k < D; ++k) v = fma(Inv[k * D + i], Inv[k * D + j], v);
        Pin[t] = v;
    }
    sync();
    for (int t = lane; t < D * D; t += 64) {                          // B = c_rx pinv
        const int i = t / D, j = t - i * D;
        double v = 0.0;
        for (int k = 0; k < D; ++k) v = fma(Crx[i * D + k], Pin[k * D + j], v);
        Bm[t] = v;
    }
    sync();
    for (int t = lane; t < D * D; t += 64) {                          // cov = c_rr - B c_xr + 1e-6 I   (:68 ridge)
        const int i = t / D, j = t - i * D;
        double v = 0.0;
        for (int k = 0; k < D; ++k) v = fma(Bm[i * D + k], Crx[j * D + k], v);   // c_xr = c_rx^T
        Crr[t] = Crr[t] - v + (i == j ? 1e-6 : 0.0);
    }
    sync();
    // ---- multivariate_normal(cov).logpdf -------------------------------------------------------
    const double tr_cov = cholesky_inverse(Crr);
    const double cond_cov = tr_cov * frob_of_inverse();
    double logdet = 0.0;
    for (int k = 0; k < D; ++k) logdet += 2.0 * log(Crr[k * D + k]);
    if (!(cond_xx < kGlkMaxCond) || !(cond_cov < kGlkMaxCond)) status = status != 0.0 ? status : 2.0;
    for (int e = lane; e < D; e += 64) {
        out[e] = mu[D + e];                                           // mu_x
        out[D + e] = mu[e];                                           // m0 = mu_r
    }
    for (int t = lane; t < D * D; t += 64) {
        const int i = t / D, j = t - i * D;
        out[2 * D + t] = Bm[t];
        out[2 * D + D * D + t] = Inv[j * D + i];                      // U = Lc^-T
    }
    if (lane == 0) {
        double* tail = out + 2 * D + 2 * D * D;
        tail[0] = -0.5 * (D * kLog2Pi + logdet);                      // c0
        tail[1] = status;
        tail[2] = cond_xx;
        tail[3] = cond_cov;
    }
}

}  // namespace smcn
