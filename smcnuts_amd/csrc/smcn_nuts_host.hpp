// NUTS for targets whose density lives on the HOST (any object with the reference's StanModel
// interface, smcnuts/model/bridgestan.py:28-120: BridgeStan models other than the device-native
// functors).  SURVEY.md 8 f4.
//
// The tree building of NUTSProposal.rvs (smcnuts/proposal/nuts.py:34-175) stays on the GPU; only
// the value/gradient of the target is asked from the caller, in lock step for all particles: one
// launch of nuts_host_advance_kernel consumes the evaluation at the pending positions, advances
// every live particle by ONE step of the same per-leaf state machine as nuts_kernel
// (smcn_nuts.hpp) and leaves the next positions to evaluate.  One thread owns one particle; D is
// a run-time value, so all vectors live in global memory, [vector][coordinate][particle]
// (coalesced over particles).  Inherently latency-bound (a host round trip per leapfrog): this is
// the generality path, not the fast one.
#pragma once
#include "smcn_device.hpp"

namespace smcn {

// vectors ([HV_COUNT][D][N])
enum : int { HV_X = 0, HV_R, HV_G, HV_EMX, HV_EMR, HV_EMG, HV_EPX, HV_EPR, HV_EPG, HV_SELX, HV_SELR,
             HV_FIRSTX, HV_FIRSTR = HV_FIRSTX + 10, HV_CANDX = HV_FIRSTR + 10, HV_CANDR = HV_CANDX + 10,
             HV_COUNT = HV_CANDR + 10 };
// doubles per particle ([HS_COUNT][N])
enum : int { HS_LOGU = 0, HS_SELP, HS_SELL, HS_LP0, HS_LL0, HS_CANDP, HS_CANDL = HS_CANDP + 10,
             HS_CANDN = HS_CANDL + 10, HS_COUNT = HS_CANDN + 10 };
// ints per particle ([HI_COUNT][N])
enum : int { HI_PHASE = 0, HI_J, HI_I, HI_DIR, HI_N, HI_NLEAP, HI_Q, HI_FLAGS, HI_COUNT };
enum : int { HP_INIT = 1, HP_LEAF = 2, HP_DONE = 3 };

struct NutsHostArgs {
    int64_t N, particle_base;
    int D;
    double* vec;        // [HV_COUNT][D][N]
    double* sc;         // [HS_COUNT][N]
    int32_t* st;        // [HI_COUNT][N]
    const double* lpri; // [N]      evaluation at the pending positions (vec[HV_X])
    const double* llik; // [N]
    const double* gpri; // [D][N]
    const double* glik; // [D][N]
    double eps, phi, delta_max;
    int max_depth;
    uint64_t seed;
    uint32_t iter;
    const double* tape;       // recorded draws (tests) or null: Philox stream 0
    const int64_t* tape_off;  // [N+1]
    unsigned int* n_active;   // out: particles that still need an evaluation
};

__global__ void __launch_bounds__(256) nuts_host_advance_kernel(NutsHostArgs a) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= a.N) return;
    const int64_t N = a.N;
    const int D = a.D;
    int32_t* const st = a.st;
    int phase = st[HI_PHASE * N + p];
    if (phase == HP_DONE) return;
    auto V = [&](int v, int k) -> double& { return a.vec[((int64_t)v * D + k) * N + p]; };
    auto S = [&](int s) -> double& { return a.sc[(int64_t)s * N + p]; };
    int j = st[HI_J * N + p], i = st[HI_I * N + p], dir = st[HI_DIR * N + p], n = st[HI_N * N + p];
    int nleap = st[HI_NLEAP * N + p], flags = st[HI_FLAGS * N + p];
    uint32_t q = (uint32_t)st[HI_Q * N + p];
    const int64_t toff = a.tape ? a.tape_off[p] : 0, tlen = a.tape ? a.tape_off[p + 1] - toff : 0;
    auto draw = [&]() -> double {
        double v;
        if (a.tape) {
            if ((int64_t)q < tlen) v = a.tape[toff + q];
            else { v = 0.5; flags |= 1; }
        } else {
            v = philox_uniform(a.seed, a.iter, (uint32_t)(a.particle_base + p), kStreamNuts, q);
        }
        ++q;
        return v;
    };
    auto copyv = [&](int src, int dst) { for (int k = 0; k < D; ++k) V(dst, k) = V(src, k); };
    // (x_cur - x_other) . r_other and . r_cur; U-turn by direction (nuts.py:152-160)
    auto uturn = [&](int ox, int orr) -> bool {
        double sa = 0.0, sb = 0.0;
        for (int k = 0; k < D; ++k) {
            const double xc = V(HV_X, k), xo = V(ox, k), rc = V(HV_R, k), ro = V(orr, k);
            const double dx = dir > 0 ? (xc - xo) : (xo - xc);   // xplus - xminus
            sa = fma(dx, dir > 0 ? ro : rc, sa);                 // . r_minus
            sb = fma(dx, dir > 0 ? rc : ro, sb);                 // . r_plus
        }
        return (sa < 0.0) || (sb < 0.0);
    };

    // ---- the evaluation that was pending (bridgestan.py:45-49,79-80: non-finite -> -inf) --------
    const double lpri = a.lpri[p], llik = a.llik[p];
    double lp = lpri + a.phi * llik;
    const bool bad = !finite_d(lp);
    lp = bad ? -kInf : lp;
    for (int k = 0; k < D; ++k)
        V(HV_G, k) = bad ? -kInf : fma(a.phi, a.glik[(int64_t)k * N + p], a.gpri[(int64_t)k * N + p]);
    auto kinetic = [&]() { double s = 0.0; for (int k = 0; k < D; ++k) s = fma(V(HV_R, k), V(HV_R, k), s); return s; };

    if (phase == HP_INIT) {
        // nuts.py:66-87
        const double H0 = lp - 0.5 * kinetic();
        double ex = draw();
        if (!a.tape) ex = -log1p(-ex);
        S(HS_LOGU) = H0 - ex;
        copyv(HV_X, HV_EMX); copyv(HV_R, HV_EMR); copyv(HV_G, HV_EMG);
        copyv(HV_X, HV_EPX); copyv(HV_R, HV_EPR); copyv(HV_G, HV_EPG);
        copyv(HV_X, HV_SELX); copyv(HV_R, HV_SELR);
        S(HS_SELP) = lpri; S(HS_SELL) = llik; S(HS_LP0) = lpri; S(HS_LL0) = llik;
        j = 0; n = 1; i = 0;
        dir = (draw() < 0.5) ? 1 : -1;   // nuts.py:91
        phase = HP_LEAF;
    } else {
        // ---- leapfrog, second half (nuts.py:173) and leaf tests (:123-125)
        const double h = dir * a.eps / 2;
        for (int k = 0; k < D; ++k) V(HV_R, k) = V(HV_R, k) + h * V(HV_G, k);
        ++nleap;
        const double logu = S(HS_LOGU);
        const double joint = lp - 0.5 * kinetic();
        int nsub = (logu < joint) ? 1 : 0;
        bool ssub = (logu - a.delta_max) >= joint;
        int csrc = -1;                 // the sub-tree's candidate: -1 = this leaf, m = parked record m
        double clp = lpri, cll = llik;
        if (j > 0 && (i & 1) == 0) {
            const int s = (i == 0) ? j : (__ffs(i) - 1);
            copyv(HV_X, HV_FIRSTX + s - 1);
            copyv(HV_R, HV_FIRSTR + s - 1);
        }
        // ---- merge completed sub-trees (nuts.py:134-148)
        bool done = false;
        int m = 0;
        for (;;) {
            if (m == j) { done = true; break; }
            if (ssub) {   // every ancestor whose SECOND half stopped still consumes its merge uniform
                q += (uint32_t)__popc((unsigned)(i >> m) & ((1u << (j - m)) - 1u));
                done = true;
                break;
            }
            if (((i >> m) & 1) == 0) {   // first half of level m+1: park the candidate
                if (csrc < 0) { copyv(HV_X, HV_CANDX + m); copyv(HV_R, HV_CANDR + m); }
                else { copyv(HV_CANDX + csrc, HV_CANDX + m); copyv(HV_CANDR + csrc, HV_CANDR + m); }
                S(HS_CANDP + m) = clp; S(HS_CANDL + m) = cll; S(HS_CANDN + m) = (double)nsub;
                break;
            }
            const double u = draw();   // nuts.py:142, always
            const int n1 = (int)S(HS_CANDN + m);
            const int den = (n1 + nsub) > 1 ? (n1 + nsub) : 1;
            if (!(u < (double)nsub / (double)den)) {   // keep the first half's candidate
                csrc = m; clp = S(HS_CANDP + m); cll = S(HS_CANDL + m);
            }
            nsub += n1;   // :146
            const int i0 = (i >> (m + 1)) << (m + 1);
            const int s = (i0 == 0) ? j : (__ffs(i0) - 1);
            ssub = uturn(HV_FIRSTX + s - 1, HV_FIRSTR + s - 1);   // :148
            ++m;
        }
        if (!done) {
            ++i;
        } else {
            // ---- end of this doubling (nuts.py:93-110)
            if (!ssub) {   // :99 short-circuit: no draw after a stop
                const double u = draw();
                double ratio = (double)nsub / (double)n;
                ratio = ratio > 1.0 ? 1.0 : ratio;
                if (u < ratio) {
                    if (csrc < 0) { copyv(HV_X, HV_SELX); copyv(HV_R, HV_SELR); }
                    else { copyv(HV_CANDX + csrc, HV_SELX); copyv(HV_CANDR + csrc, HV_SELR); }
                    S(HS_SELP) = clp; S(HS_SELL) = cll;
                }
            }
            n += nsub;   // :103
            const int ex = (dir > 0) ? HV_EPX : HV_EMX;   // the edge that moved
            const int ox = (dir > 0) ? HV_EMX : HV_EPX;   // the opposite edge
            copyv(HV_X, ex); copyv(HV_R, ex + 1); copyv(HV_G, ex + 2);
            const bool stop = ssub || uturn(ox, ox + 1);   // :105
            ++j;
            if (stop || j > a.max_depth) {   // :89,109
                phase = HP_DONE;
            } else {
                dir = (draw() < 0.5) ? 1 : -1;   // :91
                const int so = (dir > 0) ? HV_EPX : HV_EMX;
                copyv(so, HV_X); copyv(so + 1, HV_R); copyv(so + 2, HV_G);
                i = 0;
            }
        }
    }
    if (phase == HP_LEAF) {
        // ---- leapfrog, first half (nuts.py:169-170): the next position to evaluate
        const double e = dir * a.eps, h = dir * a.eps / 2;
        for (int k = 0; k < D; ++k) {
            const double rk = V(HV_R, k) + h * V(HV_G, k);
            V(HV_R, k) = rk;
            V(HV_X, k) = V(HV_X, k) + e * rk;
        }
        atomicAdd(a.n_active, 1u);
    }
    st[HI_PHASE * N + p] = phase; st[HI_J * N + p] = j; st[HI_I * N + p] = i; st[HI_DIR * N + p] = dir;
    st[HI_N * N + p] = n; st[HI_NLEAP * N + p] = nleap; st[HI_Q * N + p] = (int32_t)q; st[HI_FLAGS * N + p] = flags;
}

// start of a proposal: pending position = x0, momentum = r0, phase INIT
__global__ void nuts_host_begin_kernel(const double* x, const double* r, double* vec, int32_t* st, int64_t N, int D) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N) return;
    for (int k = 0; k < D; ++k) {
        vec[((int64_t)HV_X * D + k) * N + p] = x[(int64_t)k * N + p];
        vec[((int64_t)HV_R * D + k) * N + p] = r[(int64_t)k * N + p];
    }
    for (int s = 0; s < HI_COUNT; ++s) st[(int64_t)s * N + p] = 0;
    st[(int64_t)HI_PHASE * N + p] = HP_INIT;
    st[(int64_t)HI_DIR * N + p] = 1;
    st[(int64_t)HI_N * N + p] = 1;
}

// end: selected sample, density parts, tree statistics into the context's proposal arrays
__global__ void nuts_host_finish_kernel(const double* vec, const double* sc, const int32_t* st, int64_t N, int D,
                                        double* x_new, double* r_new, double* lpri0, double* llik0, double* lpri1,
                                        double* llik1, int32_t* nleap, int32_t* depth, int32_t* ndraws, int32_t* flags) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N) return;
    for (int k = 0; k < D; ++k) {
        x_new[(int64_t)k * N + p] = vec[((int64_t)HV_SELX * D + k) * N + p];
        r_new[(int64_t)k * N + p] = vec[((int64_t)HV_SELR * D + k) * N + p];
    }
    lpri0[p] = sc[(int64_t)HS_LP0 * N + p]; llik0[p] = sc[(int64_t)HS_LL0 * N + p];
    lpri1[p] = sc[(int64_t)HS_SELP * N + p]; llik1[p] = sc[(int64_t)HS_SELL * N + p];
    nleap[p] = st[(int64_t)HI_NLEAP * N + p]; depth[p] = st[(int64_t)HI_J * N + p];
    ndraws[p] = st[(int64_t)HI_Q * N + p]; flags[p] = st[(int64_t)HI_FLAGS * N + p];
}

}  // namespace smcn
