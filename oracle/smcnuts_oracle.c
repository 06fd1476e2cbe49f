/*
 * oracle/smcnuts_oracle.c -- TEST INFRASTRUCTURE ONLY.  NOT PRODUCT CODE.
 *
 * A plain-C, single-thread CPU restatement of the SMC-NUTS hot path of the
 * reference (UoL-SignalProcessingGroup/SMC-NUTS @ 2024_10_08).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the
 * product (smcnuts_amd/) never does.
 *
 * Parity status
 *   - NUTS transition (tree build, leapfrog, U-turn, draw order): PINNED.  The
 *     recursion below follows smcnuts/proposal/nuts.py line by line and is
 *     checked in tests/test_oracle_golden.py against golden vectors produced
 *     by the real reference run in the build container
 *     (tests/golden/make_golden.py) on recorded per-particle RNG tapes.
 *   - Target densities (arma, PRMwCD): PARITY UNPINNED.  In the reference
 *     they live in BridgeStan + Stan Math (pip dependency `bridgestan`,
 *     version unpinned, README.md:19-23; call sites
 *     smcnuts/model/bridgestan.py:18,46,78,109), which is absent from
 *     /root/reference and from this image.  The functions below restate the
 *     published .stan programs (stan_models/arma/arma.stan:14-30,
 *     stan_models/PRMwCD/PRMwCD.stan:17-38) with BridgeStan's defaults
 *     (propto=True but every term is an explicit `target +=` of an _lpdf, so
 *     constants are kept; jacobian=True).  They are validated by an
 *     independent NumPy restatement, finite differences, and the posterior
 *     means in stan_models/<model>/<model>.params.
 *
 * All arithmetic is IEEE fp64, compiled with -ffp-contract=off so that the
 * operation order written here is the one executed.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_MAX_D 1024   /* largest dimension the fixed gradient scratch holds */
#define MODEL_GAUSS 0
#define MODEL_ARMA 1
#define MODEL_PRMWCD 2

#define LOG_2PI 1.8378770664093454835606594728112
#define LOG_PI 1.1447298858494001741434273513531

/* ------------------------------------------------------------------ */
/* Philox4x32-10 (Salmon et al., SC'11).  Production-mode RNG of the   */
/* build; the reference uses one shared sequential NumPy stream        */
/*  which a parallel sampler cannot consume.             */
/* ------------------------------------------------------------------ */
static void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int i = 0; i < 10; ++i) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void oracle_philox_raw(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    philox4x32_10(ctr, key, out);
}

/* 53-bit uniform in [0,1) from two words, the NumPy recipe
 * (a>>5)*2^26 + (b>>6)) / 2^53. */
static double u53(uint32_t a, uint32_t b) {
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}

/* Draw q of (seed, iteration, particle, stream): block q>>1 of the counter
 * (block, particle, iteration, stream), words (2h, 2h+1) with h = q&1. */
static double philox_uniform(uint64_t seed, uint32_t iter, uint32_t particle, uint32_t stream, uint32_t q) {
    uint32_t ctr[4] = {q >> 1, particle, iter, stream};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t o[4];
    philox4x32_10(ctr, key, o);
    return (q & 1u) ? u53(o[2], o[3]) : u53(o[0], o[1]);
}

void oracle_philox_uniforms(uint64_t seed, uint32_t iter, uint32_t particle, uint32_t stream,
                            uint32_t q0, int64_t count, double* out) {
    for (int64_t i = 0; i < count; ++i) out[i] = philox_uniform(seed, iter, particle, stream, q0 + (uint32_t)i);
}

void oracle_philox_particle_uniforms(uint64_t seed, uint32_t iter, int64_t particle_base, int64_t N,
                                     uint32_t stream, uint32_t q, double* out) {
    for (int64_t i = 0; i < N; ++i)
        out[i] = philox_uniform(seed, iter, (uint32_t)(particle_base + i), stream, q);
}

/* Box-Muller normals for one particle: pair m uses block m of `stream`:
 * z(2m) = R cos(2 pi u2), z(2m+1) = R sin(2 pi u2), R = sqrt(-2 log(1-u1)). */
void oracle_philox_normals(uint64_t seed, uint32_t iter, uint32_t particle, uint32_t stream, int D, double* out) {
    for (int m = 0; 2 * m < D; ++m) {
        double u1 = philox_uniform(seed, iter, particle, stream, 2u * (uint32_t)m);
        double u2 = philox_uniform(seed, iter, particle, stream, 2u * (uint32_t)m + 1u);
        double rad = sqrt(-2.0 * log1p(-u1));
        double ang = 6.283185307179586476925286766559 * u2;
        out[2 * m] = rad * cos(ang);
        if (2 * m + 1 < D) out[2 * m + 1] = rad * sin(ang);
    }
}

/* ------------------------------------------------------------------ */
/* Targets.  Each returns log prior (incl. Jacobian) and log           */
/* likelihood separately, so log pi_phi = lpri + phi * llik            */
/* (arma.stan:30 `target += phi * normal_lpdf(err | 0, sigma)`).       */
/* ------------------------------------------------------------------ */

/* Gaussian family (build-defined, SURVEY.md App. B "IsoGaussian"):
 * data = [D, s0, has_lik, m, s1]:  prior N(0, s0^2 I), optional likelihood
 * N(x | m 1, s1^2 I). */
static void gauss_eval(const double* data, int D, const double* x, double* lpri, double* llik,
                       double* gpri, double* glik) {
    double s0 = data[1], has = data[2], m = data[3], s1 = data[4];
    double ss = 0.0, sl = 0.0;
    for (int i = 0; i < D; ++i) {
        ss += x[i] * x[i];
        if (gpri) gpri[i] = -x[i] / (s0 * s0);
    }
    *lpri = -0.5 * ss / (s0 * s0) - D * log(s0) - 0.5 * D * LOG_2PI;
    if (has != 0.0) {
        for (int i = 0; i < D; ++i) {
            double d = x[i] - m;
            sl += d * d;
            if (glik) glik[i] = -d / (s1 * s1);
        }
        *llik = -0.5 * sl / (s1 * s1) - D * log(s1) - 0.5 * D * LOG_2PI;
    } else {
        *llik = 0.0;
        if (glik) for (int i = 0; i < D; ++i) glik[i] = 0.0;
    }
}

/* ARMA(1,1): stan_models/arma/arma.stan:8-30.  x = (mu, beta, theta, s),
 * sigma = exp(s); data = [T, y_1..y_T].  Gradient by forward sensitivities
 * (SURVEY.md App. B). */
static void arma_eval(const double* data, const double* x, double* lpri, double* llik,
                      double* gpri, double* glik) {
    int T = (int)data[0];
    const double* y = data + 1;
    double mu = x[0], beta = x[1], theta = x[2], s = x[3];
    double sigma = exp(s);
    /* arma.stan:20-23 priors + log-Jacobian of sigma = exp(s) */
    double z = sigma / 2.5;
    *lpri = (-0.5 * LOG_2PI - log(10.0) - 0.5 * (mu / 10.0) * (mu / 10.0))
          + (-0.5 * LOG_2PI - log(2.0) - 0.5 * (beta / 2.0) * (beta / 2.0))
          + (-0.5 * LOG_2PI - log(2.0) - 0.5 * (theta / 2.0) * (theta / 2.0))
          + (-LOG_PI - log(2.5) - log1p(z * z))
          + s;
    if (gpri) {
        gpri[0] = -mu / 100.0;
        gpri[1] = -beta / 4.0;
        gpri[2] = -theta / 4.0;
        gpri[3] = 1.0 - 2.0 * z * z / (1.0 + z * z);
    }
    /* arma.stan:25-30 */
    double err = y[0] - (mu + beta * mu);
    double dm = -(1.0 + beta), db = -mu, dt = 0.0;
    double ss = err * err, gm = err * dm, gb = err * db, gt = 0.0;
    for (int t = 1; t < T; ++t) {
        double nu = mu + beta * y[t - 1] + theta * err;
        double ndt = -err - theta * dt;
        double ndm = -1.0 - theta * dm;
        double ndb = -y[t - 1] - theta * db;
        err = y[t] - nu;
        dm = ndm; db = ndb; dt = ndt;
        ss += err * err;
        gm += err * dm; gb += err * db; gt += err * dt;
    }
    double w = 1.0 / (sigma * sigma);
    *llik = -0.5 * T * LOG_2PI - T * s - 0.5 * ss * w;
    if (glik) {
        glik[0] = -w * gm;
        glik[1] = -w * gb;
        glik[2] = -w * gt;
        glik[3] = -(double)T + ss * w;
    }
}

/* PRMwCD: stan_models/PRMwCD/PRMwCD.stan:11-38.  x = (Beta_1..Beta_M, g),
 * Gamma = exp(g); data = [Nobs, M, Clength, q, y_1..y_Nobs, Xkernel...]. */
static void prmwcd_eval(const double* data, const double* x, double* lpri, double* llik,
                        double* gpri, double* glik) {
    int Nobs = (int)data[0], M = (int)data[1], C = (int)data[2];
    double q = data[3];
    const double* y = data + 4;
    const double* X = data + 4 + Nobs;
    double g = x[M];
    double eg = exp(-g);
    /* inv_gamma_lpdf(Gamma | 2, 1.3) + log-Jacobian g   (PRMwCD.stan:21) */
    double lp = 2.0 * log(1.3) - lgamma(2.0) - 3.0 * g - 1.3 * eg + g;
    double dg = -3.0 + 1.3 * eg + 1.0;
    if (gpri) gpri[0] = 0.0;
    /* PRMwCD.stan:36-38 exponential-power prior on Beta_2..Beta_M */
    for (int j = 1; j < M; ++j) {
        double a = fabs(x[j]) * eg;
        double p = pow(a, q);
        lp += -g - p;
        dg += -1.0 + q * p;
        if (gpri) {
            double sgn = (x[j] > 0.0) - (x[j] < 0.0);
            gpri[j] = -q * sgn * pow(fabs(x[j]), q - 1.0) * pow(eg, q);
        }
    }
    if (gpri) gpri[M] = dg;
    *lpri = lp;
    /* PRMwCD.stan:24-33 Poisson likelihood */
    double ll = 0.0;
    if (glik) for (int j = 0; j <= M; ++j) glik[j] = 0.0;
    for (int i = 0; i < Nobs; ++i) {
        double eta = x[0];
        for (int j = 0; j < C; ++j) eta += x[j + 1] * X[i * C + j];
        double mu = exp(eta);
        double yi = y[i];
        double term;
        if (isinf(mu)) term = -INFINITY;                 /* Stan: poisson_lpmf(y|inf) = LOG_ZERO */
        else if (mu == 0.0 && yi != 0.0) term = -INFINITY; /* lambda == 0, n != 0 */
        else term = (yi == 0.0 ? 0.0 : yi * eta) - mu - lgamma(yi + 1.0);
        ll += term;
        if (glik) {
            double d = yi - mu;
            glik[0] += d;
            for (int j = 0; j < C; ++j) glik[j + 1] += d * X[i * C + j];
        }
    }
    *llik = ll;
}

static int model_cdim(int model, int D) { (void)model; return D; }

static void target_parts(int model, const double* data, int D, const double* x, double* lpri,
                         double* llik, double* gpri, double* glik) {
    switch (model) {
        case MODEL_GAUSS: gauss_eval(data, D, x, lpri, llik, gpri, glik); break;
        case MODEL_ARMA: arma_eval(data, x, lpri, llik, gpri, glik); break;
        default: prmwcd_eval(data, x, lpri, llik, gpri, glik); break;
    }
}

/* log pi_phi and gradient with the target adapter's failure convention
 * (smcnuts/model/bridgestan.py:45-49,77-80): anything that is not a finite
 * number becomes -inf, and its gradient a vector of -inf. */
static double target_logp_grad(int model, const double* data, int D, const double* x, double phi,
                               double* grad, double* lpri_out, double* llik_out) {
    double lpri, llik;
    double gp[ORACLE_MAX_D], gl[ORACLE_MAX_D];
    target_parts(model, data, D, x, &lpri, &llik, grad ? gp : NULL, grad ? gl : NULL);
    double lp = lpri + phi * llik;
    if (lpri_out) *lpri_out = lpri;
    if (llik_out) *llik_out = llik;
    if (!isfinite(lp)) {
        if (grad) for (int i = 0; i < D; ++i) grad[i] = -INFINITY;
        return -INFINITY;
    }
    if (grad) for (int i = 0; i < D; ++i) grad[i] = gp[i] + phi * gl[i];
    return lp;
}

/* Batched target evaluation: x row-major [M, D]. Any output may be NULL. */
int oracle_target_eval(int model, const double* data, int64_t M, int D, const double* x, double phi,
                       double* logp, double* grad, double* lpri, double* llik) {
    if (D > ORACLE_MAX_D) return -1;
    for (int64_t i = 0; i < M; ++i) {
        double a, b;
        double lp = target_logp_grad(model, data, D, x + i * D, phi, grad ? grad + i * D : NULL, &a, &b);
        if (logp) logp[i] = lp;
        if (lpri) lpri[i] = a;
        if (llik) llik[i] = b;
    }
    return 0;
}

/* constrain(): arma exp() on the last coordinate (sigma), PRMwCD exp() on the
 * last (Gamma), Gaussian identity. */
int oracle_constrain(int model, int64_t M, int D, const double* x, double* out) {
    int cd = model_cdim(model, D);
    for (int64_t i = 0; i < M; ++i) {
        for (int j = 0; j < D; ++j) out[i * cd + j] = x[i * D + j];
        if (model != MODEL_GAUSS) out[i * cd + D - 1] = exp(x[i * D + D - 1]);
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* RNG front: recorded tape (exact replay of the reference's draws for */
/* one particle) or Philox.                                            */
/* ------------------------------------------------------------------ */
typedef struct {
    int mode; /* 0 tape, 1 philox */
    const double* tape;
    int64_t len;
    uint32_t pos;
    uint64_t seed;
    uint32_t iter, particle;
    int overflow;
} rng_t;

static double rng_uniform(rng_t* g) {
    if (g->mode == 0) {
        if ((int64_t)g->pos >= g->len) { g->overflow = 1; g->pos++; return 0.5; }
        return g->tape[g->pos++];
    }
    return philox_uniform(g->seed, g->iter, g->particle, 0u, g->pos++);
}
/* nuts.py:69 rng.exponential(1): a tape stores the exponential itself. */
static double rng_exponential(rng_t* g) {
    if (g->mode == 0) return rng_uniform(g);
    return -log1p(-rng_uniform(g));
}

/* ------------------------------------------------------------------ */
/* NUTS, following smcnuts/proposal/nuts.py                            */
/* ------------------------------------------------------------------ */
typedef struct {
    int model, D;
    const double* data;
    double phi, eps, delta_max;
    double logu;
    rng_t* rng;
    int nleap;
    /* log prior / log lik of the most recent leaf */
    double leaf_lpri, leaf_llik;
    double* arena; /* one second-half tree (8 D doubles) per recursion depth */
} nuts_t;

static double dotv(const double* a, const double* b, int D) {
    double s = 0.0;
    for (int i = 0; i < D; ++i) s += a[i] * b[i];
    return s;
}

/* nuts.py:152-160 */
static int stop_criterion(const double* xm, const double* xp, const double* rm, const double* rp, int D) {
    double a = 0.0, b = 0.0;
    for (int i = 0; i < D; ++i) {
        double dx = xp[i] - xm[i];
        a += dx * rm[i];
        b += dx * rp[i];
    }
    return (a < 0.0) || (b < 0.0);
}

/* nuts.py:162-175 (in place) */
static void leapfrog(nuts_t* c, double* x, double* r, double* g, int direction) {
    int D = c->D;
    double h = direction * c->eps / 2;
    double e = direction * c->eps;
    for (int i = 0; i < D; ++i) r[i] = r[i] + h * g[i];
    for (int i = 0; i < D; ++i) x[i] = x[i] + e * r[i];
    /* nuts.py:171 gradient; the value from the same evaluation serves
     * nuts.py:122 (the reference calls the target twice). */
    double lp = target_logp_grad(c->model, c->data, D, x, c->phi, g, &c->leaf_lpri, &c->leaf_llik);
    (void)lp;
    for (int i = 0; i < D; ++i) r[i] = r[i] + h * g[i];
    c->nleap++;
}

typedef struct {
    double *xm, *rm, *gm, *xp, *rp, *gp, *xc, *rc; /* each D doubles */
    double c_lpri, c_llik;                         /* density parts at the candidate */
    int n, s;
} tree_t;

static void tree_alloc(tree_t* t, double* buf, int D) {
    t->xm = buf; t->rm = buf + D; t->gm = buf + 2 * D; t->xp = buf + 3 * D;
    t->rp = buf + 4 * D; t->gp = buf + 5 * D; t->xc = buf + 6 * D; t->rc = buf + 7 * D;
}

/* nuts.py:114-150.  (x, r, g) is the edge the sub-tree grows from. */
static void build_tree(nuts_t* c, const double* x, const double* r, const double* g, int direction,
                       int depth, tree_t* out) {
    int D = c->D;
    size_t vb = sizeof(double) * (size_t)D;
    if (depth == 0) {
        /* nuts.py:120-131 */
        memcpy(out->xc, x, vb); memcpy(out->rc, r, vb);
        double* gl = out->gm;
        memcpy(gl, g, vb);
        leapfrog(c, out->xc, out->rc, gl, direction);
        double lp = c->leaf_lpri + c->phi * c->leaf_llik;
        if (!isfinite(lp)) lp = -INFINITY;
        double joint = lp - 0.5 * dotv(out->rc, out->rc, D);
        out->n = (c->logu < joint) ? 1 : 0;
        out->s = ((c->logu - c->delta_max) >= joint) ? 1 : 0;
        out->c_lpri = c->leaf_lpri; out->c_llik = c->leaf_llik;
        memcpy(out->xm, out->xc, vb); memcpy(out->xp, out->xc, vb);
        memcpy(out->rm, out->rc, vb); memcpy(out->rp, out->rc, vb);
        memcpy(out->gp, gl, vb);
        return;
    }
    /* nuts.py:134 first half */
    build_tree(c, x, r, g, direction, depth - 1, out);
    if (out->s == 0) {
        /* nuts.py:136-140 second half from the new outer edge */
        double* buf = c->arena + (size_t)depth * 8 * (size_t)D;
        tree_t t2;
        tree_alloc(&t2, buf, D);
        if (direction == -1) {
            build_tree(c, out->xm, out->rm, out->gm, direction, depth - 1, &t2);
            memcpy(out->xm, t2.xm, vb); memcpy(out->rm, t2.rm, vb); memcpy(out->gm, t2.gm, vb);
        } else {
            build_tree(c, out->xp, out->rp, out->gp, direction, depth - 1, &t2);
            memcpy(out->xp, t2.xp, vb); memcpy(out->rp, t2.rp, vb); memcpy(out->gp, t2.gp, vb);
        }
        /* nuts.py:142-144: one uniform, always */
        double u = rng_uniform(c->rng);
        double denom = (double)(out->n + t2.n);
        if (denom < 1.0) denom = 1.0;
        if (u < (double)t2.n / denom) {
            memcpy(out->xc, t2.xc, vb); memcpy(out->rc, t2.rc, vb);
            out->c_lpri = t2.c_lpri; out->c_llik = t2.c_llik;
        }
        out->n = out->n + t2.n;                                                   /* :146 */
        out->s = (out->s || t2.s || stop_criterion(out->xm, out->xp, out->rm, out->rp, D)) ? 1 : 0; /* :148 */
    }
}

typedef struct {
    double lpri0, llik0, lpri1, llik1;
    int nleap, depth, ndraws, flags;
} nuts_stats_t;

/* nuts.py:58-112 for one particle; x, r are overwritten by the result. */
static void generate_nuts_sample(int model, const double* data, int D, double* x, double* r,
                                 double phi, double eps, int max_depth, double delta_max,
                                 rng_t* rng, nuts_stats_t* st) {
    size_t vb = sizeof(double) * (size_t)D;
    nuts_t c;
    c.model = model; c.D = D; c.data = data; c.phi = phi; c.eps = eps; c.delta_max = delta_max;
    c.rng = rng; c.nleap = 0;
    double* buf = (double*)malloc(17 * vb + (size_t)(max_depth + 2) * 8 * vb);
    c.arena = buf + 17 * D;
    double *xm = buf, *rm = buf + D, *gm = buf + 2 * D, *xp = buf + 3 * D, *rp = buf + 4 * D,
           *gp = buf + 5 * D, *g0 = buf + 6 * D, *xs = buf + 7 * D, *rs = buf + 8 * D;
    tree_t t;
    tree_alloc(&t, buf + 9 * D, D);

    /* nuts.py:66-72 */
    double lpri0, llik0;
    double logp = target_logp_grad(model, data, D, x, phi, g0, &lpri0, &llik0);
    double H0 = logp - 0.5 * dotv(r, r, D);
    c.logu = H0 - rng_exponential(rng);
    st->lpri0 = lpri0; st->llik0 = llik0;
    double s_lpri = lpri0, s_llik = llik0;

    memcpy(xm, x, vb); memcpy(xp, x, vb); memcpy(rm, r, vb); memcpy(rp, r, vb);
    memcpy(gm, g0, vb); memcpy(gp, g0, vb); memcpy(xs, x, vb); memcpy(rs, r, vb);

    int depth = 0, n = 1, stop = 0;
    while (stop == 0) {                                                           /* :89 */
        int direction = (rng_uniform(rng) < 0.5) ? 1 : -1;                        /* :91 */
        if (direction == -1) {
            build_tree(&c, xm, rm, gm, direction, depth, &t);
            memcpy(xm, t.xm, vb); memcpy(rm, t.rm, vb); memcpy(gm, t.gm, vb);
        } else {
            build_tree(&c, xp, rp, gp, direction, depth, &t);
            memcpy(xp, t.xp, vb); memcpy(rp, t.rp, vb); memcpy(gp, t.gp, vb);
        }
        if (t.s == 0) {                                                           /* :99 short-circuit */
            double ratio = (double)t.n / (double)n;
            if (ratio > 1.0) ratio = 1.0;
            if (rng_uniform(rng) < ratio) {
                memcpy(xs, t.xc, vb); memcpy(rs, t.rc, vb);
                s_lpri = t.c_lpri; s_llik = t.c_llik;
            }
        }
        n += t.n;                                                                 /* :103 */
        stop = (t.s || stop_criterion(xm, xp, rm, rp, D)) ? 1 : 0;                /* :105 */
        depth += 1;
        if (depth > max_depth) break;                                             /* :109 */
    }
    memcpy(x, xs, vb); memcpy(r, rs, vb);
    st->lpri1 = s_lpri; st->llik1 = s_llik;
    st->nleap = c.nleap; st->depth = depth; st->ndraws = (int)rng->pos;
    st->flags = rng->overflow ? 1 : 0;
    free(buf);
}

/* NUTSProposal.rvs (nuts.py:34-56) over N particles, row-major [N, D].
 * rng_mode 0: tape[tape_off[i] .. tape_off[i+1]) are particle i's recorded
 * draws (first the Exp(1), then uniforms in consumption order).
 * rng_mode 1: Philox keyed (seed, iter, particle_base + i).
 * Any of the per-particle outputs may be NULL. */
int oracle_nuts_rvs(int model, const double* data, int64_t N, int D, const double* x, const double* r,
                    double phi, double eps, int max_depth, double delta_max, int rng_mode,
                    const double* tape, const int64_t* tape_off, uint64_t seed, uint32_t iter,
                    int64_t particle_base, double* x_new, double* r_new, double* lpri0, double* llik0,
                    double* lpri1, double* llik1, int32_t* nleap, int32_t* depth, int32_t* ndraws,
                    int32_t* flags) {
    if (D > ORACLE_MAX_D) return -1;
    /* serial over particles, as the reference (nuts.py:50) */
    for (int64_t i = 0; i < N; ++i) {
        rng_t g;
        memset(&g, 0, sizeof g);
        g.mode = rng_mode;
        if (rng_mode == 0) { g.tape = tape + tape_off[i]; g.len = tape_off[i + 1] - tape_off[i]; }
        g.seed = seed; g.iter = iter; g.particle = (uint32_t)(particle_base + i);
        memcpy(x_new + i * D, x + i * D, sizeof(double) * (size_t)D);
        memcpy(r_new + i * D, r + i * D, sizeof(double) * (size_t)D);
        nuts_stats_t st;
        generate_nuts_sample(model, data, D, x_new + i * D, r_new + i * D, phi, eps, max_depth,
                             delta_max, &g, &st);
        if (lpri0) lpri0[i] = st.lpri0;
        if (llik0) llik0[i] = st.llik0;
        if (lpri1) lpri1[i] = st.lpri1;
        if (llik1) llik1[i] = st.llik1;
        if (nleap) nleap[i] = st.nleap;
        if (depth) depth[i] = st.depth;
        if (ndraws) ndraws[i] = st.ndraws;
        if (flags) flags[i] = st.flags;
    }
    return 0;
}
