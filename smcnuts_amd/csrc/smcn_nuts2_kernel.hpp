// A/B builds only (-DSMCN_VARIANTS, tools/build_variant.py): the second-generation NUTS kernel for models whose state is
// REPLICATED on the lanes of a group (arma on 4 + 4 lanes, PRMwCD replicated).  The product runs arma in nuts3_kernel
// (smcn_nuts3.hpp) and PRMwCD / Gaussians in nuts_kernel (smcn_nuts.hpp); the record layouts and the prep / post kernels
// this kernel shares with the lane kernel stay in smcn_nuts2.hpp.
#pragma once
#include "smcn_nuts2.hpp"

namespace smcn {

// TAPE = true is the test build that replays recorded draws from global memory; the
// production (Philox) build contains no global load inside the tree loop except the
// prefetch of the next input record, so no s_waitcnt vmcnt ever lands on a young request.
template <class Model, bool TAPE>
__global__ void __launch_bounds__(kNutsBlock, Model::MIN_WAVES) nuts2_kernel(Nuts2Args a) {
    static_assert(!Model::DIST, "nuts2_kernel: replicated-state models only");
    constexpr int G = Model::G, DL = Model::DL, VP = n2_vp(DL);
    constexpr int L = Model::N2_LDS_LEVELS;                           // tree-stack levels in LDS
    constexpr int SLOT = n2_slot_doubles(DL, L), INSZ = n2_in_doubles(DL), OUTSZ = n2_out_doubles(DL);
    constexpr int INCH = INSZ / 2, OUTCH = OUTSZ / 2;                 // 16-byte chunks
    constexpr int PRE = (INCH + G - 1) / G;                           // chunks of the input record per lane
    static_assert(2 * VP <= 2 * G && PRE <= 2, "x0, r0 fit one chunk per lane; only the slice exponential may lie beyond");
    constexpr int REC = 0, R_PRI1 = 2 * VP, R_PRI0 = 2 * VP + 2, R_ST = 2 * VP + 4;
    constexpr int EM = OUTSZ, EP = EM + 3 * VP, FIRST = EP + 3 * VP, CAND = FIRST + L * 2 * VP, CREC = 2 * VP + 4,
                  UBUF = CAND + L * CREC;
    constexpr int OVF = n2_ovf_doubles(DL, L), OFIRST = 0, OCAND = (10 - L) * 2 * VP;
    constexpr int GR = G < 8 ? G : 8;                                 // lanes that refill: 2 uniforms each
    static_assert(UBUF + 32 <= SLOT, "slot layout");
    static_assert(G >= 2 && L >= 1 && L <= 10, "group size / LDS levels");
    enum { NEED = 0, INIT = 1, LEAF = 2, DONE = 3 };

    extern __shared__ double lds[];
    constexpr int MSH = (Model::SHARED + 1) & ~1;
    const int lane = (int)(threadIdx.x & 63u);
    const int lg = lane & (G - 1);
    double* const slot = lds + MSH + (threadIdx.x / G) * SLOT;
    // overflow levels (index >= L) of this group: global memory, never touched by trees of depth <= L
    auto ovf_ptr = [&]() -> double* {   // rare path: not worth two live registers
        return a.ovf + ((int64_t)blockIdx.x * (kNutsBlock / G) + threadIdx.x / G) * OVF;
    };
    using d2 = double __attribute__((ext_vector_type(2)));

    Model model;
    model.init(a.mdata, lg, lds);
    const int64_t N = a.N;
    const double eps = a.eps, phi = a.phi;

    // ---- vector moves: VP/2 16-byte accesses; stores by the group leader ------
    // The *_p forms take an address-space-qualified pointer: LDS (ds_read/ds_write) for the levels
    // kept on chip, global for the overflow levels -- two instantiations, never a flat access.
    using lptr = double*;                                       // derived from `lds`: inferred LDS
    using gptr = __attribute__((address_space(1))) double*;
    using gptr2 = __attribute__((address_space(1))) d2*;
    auto ld16 = [](auto p) -> d2 {
        if constexpr (__is_same(decltype(p), gptr)) return *(gptr2)p;
        else return *reinterpret_cast<const d2*>(p);
    };
    auto st16 = [](auto p, d2 v) {
        if constexpr (__is_same(decltype(p), gptr)) *(gptr2)p = v;
        else *reinterpret_cast<d2*>(p) = v;
    };
    const lptr lslot = slot;
    auto vstore_p = [&](auto dst, const double (&v)[DL]) {
        if (lg == 0) {
#pragma unroll
            for (int i = 0; i < VP / 2; ++i) {
                d2 t;
                t.x = v[2 * i];
                t.y = (2 * i + 1 < DL) ? v[2 * i + 1 < DL ? 2 * i + 1 : 0] : 0.0;
                st16(dst + 2 * i, t);
            }
        }
    };
    auto vload_p = [&](auto src, double (&v)[DL]) {
#pragma unroll
        for (int i = 0; i < VP / 2; ++i) {
            const d2 t = ld16(src + 2 * i);
            v[2 * i] = t.x;
            if (2 * i + 1 < DL) v[2 * i + 1 < DL ? 2 * i + 1 : 0] = t.y;
        }
    };
    auto vstore = [&](int off, const double (&v)[DL]) { vstore_p(lslot + off, v); };
    auto vload = [&](int off, double (&v)[DL]) { vload_p(lslot + off, v); };
    auto store2_p = [&](auto dst, double u, double v) {
        if (lg == 0) { d2 t; t.x = u; t.y = v; st16(dst, t); }
    };
    auto store2 = [&](int off, double u, double v) { store2_p(lslot + off, u, v); };
    auto copy_rec_p = [&](auto src, auto dst) {   // (x, r, lpri, llik): 2 VP + 2 doubles
#pragma unroll
        for (int i = 0; i < VP + 1; ++i) {
            const d2 t = ld16(src + 2 * i);
            if (lg == 0) st16(dst + 2 * i, t);
        }
    };
    // the parked candidate of level m / the first leaf of level s (1-based), wherever they live:
    // f(pointer) is instantiated once for LDS and once for the overflow area
    auto with_cand = [&](int m, auto&& f) {
        if (L == 10 || m < L) f(lslot + (CAND + m * CREC));
        else f((gptr)ovf_ptr() + (OCAND + (m - L) * CREC));
    };
    auto with_first = [&](int s, auto&& f) {
        if (L == 10 || s - 1 < L) f(lslot + (FIRST + (s - 1) * 2 * VP));
        else f((gptr)ovf_ptr() + (OFIRST + (s - 1 - L) * 2 * VP));
    };
    // (x_cur - x_other) . r_other  and  . r_cur     (nuts.py:159-160 up to the direction's sign)
    auto uturn_dots = [&](int off, const double (&xc)[DL], const double (&rc)[DL], double& A, double& B) {
        double xo[DL], ro[DL];
        vload(off, xo);
        vload(off + VP, ro);
        A = 0.0; B = 0.0;
#pragma unroll
        for (int i = 0; i < DL; ++i) {
            const double d = xc[i] - xo[i];
            A = fma(d, ro[i], A);
            B = fma(d, rc[i], B);
        }
    };
    auto is_uturn = [](double A, double B, int dir) {
        // dir > 0: minus = other, plus = current: (A < 0) || (B < 0); dir < 0: dx, roles negate
        return dir > 0 ? ((A < 0.0) || (B < 0.0)) : ((B > 0.0) || (A > 0.0));
    };

    // ---- per-group state ---------------------------------------------------------
    int phase = NEED;
    int64_t p = 0, pnext = -1;
    d2 pre[1];                    // this lane's chunk of the prefetched input record
    pre[0].x = 0.0; pre[0].y = 0.0;
    double x[DL], r[DL], g[DL];
    double logu = 0.0;
    int j = 0, i = 0, dir = 1, n = 1, nleap = 0;
    int b = 0;                       // transition index of the current particle (a.B per particle)
    uint32_t q = 0, qfill = 0;
    int64_t toff = 0, tlen = 0;
    bool overflow = false;
#pragma unroll
    for (int k = 0; k < DL; ++k) { x[k] = 0.0; r[k] = 0.0; g[k] = 0.0; }

    // Work queue.  A single queue word sustains only ~90-150 claims/us on this chip, which at one
    // claim per particle was the limit of the whole kernel (65 536 claims ~ 0.45 ms).  Each
    // wavefront therefore claims CHUNKS of kChunk particle indices with one atomic (lane 0, result
    // consumed a whole chunk later) and hands indices to its groups from wave-uniform counters;
    // each group keeps the input record of its next particle in flight (`pnext` / `pre`).
    constexpr uint32_t kChunk = 64 / G > 8 ? 64 / G : 8;   // >= the leaders of a wave: one assign() spans <= 2 chunks
    const char* const in_base = reinterpret_cast<const char*>(a.in);
    char* const out_base = reinterpret_cast<char*>(a.out);
    uint32_t w_next = 0, w_end = 0;   // wave-uniform: unassigned indices of the current chunk
    unsigned int c_claim = 0;         // lane 0: base of the chunk claimed ahead (pending atomic)
    auto claim_chunk = [&]() {
        if (lane == 0) c_claim = atomicAdd(a.queue, kChunk);
    };
    // convergent code only: one index for every lane with `want` set (group leaders)
    auto assign = [&](bool want) -> uint32_t {
        const unsigned long long mask = __ballot(want);
        uint32_t t = 0xffffffffu;
        if (mask != 0ull) {
            const uint32_t cnt = (uint32_t)__popcll(mask);
            const uint32_t rank = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
            t = w_next + rank;
            if (w_next + cnt > w_end) {   // wave-uniform: continue in the chunk claimed ahead
                const uint32_t nb = (uint32_t)__builtin_amdgcn_readfirstlane((int)c_claim);
                if (t >= w_end) t = nb + (t - w_end);
                w_next = nb + (w_next + cnt - w_end);
                w_end = nb + kChunk;
                claim_chunk();
            } else {
                w_next += cnt;
            }
        }
        return t;
    };
    auto request_unit = [&](int64_t idx, int bb) {   // start loading the input record of (particle idx, transition bb)
        if (idx < N) {
            const uint32_t off = ((uint32_t)bb * (uint32_t)N + (uint32_t)idx) * (uint32_t)(INSZ * 8) + 16u * (uint32_t)lg;
            if (lg < INCH) pre[0] = *reinterpret_cast<const d2*>(in_base + off);
        }
    };
    auto refill = [&]() {         // 2 GR uniforms: block (qfill/2 + lg) of this particle's NUTS stream
        if (lg < GR) {
            const u32x4 o = philox4x32_10({(qfill >> 1) + (uint32_t)lg, (uint32_t)(a.particle_base + p),
                                           a.iter + (uint32_t)b, kStreamNuts}, (uint32_t)a.seed,
                                          (uint32_t)(a.seed >> 32));
            d2 t;
            t.x = u53(o.a, o.b);
            t.y = u53(o.c, o.d);
            *reinterpret_cast<d2*>(slot + UBUF + ((qfill + 2u * lg) & 31u)) = t;
        }
        qfill += 2u * GR;
    };
    auto draw = [&]() -> double {
        double v;
        if constexpr (TAPE) {
            if ((int64_t)q < tlen) v = a.tape[toff + q];
            else { v = 0.5; overflow = true; }
        } else {
            v = slot[UBUF + (q & 31u)];
        }
        ++q;
        return v;
    };

    claim_chunk();
    w_next = (uint32_t)__builtin_amdgcn_readfirstlane((int)c_claim);
    w_end = w_next + kChunk;
    claim_chunk();
    pnext = (int64_t)(unsigned int)group_read_i<G>((int)assign(lg == 0), 0);
    request_unit(pnext, 0);
    bool have_out = false;
    uint32_t out_off = 0;
#ifdef SMCN_PROFILE   // residency census: blocks alive at the same time
    if (threadIdx.x == 0) {
        const unsigned int now = atomicAdd(a.queue + 1, 1u) + 1u;
        atomicMax(a.queue + 2, now);
    }
#endif
    PROF_DECL;
    for (;;) {
        PROF(7);
        // ---- start the next particle ------------------------------------------------
        // A unit = one NUTS transition.  A particle's a.B transitions run back to back on the same
        // group (the sample of one is the start of the next); the queue is asked for the following
        // particle when the last transition of the current one starts.
        const bool starting = (phase == NEED);
        const bool cont = starting && have_out && (b + 1 < a.B);
        const bool newp = starting && !cont && (pnext < N);
        const int nb = cont ? b + 1 : 0;
        const uint32_t my_next = assign((cont || newp) && (nb == a.B - 1) && lg == 0);
        if (starting) {
            const bool more = cont || newp;
            if (more) {
                if (!cont) p = pnext;
                b = nb;
                // stage the prefetched record through the (free) edge area, then read it replicated
                if (lg < INCH) *reinterpret_cast<d2*>(slot + EM + 2 * lg) = pre[0];
                wave_exchange_fence();                       // every lane staged its own chunk(s)
                if (cont) vload(REC, x);                     // continue from the sample just drawn
                else vload(EM, x);
                vload(EM + VP, r);
                double e0;
                if constexpr (PRE == 1) {
                    e0 = slot[EM + 2 * VP];
                } else {   // narrow groups: the chunk beyond the prefetched ones, consumed after the first evaluation
                    e0 = *reinterpret_cast<const double*>(in_base + ((uint32_t)b * (uint32_t)N + (uint32_t)p) * (uint32_t)(INSZ * 8) + 16u * VP);
                }
                logu = e0;                                   // raw (no arithmetic: the load may still be in flight);
                                                             // becomes H0 - e0 after the first evaluation
                q = 1; qfill = 0; overflow = false; nleap = 0;
                if constexpr (TAPE) { toff = a.tape_off[p]; tlen = a.tape_off[p + 1] - toff; }
                if (b == a.B - 1) {                          // next unit: first transition of the next particle
                    pnext = (int64_t)(unsigned int)group_read_i<G>((int)my_next, 0);
                    request_unit(pnext, 0);
                } else {
                    request_unit(p, b + 1);
                }
            }
            if (have_out) {   // the finished transition's record leaves last: nothing waits on these stores
                wave_exchange_fence();   // lane c reads chunk c of what the group leader wrote
                for (int c = lg; c < OUTCH; c += G)
                    *reinterpret_cast<d2*>(out_base + out_off + 16u * c) = *reinterpret_cast<const d2*>(slot + REC + 2 * c);
                have_out = false;
            }
            phase = more ? INIT : DONE;
        }
        if (__ballot(phase != DONE) == 0ull) break;
        // ---- keep >= 16 uniforms ahead (a tree level consumes at most 12 per leaf) -----
        if constexpr (!TAPE) {
            // The Philox call is issued for the whole wave whenever ANY group runs low, so narrow groups
            // (8 uniforms per call) top up together: every group with room takes part, and the next
            // call comes when the first of them has used 8 more, not at every leaf.
#pragma unroll
            for (int rr = 0; rr < 16 / (2 * GR); ++rr) {     // GR = 8: one refill of 16; GR = 4: up to two of 8
                const int avail = (int)(qfill - q);
                const bool low = phase != DONE && avail < 16;
                bool go = low;
                if constexpr (GR < 8) go = phase != DONE && avail + 2 * GR <= 32 && __ballot(low) != 0ull;
                if (go) {
                    refill();
#ifdef SMCN_DOUBLE_REFILL   // ablation build: the same uniforms generated twice (prices the in-kernel Philox)
                    qfill -= 2u * GR;
                    refill();
#endif
                }
            }
            wave_exchange_fence();   // a draw reads what any lane of the group generated
        }
        PROF(0);

        // ---- leapfrog, first half (nuts.py:169-170) --------------------------------
        const double e = dir * eps, h = dir * eps / 2;
        if (phase == LEAF) {
#pragma unroll
            for (int k = 0; k < DL; ++k) r[k] = r[k] + h * g[k];
#pragma unroll
            for (int k = 0; k < DL; ++k) x[k] = x[k] + e * r[k];
        }
        double lpri, llik, gp[DL], gl[DL];
        PROF(1);
        model.eval(x, lpri, llik, gp, gl);
#ifdef SMCN_DOUBLE_EVAL   // ablation build: a second, discarded evaluation (prices the evaluation alone)
        {
            double x2[DL], lp2, ll2, gp2[DL], gl2[DL];
#pragma unroll
            for (int k = 0; k < DL; ++k) { x2[k] = x[k]; asm volatile("" : "+v"(x2[k])); }
            model.eval(x2, lp2, ll2, gp2, gl2);
            asm volatile("" ::"v"(lp2), "v"(ll2));
#pragma unroll
            for (int k = 0; k < DL; ++k) asm volatile("" ::"v"(gp2[k]), "v"(gl2[k]));
        }
#endif
        PROF(2);
        double lp = lpri + phi * llik;
        const bool bad = !finite_d(lp);   // bridgestan.py:47-49,79-80
        lp = bad ? -kInf : lp;
#pragma unroll
        for (int k = 0; k < DL; ++k) g[k] = bad ? -kInf : fma(phi, gl[k], gp[k]);

        bool start_doubling = false;
        if (phase == LEAF) {
            // ---- second half kick (nuts.py:173), leaf tests (:123-125) ----------------
            double kin = 0.0;
#pragma unroll
            for (int k = 0; k < DL; ++k) { r[k] = r[k] + h * g[k]; }
#pragma unroll
            for (int k = 0; k < DL; ++k) kin = fma(r[k], r[k], kin);
            ++nleap;
            const double joint = lp - 0.5 * kin;
            int nsub = (logu < joint) ? 1 : 0;
            bool ssub = (logu - a.delta_max) >= joint;
            // the sub-tree's candidate is kept BY REFERENCE: -1 = this leaf (x, r, lpri, llik in
            // registers), m >= 0 = the record parked in CAND[m]; it is only copied when parked
            // one level up or accepted at the top
            int csrc = -1;
            if (j > 0 && (i & 1) == 0) {
                const int s = (i == 0) ? j : (__ffs(i) - 1);
                with_first(s, [&](auto fp) { vstore_p(fp, x); vstore_p(fp + VP, r); });
            }
            PROF(3);
            // ---- merges (nuts.py:134-148), the top level (:99-105) being level j -------
            bool done = false, stop = false;
            int m = 0;
            for (;;) {
                if (ssub) {
                    // unwinding: every ancestor whose SECOND half stopped still draws (:142)
                    q += (uint32_t)__popc((unsigned)(i >> m) & ((1u << (j - m)) - 1u));
                    done = true; stop = true;
                    break;
                }
                if (m == j) {
                    // top level: accept with prob min(1, n'/n) (:99), U-turn on the outer edges (:105)
                    const double u = draw();
                    if (nsub >= n || fma(u, (double)n, -(double)nsub) < 0.0) {
                        if (csrc < 0) {
                            vstore(REC, x); vstore(REC + VP, r);
                            store2(REC + R_PRI1, lpri, llik);
                        } else {
                            with_cand(csrc, [&](auto cp) { copy_rec_p(cp, lslot + REC); });
                        }
                    }
                    double A, B;
                    uturn_dots(dir > 0 ? EM : EP, x, r, A, B);
                    stop = is_uturn(A, B, dir);
                    done = true;
                    break;
                }
                if (((i >> m) & 1) == 0) {   // first half of level m+1: park it
                    with_cand(m, [&](auto crec) {
                        if (csrc < 0) {
                            vstore_p(crec, x);
                            vstore_p(crec + VP, r);
                            store2_p(crec + 2 * VP, lpri, llik);
                        } else {
                            with_cand(csrc, [&](auto cp) { copy_rec_p(cp, crec); });
                        }
                        if (lg == 0) crec[2 * VP + 2] = (double)nsub;
                    });
                    break;
                }
                // one LDS round trip per level: the uniform, the parked first half and the
                // sub-tree's first leaf are all requested before anything is consumed
                const int i0 = (i >> (m + 1)) << (m + 1);
                const int s = (i0 == 0) ? j : (__ffs(i0) - 1);
                const double u = draw();     // :142, always
                double fx[DL], fr[DL];
                int n1 = 0;
                with_cand(m, [&](auto crec) { n1 = (int)crec[2 * VP + 2]; });
                with_first(s, [&](auto fp) { vload_p(fp, fx); vload_p(fp + VP, fr); });
                const int den = (n1 + nsub) > 1 ? (n1 + nsub) : 1;
                const bool keep = !(fma(u, (double)den, -(double)nsub) < 0.0);   // keep the first half's candidate
                csrc = keep ? m : csrc;
                nsub += n1;                  // :146
                double A = 0.0, B = 0.0;
#pragma unroll
                for (int k = 0; k < DL; ++k) {
                    const double d = x[k] - fx[k];
                    A = fma(d, fr[k], A);
                    B = fma(d, r[k], B);
                }
                ssub = is_uturn(A, B, dir);  // :148
                ++m;
            }
            PROF(4);
            if (!done) {
                ++i;
            } else {
                n += nsub;                   // :103  (unused after a stop)
                const int eo = (dir > 0) ? EP : EM;
                ++j;
                if (stop || j > a.max_depth) {   // :89,109 -> emit the output record
                    if (lg == 0) {
                        d2 t;
                        const unsigned long long s0 = (unsigned long long)(unsigned)nleap | ((unsigned long long)(unsigned)j << 32);
                        const unsigned long long s1 = (unsigned long long)q | ((unsigned long long)(overflow ? 1u : 0u) << 32);
                        t.x = __longlong_as_double((long long)s0);
                        t.y = __longlong_as_double((long long)s1);
                        *reinterpret_cast<d2*>(slot + REC + R_ST) = t;
                    }
                    have_out = true;
                    out_off = ((uint32_t)b * (uint32_t)N + (uint32_t)p) * (uint32_t)(OUTSZ * 8);
                    phase = NEED;
                } else {
                    vstore(eo, x); vstore(eo + VP, r); vstore(eo + 2 * VP, g);
                    start_doubling = true;
                }
            }
            PROF(5);
        } else if (phase == INIT) {
            // ---- nuts.py:66-87 --------------------------------------------------------
            double kin = 0.0;
#pragma unroll
            for (int k = 0; k < DL; ++k) kin = fma(r[k], r[k], kin);
            logu = (lp - 0.5 * kin) - logu;      // H0 - Exp(1)
            store2(REC + R_PRI0, lpri, llik);     // the record's start density (nothing else touches this field)
            vstore(REC, x); vstore(REC + VP, r);
            store2(REC + R_PRI1, lpri, llik);
            j = 0; n = 1;
            dir = 0;                              // both edges are (x0, r0, g0)
            start_doubling = true;
            phase = LEAF;
        }
        if (start_doubling) {
            // ---- nuts.py:91: direction; the moving state becomes that edge ---------------
            const int nd = (draw() < 0.5) ? 1 : -1;
            if (dir == 0) {
                const int oo = (nd > 0) ? EM : EP;   // the edge that stays behind
                vstore(oo, x); vstore(oo + VP, r); vstore(oo + 2 * VP, g);
            } else if (nd != dir) {
                const int so = (nd > 0) ? EP : EM;
                vload(so, x); vload(so + VP, r); vload(so + 2 * VP, g);
            }
            dir = nd;
            i = 0;
            PROF(6);
        }
    }
    PROF_FLUSH(a);
#ifdef SMCN_PROFILE
    __syncthreads();
    if (threadIdx.x == 0) atomicSub(a.queue + 1, 1u);
#endif
}

}  // namespace smcn
