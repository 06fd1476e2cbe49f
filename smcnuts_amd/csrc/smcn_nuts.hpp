// The NUTS proposal for all particles of a shard in ONE launch.
//
// Replaces NUTSProposal.rvs / generate_nuts_samples / build_tree /
// NUTSLeapfrog / stop_criterion (smcnuts/proposal/nuts.py:34-175), which loop
// over particles in Python and recurse per tree.
//
// Mapping (gfx950, wave64): a group of G lanes owns one particle; a wavefront
// carries 64/G particles whose trees advance in lock-step through the only
// expensive part (leapfrog + target value/gradient, evaluated cooperatively by
// the G lanes), while the cheap, divergent tree bookkeeping is predicated per
// group.  Groups pull particles from a global queue until it is empty, so
// divergent tree sizes (1..2047 leapfrogs) are absorbed inside the wavefront.
//
// The recursion of build_tree is unrolled into a per-leaf state machine.  For
// the doubling of depth j the 2^j leaves are generated in order; after leaf i,
// every completed sub-tree (level m = 0,1,.. while bit m of i is set) is merged
// with its stored first half exactly as nuts.py:136-148 does (one uniform per
// merge, candidate swap with probability n''/max(n'+n'',1), U-turn test between
// the sub-tree's first leaf and the current leaf).  Per particle the stack
// holds, in LDS: both outer edges (x, r, grad), the current sample, one
// "first leaf" (x, r) per level and one pending candidate (x, r, density parts,
// n') per level.
#pragma once
#include "smcn_models.hpp"
#include <type_traits>

namespace smcn {

constexpr int kNutsBlock = 256;
constexpr int kMaxLevels = 10;  // MAX_TREE_DEPTH, nuts.py:4

struct NutsArgs {
    int64_t N;
    int64_t particle_base;
    const double* mdata;
    const double* x;  // [D][N]
    const double* r;  // [D][N]
    double* x_new;
    double* r_new;
    double *lpri0, *llik0, *lpri1, *llik1;
    int32_t *nleap, *depth, *ndraws, *flags;
    unsigned int* queue;  // particle work queue head
    double eps, phi, delta_max;
    int max_depth;
    uint64_t seed;
    uint32_t iter;
    const double* tape;       // tape mode (tests) if non-null
    const int64_t* tape_off;  // [N+1]
    unsigned long long* prof; // SMCN_PROFILE builds: per-section cycle sums
    double* scratch;          // HBM tree stacks (one slot per resident group) for large D
    // wave-per-particle kernels (one wavefront holds the whole particle): |r|^2 at the start, |r'|^2 of the selected
    // sample and "every coordinate moved" (smc_sampler.py:97), so that the re-weighting and the acceptance statistic
    // need no pass over r, r', x, x' (4 x N x D doubles)
    double* kin0 = nullptr;
    double* kin1 = nullptr;
    int32_t* moved = nullptr;
    // Two-phase launches (kernels with register-resident edges): a launch lasts as long as its longest tree, so a tree
    // that still wants a doubling after `jcap` of them is PARKED at that boundary -- where the tree stack is empty and its
    // whole state is the two edges, the selected sample and a few scalars (8 D + 8 doubles: resume[p]) -- and finished by
    // a second launch (`resume_in`), which may give it more lanes.  pend[0] counts the parked trees, pend[1 ..] names them.
    int jcap = 0;
    int resume_in = 0;
    double* resume = nullptr;
    unsigned int* pend = nullptr;
    // several particles per wavefront (G < 64): a group takes its next particle only in a loop iteration whose number is a
    // multiple of step_align (a power of two) -- a tree takes 2^depth iterations, so the groups of a wavefront then reach
    // the leaves with many merges in the same iterations instead of one group or another in every iteration
    int step_align = 1;
    // One more park level INSIDE the launch (two-phase kernels, first launch only): a tree that still wants a doubling after
    // `mq_b` of them (mq_b < jcap) is parked as above, but on a list this launch's own groups take trees from once the
    // fresh particles have all been handed out.  Every tree then gets its short first part early and the launch ends on
    // pieces of at most 2^(jcap-1) leaves instead of on whole trees (the launch is its last tree's length
    // behind the point where the queue runs dry).  mq[0] entries allocated, mq[1] taken, mq[5] written and unclaimed, mq[16 + k] = particle + 1 of
    // entry k (0: not written yet).  Producer and consumer may sit on different XCDs, whose L2s are not coherent.  Instead of a
    // fence pair per hand-over (an L2 write-back and an invalidate, microseconds each for the whole wavefront, 50 000 times a
    // launch) the record is written THROUGH (agent-scope stores, `sc1`), the wavefront drains its stores, then the entry is
    // written through; the taking group polls the entry and reads the record with agent-scope (`sc1`) loads, which no L1
    // serves.  Records sit on 128-byte lines of their own (no line is shared by two hand-overs) and are written once and
    // read once per launch.
    unsigned int* mq = nullptr;
    double* mq_rec = nullptr;     // [N][128]: the inner level's records, one per particle on cache lines of its own
    int mq_b = 0;
    // nuts_wave_kernel: r / r_new are particle-major ([N][D]: a particle's momentum is one contiguous row) instead of [D][N]
    int r_pm = 0, r_new_pm = 0;
};

#ifdef SMCN_PROFILE
__device__ __forceinline__ unsigned long long prof_stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
// (the accumulators are per lane and lane 0's are reported: a stamp inside a divergent region counts for lane 0 only
// when lane 0 is in it -- the time of a region it skips lands on the next stamp it does execute)
#define PROF_DECL                                                                                         \
    unsigned long long pt_ = prof_stamp(), pacc_[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};        \
    bool prof_on_ = true
#define PROF_ON(c) prof_on_ = (c)   /* count the following stamps only while c holds (wave-uniform) */
#define PROF(sec)                                   \
    do {                                            \
        const unsigned long long n_ = prof_stamp(); \
        pacc_[sec] += prof_on_ ? n_ - pt_ : 0ull;   \
        pt_ = n_;                                   \
    } while (0)
#define PROF_FLUSH(a)                                                               \
    do {                                                                            \
        if ((threadIdx.x & 63u) == 0) {                                             \
            for (int s_ = 0; s_ < 14; ++s_) atomicAdd(&(a).prof[s_], pacc_[s_]);    \
        }                                                                           \
    } while (0)
#else
#define PROF_DECL
#define PROF_ON(c)
#define PROF(sec)
#define PROF_FLUSH(a)
#endif

// doubles of LDS per particle: 6 edge vectors, 2 sample vectors + 2 scalars,
// 10 x (first leaf x, r), 10 x (candidate x, r, lpri, llik, n'); padded so
// that consecutive slots start 2 banks apart (broadcast reads of the 64/G
// groups of a wavefront then hit distinct banks).
__host__ __device__ constexpr int nuts_slot_doubles(int VS) {
    int n = 48 * VS + 32;
    while (n % 32 != 1) ++n;
    return n;
}

// HBM_STACK: the per-particle tree stack (48 D + 32 doubles; 98.6 KB at D = 256) does not
// fit in LDS; each resident group owns a slot of a global scratch buffer instead (lane-
// contiguous vectors, so every access is a coalesced 512-byte row).
// With the stack in HBM the first Model::LDS_LEVELS levels (the ones touched every 2nd / 4th leaf) still live
// in LDS: 1/2 + 1/4 + .. of all parks and merges never leave the CU.
__host__ __device__ constexpr int nuts_hybrid_lds_doubles(int VS, int levels) { return levels * (2 * VS + 2 * VS + 3); }

__host__ __device__ constexpr int RL_X0(bool wide, int dl) { return wide ? dl : 1; }
// which Model / stack combinations write NutsArgs::kin0, kin1, moved
template <class Model, bool HBM_STACK>
constexpr bool nuts_kernel_writes_stats() { return HBM_STACK && Model::DIST && Model::DL <= 4 && Model::G == 64; }

// models that ask for the hybrid (LDS levels + HBM slot) stack whatever their size
template <class M, class = void>
struct model_hybrid_always { static constexpr bool value = false; };
template <class M>
struct model_hybrid_always<M, std::enable_if_t<M::HYBRID_ALWAYS>> { static constexpr bool value = true; };
// models whose kernels can park a tree at a doubling boundary / take a parked one up (NutsArgs::jcap, resume_in): opt-in,
// the extra live state costs registers (the D = 256 Gaussian kernel spilled 252 bytes per lane with it)
template <class M, class = void>
struct model_two_phase { static constexpr bool value = false; };
template <class M>
struct model_two_phase<M, std::enable_if_t<M::TWO_PHASE>> { static constexpr bool value = true; };
// models whose trees are long enough for the groups of a wavefront to start them in step (NutsArgs::step_align): opt-in
template <class M, class = void>
struct model_step_align { static constexpr int value = 1; };
template <class M>
struct model_step_align<M, std::enable_if_t<(M::STEP_ALIGN > 1)>> { static constexpr int value = M::STEP_ALIGN; };
// models with eval_partial / finish (the value is a sum of per-lane shares: GaussModel)
template <class M, class = void>
struct model_has_partial { static constexpr bool value = false; };
template <class M>
struct model_has_partial<M, std::enable_if_t<M::HAS_PARTIAL>> { static constexpr bool value = true; };

// TWO_PHASE: the instantiation that can park a tree at a doubling boundary / take a parked one up (NutsArgs::jcap,
// resume_in).  A kernel of its own: the extra live state costs registers (PRMwCD: 156 instead of 20 bytes of scratch per
// lane, the D = 256 Gaussian kernel 252 instead of none), which one-launch runs should not pay.
template <class Model, bool HBM_STACK = false, bool TWO_PHASE = false>
__global__ void __launch_bounds__(kNutsBlock, Model::MIN_WAVES) nuts_kernel(NutsArgs a) {
    constexpr int G = Model::G, DL = Model::DL;
    constexpr bool DIST = Model::DIST;
    constexpr int VS = DIST ? G * DL : DL;
    constexpr int SLOT = nuts_slot_doubles(VS);
    constexpr int EM = 0, EP = 3 * VS, SEL = 6 * VS, SELP = 8 * VS, FIRST = 8 * VS + 2,
                  CAND = FIRST + 20 * VS, CREC = 2 * VS + 3;
    enum { NEED = 0, INIT = 1, LEAF = 2, DONE = 3 };

    extern __shared__ double lds[];
    const int lane = (int)(threadIdx.x & 63u);
    const int lg = lane & (G - 1);
    constexpr int MSH = (Model::SHARED + 1) & ~1;   // block-shared model data first
    // (G == 1, one lane per particle: the HBM slots of a block are LANE-INTERLEAVED -- element k of lane t at
    //  [k][t] -- so that lanes at the same place of their trees touch one 512-byte row; SK = the stride between elements)
    constexpr int SK = (HBM_STACK && G == 1) ? kNutsBlock : 1;
    double* const slot = HBM_STACK
        ? (G == 1 ? a.scratch + (int64_t)blockIdx.x * kNutsBlock * SLOT + threadIdx.x
                  : a.scratch + ((int64_t)blockIdx.x * (kNutsBlock / G) + threadIdx.x / G) * SLOT)
        : lds + MSH + (threadIdx.x / G) * SLOT;

    // Pointers used once per tree (inputs, outputs, statistics) are re-read from the kernel-argument segment where they
    // are needed instead of occupying ~30 scalar registers across the leaf loop (the PRMwCD kernel spilled scalars to
    // vector lanes on every iteration); the laundering keeps the loads from being hoisted back out of the loop.
    auto kargs = [&]() __attribute__((always_inline)) {
        using kptr = const __attribute__((address_space(4))) NutsArgs*;
        kptr kp = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp));
        return kp;
    };
    Model model;
    model.init(a.mdata, lg, lds);
    const int D = model.dim();
    const int64_t N = a.N;
    const double eps = a.eps, phi = a.phi;

    bool cv[DL];  // coordinate held in x[i] is a real coordinate
    int64_t cidx[DL];
#pragma unroll
    for (int i = 0; i < DL; ++i) {
        const int c = DIST ? lg + G * i : i;
        cv[i] = c < D;
        cidx[i] = (int64_t)c * N;
    }

    // ---- small helpers on group-owned vectors -----------------------------
    constexpr int LDSL = HBM_STACK ? Model::LDS_LEVELS : 0;
    double* const hyb = lds + MSH + (threadIdx.x / G) * nuts_hybrid_lds_doubles(VS, LDSL);
    // offset of `off` inside the group's LDS part of a hybrid stack, or -1
    auto lds_off = [&](int off) -> int {
        if constexpr (LDSL > 0) {
            if (off >= FIRST && off < FIRST + LDSL * 2 * VS) return off - FIRST;
            if (off >= CAND && off < CAND + LDSL * CREC) return LDSL * 2 * VS + (off - CAND);
        }
        return -1;
    };
    // (hybrid stack: typed pointers -- a plain `lo >= 0 ? hyb : slot` makes the optimiser select the POINTER and
    // emit flat_load / flat_store for both the LDS and the HBM levels)
    using ldsp = __attribute__((address_space(3))) double*;
    using glbp = __attribute__((address_space(1))) double*;
    auto vstore = [&](int off, const double (&v)[DL]) {
        const int lo = lds_off(off);
        if constexpr (DIST && HBM_STACK) {
            if (lo >= 0) {
                const ldsp h = (ldsp)hyb;
#pragma unroll
                for (int i = 0; i < DL; ++i) h[lo + i * G + lg] = v[i];
            } else {
                const glbp sl = (glbp)slot;
#pragma unroll
                for (int i = 0; i < DL; ++i) sl[(off + i * G + lg) * SK] = v[i];
            }
        } else if constexpr (DIST) {
            if (lo >= 0) {
#pragma unroll
                for (int i = 0; i < DL; ++i) hyb[lo + i * G + lg] = v[i];
            } else {
#pragma unroll
                for (int i = 0; i < DL; ++i) slot[off + i * G + lg] = v[i];
            }
        } else {
            if (lg == 0) {
#pragma unroll
                for (int i = 0; i < DL; ++i) slot[off + i] = v[i];
            }
        }
    };
    auto vload = [&](int off, double (&v)[DL]) {
        const int lo = lds_off(off);
        if constexpr (DIST && HBM_STACK) {
            if (lo >= 0) {
                const ldsp h = (ldsp)hyb;
#pragma unroll
                for (int i = 0; i < DL; ++i) v[i] = h[lo + i * G + lg];
            } else {
                const glbp sl = (glbp)slot;
#pragma unroll
                for (int i = 0; i < DL; ++i) v[i] = sl[(off + i * G + lg) * SK];
            }
        } else if (DIST && lo >= 0) {
#pragma unroll
            for (int i = 0; i < DL; ++i) v[i] = hyb[lo + i * G + lg];
        } else {
#pragma unroll
            for (int i = 0; i < DL; ++i) v[i] = DIST ? slot[off + i * G + lg] : slot[off + i];
        }
    };
    auto sstore = [&](int off, double v) {
        const int lo = lds_off(off);
        if constexpr (HBM_STACK) {
            if (lg == 0) { if (lo >= 0) ((ldsp)hyb)[lo] = v; else ((glbp)slot)[off * SK] = v; }
        } else {
            if (lg == 0) { if (lo >= 0) hyb[lo] = v; else slot[off] = v; }
        }
    };
    auto sload = [&](int off) -> double {
        const int lo = lds_off(off);
        if constexpr (HBM_STACK) {
            double v;
            if (lo >= 0) v = ((ldsp)hyb)[lo]; else v = ((glbp)slot)[off * SK];
            return v;
        } else {
            return lo >= 0 ? hyb[lo] : slot[off];
        }
    };
    auto dot = [&](const double (&u)[DL], const double (&v)[DL]) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < DL; ++i) s = fma(u[i], v[i], s);
        if constexpr (DIST) s = group_sum<G>(s);
        return s;
    };
    // nuts.py:152-160 with (xm, rm) in LDS at `off` / (x, r) current, by direction
    auto uturn = [&](int off_x, int off_r, const double (&xc)[DL], const double (&rc)[DL], int dir) {
        double xo[DL], ro[DL];
        vload(off_x, xo);
        vload(off_r, ro);
        double sa = 0.0, sb = 0.0;  // dx . r_minus, dx . r_plus
#pragma unroll
        for (int i = 0; i < DL; ++i) {
            const double dx = dir > 0 ? (xc[i] - xo[i]) : (xo[i] - xc[i]);  // xplus - xminus
            const double rmn = dir > 0 ? ro[i] : rc[i];
            const double rpl = dir > 0 ? rc[i] : ro[i];
            sa = fma(dx, rmn, sa);
            sb = fma(dx, rpl, sb);
        }
        if constexpr (DIST && G == 64) wave_sum2(sa, sb, sa, sb);      // both dot products through one butterfly
        else if constexpr (DIST) { sa = group_sum<G>(sa); sb = group_sum<G>(sb); }
        return (sa < 0.0) || (sb < 0.0);
    };

    // ---- per-group state ---------------------------------------------------
    int phase = NEED;
    int64_t p = 0;
    double x[DL], r[DL], g[DL];
    double logu = 0.0;
    int j = 0, i = 0, dir = 1, n = 1, nleap = 0;
    uint32_t q = 0, qbase = 0;
    double ub0 = 0.0, ub1 = 0.0;
    int64_t toff = 0, tlen = 0;
    bool overflow = false;
#pragma unroll
    for (int k = 0; k < DL; ++k) { x[k] = 0.0; r[k] = 0.0; g[k] = 0.0; }
    // HBM-stack models (D = 256: one wavefront per particle, 4 coordinates per lane): the two edges of the
    // trajectory and the selected sample -- touched at every doubling -- stay in REGISTERS (8 vectors = 64 VGPRs),
    // updated by selects; only the deeper tree-stack levels travel to the HBM slot.
    constexpr bool REGE = HBM_STACK && DIST && DL <= 4;   // (8 coordinates per lane would spill)
    constexpr bool REGE_K = REGE && TWO_PHASE && model_two_phase<Model>::value;   // NutsArgs::jcap / resume_in
    // (edges in the slot -- one lane per particle, 13 coordinates --: such a kernel can PARK a tree for the finisher, it
    //  never takes one up)
    constexpr bool PARK_SLOT = !REGE && TWO_PHASE && model_two_phase<Model>::value;
    constexpr bool WIDE = REGE && G == 64;                // the wavefront sees the whole particle: statistics in-kernel
    double x0[RL_X0(WIDE, DL)];
    constexpr int RL = REGE ? DL : 1;
    double emx[RL], emr[RL], emg[RL], epx[RL], epr[RL], epg[RL], slx[RL], slr[RL], slp0 = 0.0, slp1 = 0.0;
#pragma unroll
    for (int k = 0; k < RL; ++k) { emx[k] = emr[k] = emg[k] = epx[k] = epr[k] = epg[k] = slx[k] = slr[k] = 0.0; }

    // (what only a refill, a tape read or a tree boundary needs -- seed, iteration, tape pointers, depth caps -- is re-read
    //  from the kernel-argument segment there: twelve scalar registers less across the leaf loop, whose scalar spills to
    //  vector lanes cost a v_readlane / v_writelane each)
    const bool taped = a.tape != nullptr;
    auto refill = [&]() {
        const auto ka = kargs();
        const uint64_t seed = ka->seed;
        const u32x4 o = philox4x32_10({(qbase >> 1) + (uint32_t)lg, (uint32_t)(ka->particle_base + p), ka->iter,
                                       kStreamNuts}, (uint32_t)seed, (uint32_t)(seed >> 32));
        ub0 = u53(o.a, o.b);
        ub1 = u53(o.c, o.d);
    };
    auto draw = [&]() -> double {
        double v;
        if (taped) {
            if ((int64_t)q < tlen) v = kargs()->tape[toff + q];
            else { v = 0.5; overflow = true; }
        } else {
            if (q >= qbase + 2u * G) { qbase += 2u * G; refill(); }
            const int src = (int)((q - qbase) >> 1);
            const double v0 = group_read<G>(ub0, src), v1 = group_read<G>(ub1, src);
            v = (q & 1u) ? v1 : v0;
        }
        ++q;
        return v;
    };

    PROF_DECL;
    for (unsigned int it = 0u;; ++it) {
        PROF(7);
        if constexpr (G == 64) {
            // one particle per wavefront: the tree's control state is the same in every lane, but the compiler cannot
            // know that (it descends from an atomic and from loads).  Reading it from the first lane makes it scalar:
            // branches on it become scalar branches, its arithmetic moves to the scalar unit.
            phase = __builtin_amdgcn_readfirstlane(phase);
            i = __builtin_amdgcn_readfirstlane(i);
            j = __builtin_amdgcn_readfirstlane(j);
            n = __builtin_amdgcn_readfirstlane(n);
            dir = __builtin_amdgcn_readfirstlane(dir);
            nleap = __builtin_amdgcn_readfirstlane(nleap);
            q = (uint32_t)__builtin_amdgcn_readfirstlane((int)q);
            qbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)qbase);
        }
        // ---- fetch work -----------------------------------------------------
        if (phase == NEED && (G == 64 || (it & (unsigned int)(a.step_align - 1)) == 0u)) {
            unsigned int t = 0;
            bool none;
            bool resumed = false;
            if (REGE_K && a.resume_in) {
                // a parked tree: its state as the first launch left it at the doubling boundary, then the next doubling
                if (lg == 0) t = atomicAdd(a.queue, 1u);
                t = (unsigned int)group_read_i<G>((int)t, 0);
                none = t >= a.pend[0];
                resumed = !none;
                if (resumed) t = a.pend[1 + t];
            } else if constexpr (HBM_STACK && DIST && G == 64) {
                // One wavefront per particle and the [D][N] layout: a wavefront touches 8 bytes of every 64-byte line
                // of its particle's coordinates, and the other 56 belong to the 7 neighbouring particles.  Lines
                // (8 particles) are dealt to the XCDs -- blocks go round-robin over the XCDs, so blockIdx % 8 names the
                // XCD, each with a queue of its own -- and consecutive grabs of an XCD take the particles of one line:
                // its L2 then fetches / writes back every line once instead of up to 8 times.
                const unsigned int nq = gridDim.x < 8u ? gridDim.x : 8u;
                const unsigned int xq = blockIdx.x % nq;
                int64_t pp = -1;
                for (;;) {
                    t = 0;
                    if (lg == 0) t = atomicAdd(a.queue + 8 + xq, 1u);
                    t = (unsigned int)__builtin_amdgcn_readfirstlane((int)t);   // (G == 64: lane 0 of the wavefront)
                    const int64_t line = (int64_t)(t >> 3) * nq + xq;
                    if (line * 8 >= N) break;
                    if (line * 8 + (t & 7u) < N) { pp = line * 8 + (t & 7u); break; }
                }
                none = pp < 0;
                t = none ? 0u : (unsigned int)pp;
            } else {
                if (lg == 0) t = atomicAdd(a.queue, 1u);
                t = (unsigned int)group_read_i<G>((int)t, 0);
                none = (int64_t)t >= N;
                if constexpr (REGE_K && G < 64) {
                    if (none) {
                        // no fresh particle left: a tree this launch has parked at its inner level, the oldest first.  (A
                        // group that finds none leaves: whoever parks a tree looks here right afterwards, so every entry is
                        // taken -- by its own group at the latest.)
                        unsigned int* const mq = kargs()->mq;
                        if (mq) {
                            unsigned int got = 0u;
                            if (lg == 0) {
                                // mq[5] counts the entries written and not yet claimed: a claim is ONE fetch-add whatever the
                                // crowd (a compare-and-swap loop on the take counter, tried first, made 16 000 groups arriving
                                // together retry each other for seconds); the claimant then draws its index, which names an
                                // entry already allocated -- and, at worst, written a few instructions later
                                int* const avail = reinterpret_cast<int*>(mq + 5);
                                for (;;) {
                                    if (atomicSub(avail, 1) > 0) {
                                        const unsigned int tk = atomicAdd(mq + 1, 1u);
                                        // (the wait is bounded so that the launch ends whatever happens: an entry never seen
                                        //  is counted in mq[2], which the host checks -- smcn_propose_nuts fails then)
                                        unsigned int spins = 0u;
                                        while ((got = __hip_atomic_load(mq + 16 + tk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u) {
                                            __builtin_amdgcn_s_sleep(1);
                                            if (++spins > 100000u) { atomicAdd(mq + 2, 1u); break; }
                                        }
                                        break;
                                    }
                                    // nothing for this group -- unless the count was only held down by other groups' failed
                                    // claims: whoever gives its claim back LAST sees the true count, and tries again
                                    if (atomicAdd(avail, 1) + 1 <= 0) break;
                                }
                            }
                            got = (unsigned int)group_read_i<G>((int)got, 0);
                            if (got != 0u) {
                                none = false;
                                resumed = true;
                                t = got - 1u;
                            }
                        }
                    }
                }
            }
            if (none) {
                phase = DONE;
            } else if (resumed) {
                if constexpr (REGE_K) {
                    p = (int64_t)t;
                    const auto ka = kargs();
                    // (the record offsets are loop invariants the optimiser would compute once, in front of the leaf loop,
                    //  and keep -- in scratch: an opaque copy of D keeps them here, where a parked tree is taken up)
                    int Dr = D;
                    asm volatile("" : "+v"(Dr));
                    // (first launch: a record of its inner level, written through by a group that may sit on another XCD)
                    const bool thru = !ka->resume_in;
                    const double* const rec = thru ? ka->mq_rec + p * 128 : ka->resume + p * (8 * (int64_t)Dr + 8);
                    auto ld = [&](const double* q_) __attribute__((always_inline)) -> double {
                        return thru ? __hip_atomic_load(q_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *q_;
                    };
#pragma unroll
                    for (int k = 0; k < DL; ++k) {
                        const int c = lg + G * k;
                        emx[k] = cv[k] ? ld(rec + c) : 0.0;          emr[k] = cv[k] ? ld(rec + Dr + c) : 0.0;     emg[k] = cv[k] ? ld(rec + 2 * Dr + c) : 0.0;
                        epx[k] = cv[k] ? ld(rec + 3 * Dr + c) : 0.0; epr[k] = cv[k] ? ld(rec + 4 * Dr + c) : 0.0; epg[k] = cv[k] ? ld(rec + 5 * Dr + c) : 0.0;
                        slx[k] = cv[k] ? ld(rec + 6 * Dr + c) : 0.0; slr[k] = cv[k] ? ld(rec + 7 * Dr + c) : 0.0;
                    }
                    const double* const sc = rec + 8 * Dr;
                    slp0 = ld(sc); slp1 = ld(sc + 1); logu = ld(sc + 2);
                    n = (int)ld(sc + 3); j = (int)ld(sc + 4); nleap = (int)ld(sc + 5); q = (uint32_t)ld(sc + 6); overflow = ld(sc + 7) != 0.0;
                    if constexpr (WIDE) {
                        const double* const xin = ka->x;
#pragma unroll
                        for (int k = 0; k < DL; ++k) x0[k] = cv[k] ? xin[cidx[k] + p] : 0.0;
                    }
                    qbase = q - (q % (2u * G));
                    if (taped) { const int64_t* const to = kargs()->tape_off; toff = to[p]; tlen = to[p + 1] - toff; }
                    else refill();
                    dir = (draw() < 0.5) ? 1 : -1;  // :91
#pragma unroll
                    for (int k = 0; k < DL; ++k) {
                        x[k] = dir > 0 ? epx[k] : emx[k]; r[k] = dir > 0 ? epr[k] : emr[k]; g[k] = dir > 0 ? epg[k] : emg[k];
                    }
                    i = 0;
                    phase = LEAF;
                }
            } else {
                p = (int64_t)t;
                const auto ka = kargs();
                const double* const xin = ka->x;
                const double* const rin = ka->r;
#pragma unroll
                for (int k = 0; k < DL; ++k) {
                    x[k] = cv[k] ? xin[cidx[k] + p] : 0.0;
                    r[k] = cv[k] ? rin[cidx[k] + p] : 0.0;
                }
                q = 0; qbase = 0; overflow = false; nleap = 0;
                if (taped) { const int64_t* const to = kargs()->tape_off; toff = to[p]; tlen = to[p + 1] - toff; }
                else refill();
                phase = INIT;
            }
        }
        if (__ballot(phase != DONE) == 0ull) break;
        PROF(0);

        // ---- leapfrog, first half (nuts.py:169-170) --------------------------
        const double e = dir * eps, h = dir * eps / 2;
        if (phase == LEAF) {
#pragma unroll
            for (int k = 0; k < DL; ++k) r[k] = r[k] + h * g[k];
#pragma unroll
            for (int k = 0; k < DL; ++k) x[k] = x[k] + e * r[k];
        }
        // ---- target value + gradient (nuts.py:66,72,122,171): all lanes ------
        double lpri, llik, gp[DL], gl[DL];
        PROF(1);
        // One wavefront per particle and a model whose value is a sum of per-lane shares: a LEAF puts the shares and
        // |r'|^2 of the second half kick (nuts.py:173) through ONE four-value butterfly instead of three (the gradient
        // needs no reduction, so r' is known before the value is).  A non-finite density -- the rare case in which
        // the reference's adapter overrides the gradient (bridgestan.py:79-80) -- redoes the leaf the plain way.
        constexpr bool FUSED_LEAF = G == 64 && DIST && model_has_partial<Model>::value;
        bool kicked = false;
        double kin_leaf = 0.0;
        if constexpr (FUSED_LEAF) {
            if (phase == LEAF) {
                double ssp, slp;
                model.eval_partial(x, ssp, slp, gp, gl);
                double rk[DL], kp = 0.0;
#pragma unroll
                for (int k = 0; k < DL; ++k) {
                    const double gk = cv[k] ? fma(phi, gl[k], gp[k]) : 0.0;
                    rk[k] = r[k] + h * gk;
                    kp = fma(rk[k], rk[k], kp);
                }
                double sst, slt, kt, unused;
                wave_sum4(ssp, slp, kp, 0.0, sst, slt, kt, unused);
                model.finish(sst, slt, lpri, llik);
                if (finite_d(lpri + phi * llik)) {           // (wave-uniform)
#pragma unroll
                    for (int k = 0; k < DL; ++k) r[k] = rk[k];
                    kin_leaf = kt;
                    kicked = true;
                }
            }
        }
        if (!kicked) model.eval(x, lpri, llik, gp, gl);
        PROF(2);
        double lp = lpri + phi * llik;
        const bool bad = !finite_d(lp);  // bridgestan.py:47-49,79-80
        lp = bad ? -kInf : lp;
#pragma unroll
        for (int k = 0; k < DL; ++k) g[k] = cv[k] ? (bad ? -kInf : fma(phi, gl[k], gp[k])) : 0.0;

        if (phase == INIT) {
            // nuts.py:66-87
            const double kin_start = dot(r, r);
            const double H0 = lp - 0.5 * kin_start;
            if constexpr (WIDE) {
#pragma unroll
                for (int k = 0; k < DL; ++k) x0[k] = x[k];
                if (lg == 0) { const auto kw = kargs(); if (kw->kin0) kw->kin0[p] = kin_start; }
            }
            double ex = draw();
            if (!taped) ex = -log1p(-ex);
            logu = H0 - ex;
            if constexpr (REGE) {
#pragma unroll
                for (int k = 0; k < DL; ++k) {
                    emx[k] = epx[k] = slx[k] = x[k];
                    emr[k] = epr[k] = slr[k] = r[k];
                    emg[k] = epg[k] = g[k];
                }
                slp0 = lpri; slp1 = llik;
            } else {
                vstore(EM, x); vstore(EM + VS, r); vstore(EM + 2 * VS, g);
                vstore(EP, x); vstore(EP + VS, r); vstore(EP + 2 * VS, g);
                vstore(SEL, x); vstore(SEL + VS, r);
                sstore(SELP, lpri); sstore(SELP + 1, llik);
            }
            if (lg == 0) { const auto ka = kargs(); ka->lpri0[p] = lpri; ka->llik0[p] = llik; }
            j = 0; n = 1; i = 0;
            dir = (draw() < 0.5) ? 1 : -1;  // nuts.py:91
            phase = LEAF;
        } else if (phase == LEAF) {
            // ---- leapfrog, second half (nuts.py:173) and leaf tests (:123-125)
            if (!kicked) {
#pragma unroll
                for (int k = 0; k < DL; ++k) r[k] = r[k] + h * g[k];
            }
            ++nleap;
            const double joint = lp - 0.5 * (kicked ? kin_leaf : dot(r, r));
            int nsub = (logu < joint) ? 1 : 0;
            bool ssub = (logu - a.delta_max) >= joint;
            if constexpr (G == 64) {   // (wave-uniform, see the loop top)
                nsub = __builtin_amdgcn_readfirstlane(nsub);
                ssub = __builtin_amdgcn_readfirstlane((int)ssub) != 0;
            }
            double cx[DL], cr[DL], clp = lpri, cll = llik;
#pragma unroll
            for (int k = 0; k < DL; ++k) { cx[k] = x[k]; cr[k] = r[k]; }
            if (j > 0 && (i & 1) == 0) {
                const int s = (i == 0) ? j : (__ffs(i) - 1);  // slot = ctz(i), or j for the first leaf
                vstore(FIRST + (s - 1) * 2 * VS, x);
                vstore(FIRST + (s - 1) * 2 * VS + VS, r);
            }
            // ---- merge completed sub-trees (nuts.py:134-148) --------------------
            PROF(3);
            bool done = false;
            int m = 0;
            for (;;) {
                if (m == j) { done = true; break; }
                if (ssub) {
                    // the stop unwinds the recursion: each ancestor for which the stopped
                    // sub-tree is the SECOND half still consumes its merge uniform
                    q += (uint32_t)__popc((unsigned)(i >> m) & ((1u << (j - m)) - 1u));
                    done = true;
                    break;
                }
                const int crec = CAND + m * CREC;  // pending first half of level m+1
                if (((i >> m) & 1) == 0) {
                    vstore(CAND + m * CREC, cx);
                    vstore(CAND + m * CREC + VS, cr);
                    sstore(crec + 2 * VS, clp); sstore(crec + 2 * VS + 1, cll); sstore(crec + 2 * VS + 2, (double)nsub);
                    break;
                }
                if constexpr (G == 1 && HBM_STACK) {
                    // One lane per particle: the stack levels above the LDS one live in HBM, and nothing hides a round trip
                    // at one wavefront per SIMD.  Everything this merge may need -- the pending candidate, its three scalars,
                    // the sub-tree's first leaf -- is asked for at once (one trip per level instead of three dependent ones);
                    // the candidate is loaded whether or not the draw keeps it.
                    const int i0 = (i >> (m + 1)) << (m + 1);
                    const int s = (i0 == 0) ? j : (__ffs(i0) - 1);
                    double tx[DL], tr[DL], fx[DL], fr[DL];
                    vload(CAND + m * CREC, tx);
                    vload(CAND + m * CREC + VS, tr);
                    vload(FIRST + (s - 1) * 2 * VS, fx);
                    vload(FIRST + (s - 1) * 2 * VS + VS, fr);
                    const double n1d = sload(crec + 2 * VS + 2), tlp = sload(crec + 2 * VS), tll = sload(crec + 2 * VS + 1);
                    const double u = draw();  // nuts.py:142, always
                    const int n1 = (int)n1d;
                    const int den = (n1 + nsub) > 1 ? (n1 + nsub) : 1;
                    if (!(u < (double)nsub / (double)den)) {
#pragma unroll
                        for (int k = 0; k < DL; ++k) { cx[k] = tx[k]; cr[k] = tr[k]; }
                        clp = tlp; cll = tll;
                    }
                    nsub += n1;  // :146
                    double sa = 0.0, sb = 0.0;  // nuts.py:152-160
#pragma unroll
                    for (int k = 0; k < DL; ++k) {
                        const double dx = dir > 0 ? (x[k] - fx[k]) : (fx[k] - x[k]);
                        const double rmn = dir > 0 ? fr[k] : r[k];
                        const double rpl = dir > 0 ? r[k] : fr[k];
                        sa = fma(dx, rmn, sa);
                        sb = fma(dx, rpl, sb);
                    }
                    ssub = (sa < 0.0) || (sb < 0.0);  // :148
                    ++m;
                    continue;
                }
                const double u = draw();  // nuts.py:142, always
                int n1 = (int)sload(crec + 2 * VS + 2);
                if constexpr (G == 64) n1 = __builtin_amdgcn_readfirstlane(n1);
                const int den = (n1 + nsub) > 1 ? (n1 + nsub) : 1;
                if (!(u < (double)nsub / (double)den)) {
                    vload(CAND + m * CREC, cx);
                    vload(CAND + m * CREC + VS, cr);
                    clp = sload(crec + 2 * VS); cll = sload(crec + 2 * VS + 1);
                }
                nsub += n1;  // :146
                const int i0 = (i >> (m + 1)) << (m + 1);
                const int s = (i0 == 0) ? j : (__ffs(i0) - 1);
                ssub = uturn(FIRST + (s - 1) * 2 * VS, FIRST + (s - 1) * 2 * VS + VS, x, r, dir);  // :148
                if constexpr (G == 64) ssub = __builtin_amdgcn_readfirstlane((int)ssub) != 0;
                ++m;
            }
            PROF(4);
            if (!done) {
                ++i;
            } else {
                // ---- end of this doubling (nuts.py:93-110) ------------------------
                if (!ssub) {  // :99 short-circuit: no draw after a stop
                    const double u = draw();
                    double ratio = (double)nsub / (double)n;
                    ratio = ratio > 1.0 ? 1.0 : ratio;
                    if constexpr (REGE) {
                        if (u < ratio) {     // (moves under the branch, not selects: smcn_device.hpp)
#pragma unroll
                            for (int k = 0; k < DL; ++k) { mov64_under_branch(slx[k], cx[k]); mov64_under_branch(slr[k], cr[k]); }
                            mov64_under_branch(slp0, clp); mov64_under_branch(slp1, cll);
                        }
                    } else if (u < ratio) {
                        vstore(SEL, cx); vstore(SEL + VS, cr);
                        sstore(SELP, clp); sstore(SELP + 1, cll);
                    }
                }
                n += nsub;  // :103
                const int eo = (dir > 0) ? EP : EM;   // the edge that moved
                const int oo = (dir > 0) ? EM : EP;   // the opposite edge
                bool stop;
                if constexpr (REGE) {
                    const bool fw = dir > 0;
                    double sa = 0.0, sb = 0.0;        // (x+ - x-) . r-, (x+ - x-) . r+   (nuts.py:152-160)
                    if (fw) {
#pragma unroll
                        for (int k = 0; k < DL; ++k) { mov64_under_branch(epx[k], x[k]); mov64_under_branch(epr[k], r[k]); mov64_under_branch(epg[k], g[k]); }
                    } else {
#pragma unroll
                        for (int k = 0; k < DL; ++k) { mov64_under_branch(emx[k], x[k]); mov64_under_branch(emr[k], r[k]); mov64_under_branch(emg[k], g[k]); }
                    }
#pragma unroll
                    for (int k = 0; k < DL; ++k) {
                        const double dx = epx[k] - emx[k];
                        sa = fma(dx, emr[k], sa);
                        sb = fma(dx, epr[k], sb);
                    }
                    if constexpr (G == 64) wave_sum2(sa, sb, sa, sb);
                    else { sa = group_sum<G>(sa); sb = group_sum<G>(sb); }
                    stop = ssub || (sa < 0.0) || (sb < 0.0);  // :105
                } else {
                    vstore(eo, x); vstore(eo + VS, r); vstore(eo + 2 * VS, g);
                    stop = ssub || uturn(oo, oo + VS, x, r, dir);  // :105
                }
                ++j;
                if (stop || j > kargs()->max_depth) {  // :89,109
                    double xs[DL], rs[DL];
                    if constexpr (REGE) {
#pragma unroll
                        for (int k = 0; k < DL; ++k) { xs[k] = slx[k]; rs[k] = slr[k]; }
                    } else {
                        vload(SEL, xs); vload(SEL + VS, rs);
                    }
                    const auto ka = kargs();
                    if (DIST || lg == 0) {
                        double* const xo = ka->x_new;
                        double* const ro = ka->r_new;
#pragma unroll
                        for (int k = 0; k < DL; ++k) {
                            if (cv[k]) { xo[cidx[k] + p] = xs[k]; ro[cidx[k] + p] = rs[k]; }
                        }
                    }
                    if constexpr (WIDE) {
                        const double kin_end = dot(rs, rs);
                        bool all_moved = true;
#pragma unroll
                        for (int k = 0; k < DL; ++k) all_moved = all_moved && (!cv[k] || xs[k] != x0[k]);
                        const bool every = __ballot(!all_moved) == 0ull;
                        if (lg == 0 && ka->kin1) { ka->kin1[p] = kin_end; ka->moved[p] = every ? 1 : 0; }
                    }
                    if (lg == 0) {
                        ka->lpri1[p] = REGE ? slp0 : slot[SELP * SK]; ka->llik1[p] = REGE ? slp1 : slot[(SELP + 1) * SK];
                        ka->nleap[p] = nleap; ka->depth[p] = j; ka->ndraws[p] = (int32_t)q;
                        ka->flags[p] = overflow ? 1 : 0;
                    }
                    phase = NEED;
                } else if (REGE_K && kargs()->jcap > 0 && (j == kargs()->jcap || (kargs()->mq && j == kargs()->mq_b))) {
                    if constexpr (REGE_K) {       // park: the second launch (or, inner level, a group of this one) takes the tree from here
                        const auto ka = kargs();
                        int Dr = D;                       // (opaque: see where a parked tree is taken up)
                        asm volatile("" : "+v"(Dr));
                        const bool inner = j != ka->jcap;
                        double* const rec = inner ? ka->mq_rec + p * 128 : ka->resume + p * (8 * (int64_t)Dr + 8);
                        auto st = [&](double* q_, double v_) __attribute__((always_inline)) {
                            if (inner) __hip_atomic_store(q_, v_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            else *q_ = v_;
                        };
#pragma unroll
                        for (int k = 0; k < DL; ++k) {
                            const int c = lg + G * k;
                            if (cv[k]) {
                                st(rec + c, emx[k]);          st(rec + Dr + c, emr[k]);     st(rec + 2 * Dr + c, emg[k]);
                                st(rec + 3 * Dr + c, epx[k]); st(rec + 4 * Dr + c, epr[k]); st(rec + 5 * Dr + c, epg[k]);
                                st(rec + 6 * Dr + c, slx[k]); st(rec + 7 * Dr + c, slr[k]);
                            }
                        }
                        if (lg == 0) {
                            double* const sc = rec + 8 * Dr;
                            st(sc, slp0); st(sc + 1, slp1); st(sc + 2, logu);
                            st(sc + 3, (double)n); st(sc + 4, (double)j); st(sc + 5, (double)nleap); st(sc + 6, (double)q); st(sc + 7, overflow ? 1.0 : 0.0);
                        }
                        if (!inner) {
                            if (lg == 0) {
                                const unsigned int at = atomicAdd(ka->pend, 1u);
                                ka->pend[1 + at] = (unsigned int)p;
                            }
                        } else {
                            // inner level: the record has left this wavefront (and, written through, its XCD) before the entry does
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            if (lg == 0) {
                                const unsigned int at = atomicAdd(ka->mq, 1u);
                                __hip_atomic_store(ka->mq + 16 + at, (unsigned int)p + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                atomicAdd(reinterpret_cast<int*>(ka->mq + 5), 1);
                            }
                        }
                    }
                    phase = NEED;
                } else if (PARK_SLOT && kargs()->jcap > 0 && j == kargs()->jcap) {
                    if constexpr (PARK_SLOT) {    // park from the slot: the same record as above
                        const auto ka = kargs();
                        int Dr = D;
                        asm volatile("" : "+v"(Dr));
                        double* const rec = ka->resume + p * (8 * (int64_t)Dr + 8);
                        constexpr int src[8] = {EM, EM + VS, EM + 2 * VS, EP, EP + VS, EP + 2 * VS, SEL, SEL + VS};
#pragma unroll
                        for (int v8 = 0; v8 < 8; ++v8) {
                            double t[DL];
                            vload(src[v8], t);
#pragma unroll
                            for (int k = 0; k < DL; ++k) {
                                const int c = lg + G * k;
                                if (cv[k]) rec[v8 * Dr + c] = t[k];
                            }
                        }
                        const double s0 = sload(SELP), s1 = sload(SELP + 1);
                        if (lg == 0) {
                            double* const sc = rec + 8 * Dr;
                            sc[0] = s0; sc[1] = s1; sc[2] = logu;
                            sc[3] = (double)n; sc[4] = (double)j; sc[5] = (double)nleap; sc[6] = (double)q; sc[7] = overflow ? 1.0 : 0.0;
                            const unsigned int at = atomicAdd(ka->pend, 1u);
                            ka->pend[1 + at] = (unsigned int)p;
                        }
                    }
                    phase = NEED;
                } else {
                    dir = (draw() < 0.5) ? 1 : -1;  // :91
                    const int so = (dir > 0) ? EP : EM;
                    if constexpr (REGE) {
                        if (dir > 0) {
#pragma unroll
                            for (int k = 0; k < DL; ++k) { mov64_under_branch(x[k], epx[k]); mov64_under_branch(r[k], epr[k]); mov64_under_branch(g[k], epg[k]); }
                        } else {
#pragma unroll
                            for (int k = 0; k < DL; ++k) { mov64_under_branch(x[k], emx[k]); mov64_under_branch(r[k], emr[k]); mov64_under_branch(g[k], emg[k]); }
                        }
                    } else {
                        vload(so, x); vload(so + VS, r); vload(so + 2 * VS, g);
                    }
                    i = 0;
                }
            }
            PROF(5);
        }
    }
    PROF_FLUSH(a);
}

}  // namespace smcn
