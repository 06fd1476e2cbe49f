// C ABI of libsmcnuts_hip.so (see include/smcnuts_hip.h for the contract and
// the reference interfaces each entry point replaces).
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/smcnuts_hip.h"
#include "smcn_nuts.hpp"
#include "smcn_nuts_wave.hpp"
#include "smcn_nuts_fin.hpp"
#include "smcn_nuts2.hpp"
#include "smcn_nuts3.hpp"
#ifndef SMCN_WPE2_LC
#define SMCN_WPE2_LC 1
#define SMCN_WPE2_LF 0
#endif
#include "smcn_temper.hpp"
#ifdef SMCN_VARIANTS
#include "smcn_models_variants.hpp"
#include "smcn_nuts_lane.hpp"
#include "smcn_nuts2_kernel.hpp"
#endif
#include "smcn_nuts_host.hpp"
#include "smcn_weights.hpp"
#include "smcn_glk.hpp"
#include "smcn_step.hpp"
#include "smcn_comm.hpp"

using namespace smcn;

namespace {
// roctx ranges around the phases of the SMC loop (rocprofv3 --marker-trace attributes kernels and HBM bytes
// to them).  The marker library is looked up at run time, so the product has no link-time dependency on it.
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_LAZY | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so", RTLD_LAZY | RTLD_GLOBAL);
        if (h) {
            push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
            pop = (int (*)())dlsym(h, "roctxRangePop");
        }
    }
};
inline Roctx& roctx() { static Roctx r; return r; }
struct Range {
    explicit Range(const char* name) { if (roctx().push && roctx().pop) roctx().push(name); }
    ~Range() { if (roctx().push && roctx().pop) roctx().pop(); }
    Range(const Range&) = delete;
};
thread_local std::string g_create_error;
constexpr int kMaxPart = 1024;    // block partials per reduction
constexpr int kTimerRing = 512;   // NUTS launches timed between two smcn_timers calls
constexpr int64_t kDlChunk = 8;   // generations per staging pass of the history download
}  // namespace

struct smcn_ctx {
    int device = 0;
    int64_t N = 0, base = 0;
    int model = 0, D = 0, Dc = 0;
    int num_cu = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    uint64_t seed = 0;
    std::string err;
    std::vector<double> mdata_h;

    // device state, all fp64; vectors are [D][N]
    double *mdata = nullptr, *x = nullptr, *x_new = nullptr, *x_tmp = nullptr, *r = nullptr, *r_new = nullptr;
    double *logw = nullptr, *logw_new = nullptr, *wn = nullptr, *work = nullptr;
    double *lpri0 = nullptr, *llik0 = nullptr, *lpri1 = nullptr, *llik1 = nullptr, *Lg = nullptr, *qv = nullptr;
    double *scan_local = nullptr, *ttot = nullptr, *toff = nullptr, *part = nullptr, *scal = nullptr;
    double* glk_buf = nullptr;          // device-side Gaussian L-kernel: mean, both moment sums, parameters (lazy)
    double* glk_xchg = nullptr;         // ... over shards: [local row | world gathered rows] of moment sums
    int glk_world = 0;
    // two-phase NUTS launches (smcn_set_nuts_cap): doublings of the first launch, records and list of the parked trees
    int nuts_jcap = 0, nuts_wide2 = 1;
    double* nuts_resume = nullptr;
    unsigned int* nuts_pend = nullptr;
    unsigned int* nuts_mq = nullptr;    // list of the trees parked at the inner level of a launch (smcn_set_nuts_requeue)
    double* nuts_mq_rec = nullptr;      // ... and their records ([N][128])
    int nuts_mq_b = 0;                  // ... after this many doublings (0: no inner level)
    bool nuts_mq_used = false;          // the last proposal had an inner level (its hand-over count is checked behind the wait)
    bool nuts_mq_ok = false;            // set by the entry points that wait for the proposal and check that count (smcn_propose_nuts)
    int64_t nuts_parked = 0;            // trees the last launch parked
    double* stage = nullptr;  // [N*D] host<->device staging, also [M*D] for target_eval
    int64_t stage_len = 0;
    double* stage2 = nullptr;
    int64_t stage2_len = 0;
    int32_t *nleap = nullptr, *depth = nullptr, *ndraws = nullptr, *flags = nullptr;
    int64_t* idx = nullptr;
    unsigned int* queue = nullptr;
    double* kin0 = nullptr;     // wave-per-particle NUTS kernels: |r|^2, |r'|^2 and the all-coordinates-moved flag per particle
    double* kin1 = nullptr;
    int32_t* moved_i = nullptr;
    bool kin_valid = false;     // written by the last NUTS launch (consumed by the device-resident re-weighting)
    // Momentum layout.  Targets whose particle fills a wavefront (nuts_wave_kernel: coordinate c on lane c % 64) read and
    // write a particle's momentum as ONE contiguous row: r / r_new then hold [N][D] ("particle-major": 512-byte coalesced
    // rows) instead of the [D][N] every other kernel uses -- converted in place (momentum_dn) before any of those reads them
    bool r_pm = false, r_new_pm = false;
    unsigned long long* prof = nullptr;
    double* tape_d = nullptr;
    int64_t* tape_off_d = nullptr;
    int64_t tape_cap = 0;
    bool momentum_set = false, lg_set = false, q_set = false, u_set = false;
    // device-resident loop (smcn_fast_*)
    double *hist = nullptr, *ss = nullptr, *lp = nullptr, *gath = nullptr, *hist_x = nullptr, *hist_logw = nullptr;
    double* u_res = nullptr;
    double *in_rec = nullptr, *out_rec = nullptr;   // nuts2 per-particle records
    double* nuts_scratch = nullptr;                  // HBM tree stacks (large D)
    int64_t nuts_scratch_len = 0;
    int64_t fast_K = -1;
    bool fast_hist = false;
    // fused transitions (smcn_super_*)
    int fuse_max = 0;
    int64_t rec_cap = 0;          // transitions the record buffers hold
    int resample_scheme = 0;   // 0 multinomial (reference), 1 systematic
    bool fused_ok = false;     // the model's NUTS kernel takes B > 1 transitions per launch
    bool lane_kernel = false;  // NUTS by nuts3_kernel (one lane per particle)
    int64_t arma_T = 0;        // series length of an arma context
    double *tb_state = nullptr, *tb_part = nullptr, *tb_local = nullptr, *tb_gath = nullptr;   // device-side ESS bisection
    int tb_world = 0, tb_blocks = 0;
    int wide_eval = 1;         // nuts3_kernel: lane groups evaluate a wavefront's last stragglers (smcn_set_wide_eval)
    int64_t lane_grid_cap = 0; // nuts3_kernel: wavefronts launched at most (0: one per SIMD; < 0: no cap, a wavefront per 64 particles)
    int lane_segments = 0;     // nuts3_kernel, fewer lanes than particles: segments a block is worked off in (0: auto, 1: whole blocks)
    unsigned long long* handover = nullptr;  // [N][segments - 1][VH + 1 pairs] (lane queue: x', running log-weight between segments)
    int64_t handover_len = 0;
    bool plain_block = false;  // the last smcn_fuse_run ran ONE transition of a model without fused transitions
    // in-library shard exchange (RCCL) and the routed global resampling (smcn_gres_*)
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_world = 1;
    double *g_ttot_all = nullptr, *g_toff_all = nullptr, *g_keys = nullptr, *g_keys_send = nullptr, *g_keys_recv = nullptr,
           *g_rows_send = nullptr, *g_rows_recv = nullptr;
    int32_t *g_dest = nullptr, *g_order = nullptr;
    int64_t g_serve_cap = 0;
    int g_world = 0;
    smcn_host_target_fn host_fn = nullptr;   // SMCN_MODEL_HOST: the caller's density
    void* host_user = nullptr;
    double *hc_vec = nullptr, *hc_sc = nullptr, *hc_gp = nullptr, *hc_gl = nullptr;   // host-target NUTS state
    int32_t* hc_st = nullptr;
    std::vector<double> hx, hlp, hll, hgp, hgl;   // host staging of the callback
    double* n2_ovf = nullptr;           // v2 kernel: overflow tree-stack levels
    int64_t n2_ovf_len = 0;
    double* ss_scratch = nullptr;       // pipelined blocks: step scalars of the inner generations
    double* rows_h = nullptr;           // pinned: history rows of the block being validated
    double* hist_h = nullptr;           // pinned: the whole scalar history, downloaded behind a run's last block
    bool hist_h_valid = false;
    hipEvent_t ev_rows = nullptr;
    hipStream_t dl_stream = nullptr;    // history download beside the loop (smcn_history_download)
    double* dl_stage = nullptr;
    int64_t dl_stage_len = 0;
    double *lpB = nullptr, *gathB = nullptr, *gen_x = nullptr, *gen_logw = nullptr, *cnt = nullptr, *shiftB = nullptr;

    // NUTS kernel timing (HIP events on the launch stream)
    hipEvent_t ev0[kTimerRing], ev1[kTimerRing];
    int ev_n = 0;
    double nuts_ms = 0.0;
    int64_t nuts_launches = 0;
};

#define CHECK_CTX(c)             \
    if (!(c)) return -1;         \
    (void)hipSetDevice((c)->device)
#define HIPC(c, call)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            (c)->err = std::string(#call) + ": " + hipGetErrorString(e_);                     \
            return -2;                                                                        \
        }                                                                                     \
    } while (0)
// Waiting for the stream: a blocking hipStreamSynchronize wakes the host ~20-30 us after the last kernel has ended (interrupt,
// scheduler), and a step-by-step iteration waits a dozen times for kernels of a few microseconds.  So: poll the stream for a
// short while first (a query is ~1 us), then block.  SMCN_SPIN_US (default 120) bounds the polling, 0 switches it off.
static inline hipError_t stream_wait(hipStream_t s) {
    static const long spin_us = [] { const char* e = getenv("SMCN_SPIN_US"); return e ? atol(e) : 120L; }();
    if (spin_us > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            const hipError_t q = hipStreamQuery(s);
            if (q != hipErrorNotReady) return q;
            if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() >= spin_us) break;
        }
    }
    return hipStreamSynchronize(s);
}

#define FAIL(c, msg)      \
    do {                  \
        (c)->err = (msg); \
        return -3;        \
    } while (0)

static int dl_prepare(smcn_ctx* c);
static int grid_for(int64_t n, int block) { return (int)((n + block - 1) / block); }
static int final_grid(int nv) { return nv < 1 ? 1 : (nv < 1024 ? nv : 1024); }   // sum_final_kernel: a block per vector
static int red_grid(int64_t n) {
    int g = grid_for(n, kRedBlock);
    return g < 1 ? 1 : (g > kMaxPart ? kMaxPart : g);
}

// ---- device memory ---------------------------------------------------------------------------------------------------
// A context's buffers are not given back to the driver when it goes: fresh memory costs its first touch, hipFree waits
// for the device and releases the memory behind the call.  Freed buffers go to a per-device cache and are handed out
// again, zeroed, to a request of the same size (every buffer of a sampler has a size that depends on N, D, K only); the
// cache holds at most 1 GiB per device -- about one sampler's working set at the headline size -- (SMCN_DEVICE_CACHE_MB),
// is emptied when an allocation fails, and by smcn_device_cache_trim.  Like hipFree, a cached free waits for the device
// first, unless the caller has already waited for every stream that used the buffer (`synced`: a context's teardown waits
// ONCE for its streams instead of once per buffer -- with several shards in one process a device-wide wait per buffer
// stalled on the other shards' kernels).
namespace {
struct BufCache {
    std::mutex mu;
    struct Info { size_t bytes; int dev; };
    std::unordered_map<void*, Info> live;                  // what this library has allocated
    std::multimap<size_t, void*> idle[16];
    size_t idle_bytes[16] = {};
};
BufCache& buf_cache() { static BufCache b; return b; }
size_t cache_max() {       // bytes per device (SMCN_DEVICE_CACHE_MB, read once)
    static const size_t v = []() {
        const char* e = getenv("SMCN_DEVICE_CACHE_MB");
        return (size_t)(e ? (atoll(e) > 0 ? atoll(e) : 0) : 1024) << 20;
    }();
    return v;
}

hipError_t cached_malloc(void** p, size_t n) {
    const size_t bytes = (n + 255) & ~(size_t)255;
    int dev = 0;
    (void)hipGetDevice(&dev);
    BufCache& bc = buf_cache();
    if (dev >= 0 && dev < 16) {
        void* hit = nullptr;
        {
            std::lock_guard<std::mutex> g(bc.mu);
            auto it = bc.idle[dev].find(bytes);
            if (it != bc.idle[dev].end()) {
                hit = it->second;
                bc.idle[dev].erase(it);
                bc.idle_bytes[dev] -= bytes;
                bc.live[hit] = {bytes, dev};
            }
        }
        if (hit) {
            // (what fresh memory of this driver reads as.  hipMemset returns before the device has done it, and the null
            //  stream is not ordered against the contexts' non-blocking streams: wait.)
            hipError_t e = hipMemsetAsync(hit, 0, bytes, nullptr);
            if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
            if (e != hipSuccess) return e;
            *p = hit;
            return hipSuccess;
        }
    }
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess && dev >= 0 && dev < 16) {           // out of memory: give the cache back and try once more
        std::vector<void*> drop;
        {
            std::lock_guard<std::mutex> g(bc.mu);
            for (auto& kv : bc.idle[dev]) drop.push_back(kv.second);
            bc.idle[dev].clear();
            bc.idle_bytes[dev] = 0;
        }
        (void)hipGetLastError();
        for (void* q : drop) (void)hipFree(q);
        e = hipMalloc(p, bytes);
    }
    if (e == hipSuccess) {
        std::lock_guard<std::mutex> g(bc.mu);
        bc.live[*p] = {bytes, dev};
    }
    return e;
}
hipError_t cached_free(void* p, bool synced = false, bool keep = true) {
    if (!p) return hipSuccess;
    BufCache& bc = buf_cache();
    BufCache::Info info{0, -1};
    {
        std::lock_guard<std::mutex> g(bc.mu);
        auto it = bc.live.find(p);
        if (it != bc.live.end()) { info = it->second; bc.live.erase(it); }
    }
    if (info.dev < 0 || info.dev >= 16 || !keep) return hipFree(p);   // not ours (or no slot, or not worth keeping): the driver's
    if (!synced) (void)hipDeviceSynchronize();
    {
        std::lock_guard<std::mutex> g(bc.mu);
        if (bc.idle_bytes[info.dev] + info.bytes <= cache_max()) {
            bc.idle[info.dev].emplace(info.bytes, p);
            bc.idle_bytes[info.dev] += info.bytes;
            return hipSuccess;
        }
    }
    return hipFree(p);
}
}  // namespace

template <class T>
static hipError_t dalloc(T** p, int64_t n) {
    return cached_malloc((void**)p, sizeof(T) * (size_t)(n > 0 ? n : 1));
}

#ifdef SMCN_TRACE_SETUP   // (diagnostic build: where a context's set-up time goes, on stderr)
#include <chrono>
struct SetupTrace {
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void operator()(const char* what) {
        const auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "  setup %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};
#define SETUP_TRACE_DECL SetupTrace setup_trace_
#define SETUP_TRACE(w) setup_trace_(w)
#else
#define SETUP_TRACE_DECL
#define SETUP_TRACE(w)
#endif

// ---- streams -----------------------------------------------------------------------------------------------------------
// Creating a stream costs 2.4 - 8.7 ms on this stack (the first ones of a process map new hardware queues), and a context
// has two (its own, and the one its history travels on): more than everything else a constructor does.  Streams are
// therefore never destroyed: a context gives its streams back to a pool per device, and the first context of a process
// has a thread make three spare ones while it goes on, so that the next context (the cold `SMCSampler(...)` beside a
// running one) finds them ready.
namespace {
#ifndef SMCN_POOL_SPARE
#define SMCN_POOL_SPARE 3
#endif
constexpr int kPoolDevices = 16, kPoolSpare = SMCN_POOL_SPARE, kPoolMax = 32;
struct StreamPool {
    std::mutex mu;
    std::vector<hipStream_t> idle[kPoolDevices];
    bool prefilled[kPoolDevices] = {};
    std::thread filler[kPoolDevices];
    ~StreamPool() {                      // (the streams themselves go with the process)
        for (auto& t : filler)
            if (t.joinable()) t.join();
    }
};
StreamPool& stream_pool() { static StreamPool p; return p; }

hipError_t pool_take(int dev, hipStream_t* out) {
    StreamPool& sp = stream_pool();
    bool fill = false;
    if (dev >= 0 && dev < kPoolDevices) {
        std::lock_guard<std::mutex> g(sp.mu);
        if (!sp.idle[dev].empty()) {
            *out = sp.idle[dev].back();
            sp.idle[dev].pop_back();
            return hipSuccess;
        }
        fill = !sp.prefilled[dev];
        sp.prefilled[dev] = true;
    }
    const hipError_t e = hipStreamCreateWithFlags(out, hipStreamNonBlocking);
    if (fill && e == hipSuccess) {       // (once per device: `prefilled`)
        sp.filler[dev] = std::thread([dev]() {
            if (hipSetDevice(dev) != hipSuccess) return;
            for (int i = 0; i < kPoolSpare; ++i) {
                hipStream_t s = nullptr;
                if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return;
                StreamPool& q = stream_pool();
                std::lock_guard<std::mutex> g2(q.mu);
                q.idle[dev].push_back(s);
            }
        });
    }
    return e;
}
void pool_give(int dev, hipStream_t s) {
    if (!s) return;
    (void)hipStreamSynchronize(s);
    StreamPool& sp = stream_pool();
    if (dev >= 0 && dev < kPoolDevices) {
        std::lock_guard<std::mutex> g(sp.mu);
        if ((int)sp.idle[dev].size() < kPoolMax) { sp.idle[dev].push_back(s); return; }
    }
    (void)hipStreamDestroy(s);
}
}  // namespace

// ---- model dispatch ------------------------------------------------------------
// Calls f(Model{}) with the device functor matching (model id, data).
template <class F>
static int with_model(smcn_ctx* c, F&& f) {
    if (c->model == SMCN_MODEL_GAUSS) {
        if (c->D <= 4) return f(GaussModel<4, 1>{});
        if (c->D <= 32) return f(GaussModel<32, 1>{});
        if (c->D <= 64) return f(GaussModel<64, 1>{});
#ifdef SMCN_VARIANTS   // A/B builds: particles per wavefront x tree-stack levels in LDS at D <= 256
        if (c->D <= 256 && getenv("SMCN_GAUSS256")) {
            const int v = atoi(getenv("SMCN_GAUSS256"));
            if (v == 1) return f(GaussModel<16, 16, 0>{});
            if (v == 2) return f(GaussModel<16, 16, 1>{});
            if (v == 3) return f(GaussModel<32, 8, 1>{});
            if (v == 4) return f(GaussModel<32, 8, 2>{});
            if (v == 5) return f(GaussModel<64, 4, 4, 1>{});   // one wavefront per SIMD, tree-stack levels 0-3 in LDS
            if (v == 6) return f(GaussModel<64, 4, 3, 1>{});
            if (v == 7) return f(GaussModel<64, 4, 1, 3>{});   // three wavefronts per SIMD (<= 168 VGPRs), one LDS level
            if (v == 8) return f(GaussModel<64, 4, 0, 3>{});   // three wavefronts per SIMD, no LDS level
        }
#endif
        if (c->D <= 256) return f(GaussModel<64, 4>{});   // tree stack in HBM (BASELINE config 5)
        if (c->D <= 512) return f(GaussModel<64, 8, 2, 1>{});   // one wavefront per SIMD: 512 registers hold 8 coordinates per lane without scratch (two per SIMD spilled 620-756 B per lane)
        FAIL(c, "Gaussian target: the device functor covers D <= 512; larger targets run host-evaluated "
                "(SMCN_MODEL_HOST + smcn_set_host_target: any object with logpdf / logpdfgrad through HostTarget)");
    }
    if (c->model == SMCN_MODEL_ARMA) {
        // the product runs arma in the lane kernels (smcn_nuts3.hpp: any series length); the group functors
        // (8 lanes x 25 time steps, scan of the carries) remain for A/B builds
#ifdef SMCN_VARIANTS
        const int T = (int)c->mdata_h[0];
        static const bool pair = !(getenv("SMCN_ARMA_PAIR") && atoi(getenv("SMCN_ARMA_PAIR")) == 0);
        if (!pair && T == 200) return f(ArmaModel<8, 25, true>{});
        if (!pair && T >= 1 && T < 200) return f(ArmaModel<8, 25, false>{});
        if (T == 200) return f(ArmaModel<8, 25, true, 2>{});
        if (T >= 1 && T < 200) return f(ArmaModel<8, 25, false, 2>{});
#endif
        FAIL(c, "arma target: this entry point has no group functor (lane kernels only)");
    }
    if (c->model == SMCN_MODEL_PRMWCD) {
        const int nobs = (int)c->mdata_h[0], M = (int)c->mdata_h[1], C = (int)c->mdata_h[2];
#ifdef SMCN_VARIANTS   // A/B builds (the shipped shape only): 16 lanes, or the replicated-state functor (v2 kernel)
        if (nobs == 100 && C == 11 && M == 12) {
            static const int dist = getenv("SMCN_PRMWCD_DIST") ? atoi(getenv("SMCN_PRMWCD_DIST")) : 8;
            if (dist == 16) return f(PrmwcdDistModel<16, 100, 11, 0, 2>{});
            if (dist == 162) return f(PrmwcdDistModel<16, 100, 11, 2, 4>{});   // round 4: is the launch its longest tree's
            if (dist == 322) return f(PrmwcdDistModel<32, 100, 11, 2, 4>{});   // critical path?  shorter leaves, fewer trees per wave
            if (dist != 8) return f(PrmwcdModel<16, 100, 11>{});
        }
#endif
        // state distributed over 8 lanes (v1 kernel, hybrid LDS/HBM tree stack); 100 observations and 11 kernel
        // columns are the functor's capacity (the shipped data fills it), smaller data sets run in the same kernel
        if (nobs >= 1 && nobs <= 100 && C >= 1 && C <= 11 && M == C + 1) {
            if ((int64_t)c->mdata_h.size() != 4 + (int64_t)nobs * (C + 1))
                FAIL(c, "PRMwCD target: data = [N, M, Clength, q, y_1..y_N, Xkernel (N x Clength, row-major)]");
            const bool fast_shape = nobs > 96 && C == 11 && c->mdata_h[3] == 0.5;   // what the FAST functors are unrolled for
#ifdef SMCN_VARIANTS   // A/B builds (round 5): four lanes per particle; one lane per particle (phase 1 of a two-phase launch)
            if (nobs == 100 && fast_shape) {
                static const int g4 = getenv("SMCN_PRMWCD_G4") ? atoi(getenv("SMCN_PRMWCD_G4")) : 0;
                static const int lane1 = getenv("SMCN_PRMWCD_LANE") ? atoi(getenv("SMCN_PRMWCD_LANE")) : 0;
                if (lane1 == 1) return f(PrmwcdLaneModel<100, 11, 1>{});
                if (lane1 == 2) return f(PrmwcdLaneModel<100, 11, 1, true>{});
                if (g4 == 1) return f(PrmwcdDistModel<4, 100, 11, 2, 4, true, 1>{});
                if (g4 == 2) return f(PrmwcdDistModel<4, 100, 11, 2, 1, true, 2>{});
            }
#endif
            if (fast_shape) return f(PrmwcdDistModel<8, 100, 11, 2, 4, true>{});   // the shipped shape: unrolled observation loop
            return f(PrmwcdDistModel<8, 100, 11, 2, 4>{});
        }
        FAIL(c, "PRMwCD target: the device functor holds up to N=100 observations and Clength=11 columns (M = Clength + 1); "
                "larger data: pass the model object as a host-evaluated target");
    }
    if (c->model == SMCN_MODEL_HOST) FAIL(c, "host target: this entry point needs a device-native model");
    FAIL(c, "model not available in this build");
}

extern "C" {

int smcn_version(void) { return 1; }

int smcn_device_cache_trim(int device, int64_t* released_bytes, int64_t* idle_bytes) {
    BufCache& bc = buf_cache();
    int64_t idle = 0, released = 0;
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (int dev = 0; dev < 16; ++dev) {
        if (device >= 0 && dev != device) continue;
        std::vector<std::pair<size_t, void*>> drop;
        {
            std::lock_guard<std::mutex> g(bc.mu);
            idle += (int64_t)bc.idle_bytes[dev];
            for (auto& kv : bc.idle[dev]) drop.push_back(kv);
            bc.idle[dev].clear();
            bc.idle_bytes[dev] = 0;
        }
        if (drop.empty()) continue;
        (void)hipSetDevice(dev);
        for (auto& kv : drop)
            if (hipFree(kv.second) == hipSuccess) released += (int64_t)kv.first;
    }
    (void)hipSetDevice(cur);
    if (released_bytes) *released_bytes = released;
    if (idle_bytes) *idle_bytes = idle;
    return 0;
}

const char* smcn_last_error(const smcn_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

static void free_all(smcn_ctx* c) {
    void* ptrs[] = {c->mdata, c->x, c->x_new, c->x_tmp, c->r, c->r_new, c->logw, c->logw_new, c->wn, c->work,
                    c->lpri0, c->llik0, c->lpri1, c->llik1, c->Lg, c->qv, c->scan_local, c->ttot, c->toff, c->part,
                    c->scal, c->stage, c->stage2, c->nleap, c->depth, c->ndraws, c->flags, c->idx, c->queue,
                    c->tape_d, c->tape_off_d, c->prof, c->hist, c->ss, c->lp, c->gath, c->hist_x, c->hist_logw, c->u_res, c->in_rec, c->out_rec, c->nuts_scratch, c->lpB, c->gathB, c->gen_x, c->gen_logw, c->cnt, c->shiftB, c->ss_scratch, c->n2_ovf, c->hc_vec, c->hc_sc, c->hc_gp, c->hc_gl, c->hc_st, c->kin0, c->kin1, c->moved_i, c->tb_state, c->tb_part, c->tb_local,
                    c->tb_gath, c->glk_buf, c->glk_xchg, c->nuts_resume, c->nuts_pend, c->nuts_mq, c->nuts_mq_rec, c->handover};
    if (c->rows_h) (void)hipHostFree(c->rows_h);
    if (c->hist_h) (void)hipHostFree(c->hist_h);
    if (c->ev_rows) (void)hipEventDestroy(c->ev_rows);
    // ONE wait for everything this context has in flight (its own stream, the history stream); the buffers below were
    // used by these streams only
    if (c->stream) (void)stream_wait(c->stream);
    if (c->dl_stream) (void)hipStreamSynchronize(c->dl_stream);
    if (c->dl_stream) pool_give(c->device, c->dl_stream);
    c->dl_stream = nullptr;
    if (c->dl_stage) (void)cached_free(c->dl_stage, true);
    if (c->comm && rccl().ok) (void)rccl().CommDestroy(c->comm);
    c->comm = nullptr;
    for (void* q : {(void*)c->g_ttot_all, (void*)c->g_toff_all, (void*)c->g_keys, (void*)c->g_keys_send, (void*)c->g_keys_recv,
                    (void*)c->g_rows_send, (void*)c->g_rows_recv, (void*)c->g_dest, (void*)c->g_order})
        if (q) (void)cached_free(q, true);
    for (void* p : ptrs)
        if (p) (void)cached_free(p, true);
    for (int i = 0; i < kTimerRing; ++i) {
        if (c->ev0[i]) (void)hipEventDestroy(c->ev0[i]);
        if (c->ev1[i]) (void)hipEventDestroy(c->ev1[i]);
    }
    if (c->own_stream && c->stream) pool_give(c->device, c->stream);
    c->stream = nullptr;
}

int smcn_ctx_create(smcn_ctx** out, int device_id, int64_t n_particles, int64_t particle_base, int model_id,
                    const double* model_data, int64_t model_data_len) {
    if (!out || n_particles < 1 || !model_data || model_data_len < 1) {
        g_create_error = "smcn_ctx_create: bad arguments";
        return -1;
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1) {
        g_create_error = "smcn_ctx_create: no HIP device (this library has no CPU path)";
        return -2;
    }
    if (device_id < 0 || device_id >= ndev) {
        g_create_error = "smcn_ctx_create: device id out of range";
        return -1;
    }
    smcn_ctx* c = new smcn_ctx();
    memset(c->ev0, 0, sizeof c->ev0);
    memset(c->ev1, 0, sizeof c->ev1);
    c->device = device_id;
    c->N = n_particles;
    c->base = particle_base;
    c->model = model_id;
    c->mdata_h.assign(model_data, model_data + model_data_len);
    switch (model_id) {
        case SMCN_MODEL_GAUSS: c->D = (int)model_data[0]; break;
        case SMCN_MODEL_ARMA: c->D = 4; break;
        case SMCN_MODEL_PRMWCD: c->D = (int)model_data[1] + 1; break;
        case SMCN_MODEL_HOST: c->D = (int)model_data[0]; break;
        default:
            g_create_error = "smcn_ctx_create: unknown model id";
            delete c;
            return -1;
    }
    c->Dc = c->D;
    auto fail = [&](const char* what, hipError_t er) {
        g_create_error = std::string("smcn_ctx_create: ") + what + ": " + hipGetErrorString(er);
        free_all(c);
        delete c;
        return -2;
    };
    SETUP_TRACE_DECL;
    if ((e = hipSetDevice(device_id)) != hipSuccess) return fail("hipSetDevice", e);
    SETUP_TRACE("hipSetDevice");
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess) return fail("props", e);
    c->num_cu = prop.multiProcessorCount;
    SETUP_TRACE("hipGetDeviceProperties");
    if ((e = pool_take(device_id, &c->stream)) != hipSuccess) return fail("stream", e);
    c->own_stream = true;
    SETUP_TRACE("stream");
    const int64_t N = c->N, ND = N * c->D;
    const int nt = grid_for(N, kScanTile);
#define A_(p, n) \
    if ((e = dalloc(&c->p, (n))) != hipSuccess) return fail(#p, e)
    // PRMwCD: behind the caller's data a table the one-lane-per-particle functor reads by scalar loads -- a row per observation,
    // [X_i1 .. X_iC, y_i] padded to an even count, at a 128-byte boundary, two zero rows behind it for the look-ahead
    std::vector<double> mup(model_data, model_data + model_data_len);
    if (model_id == SMCN_MODEL_PRMWCD && model_data_len >= 4) {
        const int nobs = (int)model_data[0], C = (int)model_data[2];
        if (nobs >= 1 && C >= 1 && model_data_len == 4 + (int64_t)nobs * (C + 1)) {
            const int RS = (C + 2) & ~1;
            mup.resize(((size_t)model_data_len + 15) / 16 * 16, 0.0);
            for (int i = 0; i < nobs + 2; ++i)
                for (int j = 0; j < RS; ++j)
                    mup.push_back(i < nobs ? (j < C ? model_data[4 + nobs + (size_t)i * C + j] : (j == RS - 1 ? model_data[4 + i] : 0.0)) : 0.0);
        }
    }
    const int64_t mlen = (int64_t)mup.size();
    A_(mdata, mlen + 32);   // padded: the lane kernels read the series one chunk ahead
    A_(x, ND); A_(x_new, ND); A_(x_tmp, ND); A_(r, ND); A_(r_new, ND);
    A_(logw, N); A_(logw_new, N); A_(wn, N); A_(work, N);
    A_(lpri0, N); A_(llik0, N); A_(lpri1, N); A_(llik1, N); A_(Lg, N); A_(qv, N);
    A_(scan_local, N); A_(ttot, nt + 1); A_(toff, nt + 2);
    A_(part, (int64_t)kMaxPart * (4 * c->D * c->D + 2 * c->D + 8)); A_(scal, 4 * c->D * c->D + 2 * c->D + 64);
    A_(stage, ND); A_(nleap, N); A_(depth, N); A_(ndraws, N); A_(flags, N); A_(idx, N); A_(queue, 16); A_(prof, 16);
#undef A_
    SETUP_TRACE("35 x hipMalloc");
    c->stage_len = ND;
    // (everything below goes through the context's OWN stream: it is non-blocking, so a hipMemset on the null stream is not
    //  ordered against it and could land AFTER the first kernels the caller enqueues -- seen with eight 1.6 GB contexts
    //  created at once: part of a shard's initial particles zeroed behind smcn_init_particles_std_normal)
    (void)hipMemsetAsync(c->mdata, 0, sizeof(double) * (mlen + 32), c->stream);
    if ((e = hipMemcpyAsync(c->mdata, mup.data(), sizeof(double) * mlen, hipMemcpyHostToDevice, c->stream)) != hipSuccess)
        return fail("mdata copy", e);
    if ((e = stream_wait(c->stream)) != hipSuccess) return fail("mdata copy", e);     // (mup is a local)
    (void)hipMemsetAsync(c->x, 0, sizeof(double) * ND, c->stream);
    (void)hipMemsetAsync(c->x_new, 0, sizeof(double) * ND, c->stream);
    (void)hipMemsetAsync(c->r, 0, sizeof(double) * ND, c->stream);
    (void)hipMemsetAsync(c->r_new, 0, sizeof(double) * ND, c->stream);
    (void)hipMemsetAsync(c->logw, 0, sizeof(double) * N, c->stream);
    (void)hipMemsetAsync(c->nleap, 0, sizeof(int32_t) * N, c->stream);
    (void)hipMemsetAsync(c->prof, 0, sizeof(unsigned long long) * 16, c->stream);
    if ((e = stream_wait(c->stream)) != hipSuccess) return fail("initial memsets", e);   // (model_data is the caller's)
    SETUP_TRACE("memsets + wait");
    for (int i = 0; i < kTimerRing; ++i) {
        if ((e = hipEventCreate(&c->ev0[i])) != hipSuccess) return fail("event", e);
        if ((e = hipEventCreate(&c->ev1[i])) != hipSuccess) return fail("event", e);
    }
    SETUP_TRACE("events");
    // refuse models this build has no device functor for, at creation time
    int rc = 0;
    if (model_id == SMCN_MODEL_HOST) {
        if (c->D < 1 || c->D > 4096) { c->err = "host target: D out of range"; rc = -1; }
    } else if (model_id == SMCN_MODEL_ARMA
#ifdef SMCN_VARIANTS   // (A/B builds can route arma to the group functors, which validate their own data)
               && getenv("SMCN_ARMA_NUTS2") == nullptr
#endif
    ) {
        const double T = model_data[0];
        if (!(T >= 1.0) || T != (double)(int64_t)T || (int64_t)T + 1 != model_data_len) { c->err = "arma target: data = [T, y_1..y_T]"; rc = -1; }
        else c->arma_T = (int64_t)T;
    } else {
        rc = with_model(c, [&](auto m) {
#ifdef SMCN_VARIANTS
            c->fused_ok = !decltype(m)::DIST && getenv("SMCN_NUTS_V1") == nullptr;
#endif
            return 0;
        });
    }
    bool lane = model_id == SMCN_MODEL_ARMA;
#ifdef SMCN_VARIANTS
    lane = lane && getenv("SMCN_ARMA_NUTS2") == nullptr;
#endif
    if (rc == 0 && lane) {
        c->lane_kernel = true;
        c->fused_ok = true;
    }
    if (rc != 0) {
        g_create_error = "smcn_ctx_create: " + c->err;
        free_all(c);
        delete c;
        return rc;
    }
    *out = c;
    return 0;
}

void smcn_ctx_destroy(smcn_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)stream_wait(c->stream);
    free_all(c);
    delete c;
}

int smcn_dim(const smcn_ctx* c) { return c ? c->D : -1; }
int smcn_constrained_dim(const smcn_ctx* c) { return c ? c->Dc : -1; }
int smcn_fused_transitions(const smcn_ctx* c) { return c ? (c->fused_ok ? 1 : 0) : -1; }

int smcn_set_stream(smcn_ctx* c, void* s) {
    CHECK_CTX(c);
    HIPC(c, stream_wait(c->stream));
    if (c->own_stream && c->stream) pool_give(c->device, c->stream);
    c->stream = (hipStream_t)s;
    c->own_stream = false;
    return 0;
}

int smcn_synchronize(smcn_ctx* c) {
    CHECK_CTX(c);
    HIPC(c, stream_wait(c->stream));
    return 0;
}

// 0: multinomial (rng.choice, samples.py:139 -- the reference); 1: systematic (one uniform per
// resampling, keys (i + u0) / N on the same CDF and search)
int smcn_set_resample_scheme(smcn_ctx* c, int scheme) {
    CHECK_CTX(c);
    if (scheme != 0 && scheme != 1) FAIL(c, "smcn_set_resample_scheme: 0 (multinomial) or 1 (systematic)");
    c->resample_scheme = scheme;
    return 0;
}
int smcn_set_wide_eval(smcn_ctx* c, int on) {
    CHECK_CTX(c);
    if (on != 0 && on != 1) FAIL(c, "smcn_set_wide_eval: 0 or 1");
    c->wide_eval = on;
    return 0;
}

// nuts3_kernel (one lane per particle): the grid is capped at `waves` wavefronts; with fewer lanes than particles every
// wavefront works through a contiguous run of particles (smcn_nuts3.hpp, QUEUE).  0 (default): one wavefront per SIMD;
// < 0: no cap -- a wavefront per 64 particles, the round-3 schedule (A/B, tests: the results do not depend on the schedule
// with smcn_set_wide_eval(0)).
int smcn_set_lane_grid(smcn_ctx* c, int64_t waves) {
    CHECK_CTX(c);
    c->lane_grid_cap = waves;
    return 0;
}

// ... and in how many SEGMENTS a particle's block of transitions is worked off then (0 = auto: 4; 1 = a lane keeps a
// particle for the whole block).
int smcn_set_lane_segments(smcn_ctx* c, int segments) {
    CHECK_CTX(c);
    if (segments < 0 || segments > 64) FAIL(c, "smcn_set_lane_segments: 0 (auto) .. 64");
    c->lane_segments = segments;
    return 0;
}

int smcn_set_host_target(smcn_ctx* c, smcn_host_target_fn fn, void* user) {
    CHECK_CTX(c);
    if (c->model != SMCN_MODEL_HOST) FAIL(c, "smcn_set_host_target: the context was not created with SMCN_MODEL_HOST");
    c->host_fn = fn;
    c->host_user = user;
    return 0;
}

int smcn_set_seed(smcn_ctx* c, uint64_t seed) {
    CHECK_CTX(c);
    c->seed = seed;
    return 0;
}

// ---- staging helpers --------------------------------------------------------------
static int ensure_stage(smcn_ctx* c, int64_t n) {
    if (n <= c->stage_len) return 0;
    if (c->stage) (void)cached_free(c->stage);
    c->stage = nullptr;
    HIPC(c, dalloc(&c->stage, n));
    c->stage_len = n;
    return 0;
}
static int ensure_stage2(smcn_ctx* c, int64_t n) {
    if (n <= c->stage2_len) return 0;
    if (c->stage2) (void)cached_free(c->stage2);
    c->stage2 = nullptr;
    HIPC(c, dalloc(&c->stage2, n));
    c->stage2_len = n;
    return 0;
}
// host [N][D] -> device [D][N]
static int upload_nd(smcn_ctx* c, const double* h, double* d) {
    const int64_t n = c->N * c->D;
    int rc = ensure_stage(c, n);
    if (rc) return rc;
    HIPC(c, hipMemcpyAsync(c->stage, h, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    transpose_kernel<<<grid_for(n, 256), 256, 0, c->stream>>>(c->stage, d, c->N, c->D);
    HIPC(c, hipGetLastError());
    HIPC(c, stream_wait(c->stream));
    return 0;
}
// device [D][N] -> host [N][D]
static int download_nd(smcn_ctx* c, const double* d, double* h) {
    const int64_t n = c->N * c->D;
    int rc = ensure_stage(c, n);
    if (rc) return rc;
    transpose_kernel<<<grid_for(n, 256), 256, 0, c->stream>>>(d, c->stage, c->D, c->N);
    HIPC(c, hipGetLastError());
    HIPC(c, hipMemcpyAsync(h, c->stage, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
    HIPC(c, stream_wait(c->stream));
    return 0;
}
static int download_n(smcn_ctx* c, const void* d, void* h, size_t elem) {
    HIPC(c, hipMemcpyAsync(h, d, elem * (size_t)c->N, hipMemcpyDeviceToHost, c->stream));
    HIPC(c, stream_wait(c->stream));
    return 0;
}

// r / r_new back to [D][N] where a wave-kernel launch (or its momentum draw) left them particle-major
static int momentum_dn(smcn_ctx* c) {
    const int64_t n = c->N * c->D;
    for (int which = 0; which < 2; ++which) {
        bool& pm = which ? c->r_new_pm : c->r_pm;
        double* const buf = which ? c->r_new : c->r;
        if (!pm) continue;
        int rc = ensure_stage(c, n);
        if (rc) return rc;
        transpose_kernel<<<grid_for(n, 256), 256, 0, c->stream>>>(buf, c->stage, c->N, c->D);     // [N][D] -> [D][N]
        HIPC(c, hipGetLastError());
        HIPC(c, hipMemcpyAsync(buf, c->stage, sizeof(double) * n, hipMemcpyDeviceToDevice, c->stream));
        pm = false;
    }
    return 0;
}
// device momentum -> host [N][D]
static int download_momentum(smcn_ctx* c, const double* d, bool pm, double* h) {
    if (!pm) return download_nd(c, d, h);
    HIPC(c, hipMemcpyAsync(h, d, sizeof(double) * c->N * c->D, hipMemcpyDeviceToHost, c->stream));
    HIPC(c, stream_wait(c->stream));
    return 0;
}

static int eval_resident(smcn_ctx* c, const double* xdev, double phi, double* logp, double* lpri, double* llik);

int smcn_set_state(smcn_ctx* c, const double* x, const double* logw) {
    CHECK_CTX(c);
    HIPC(c, hipSetDevice(c->device));
    int rc = 0;
    if (x && (rc = upload_nd(c, x, c->x))) return rc;
    if (logw) {
        HIPC(c, hipMemcpyAsync(c->logw, logw, sizeof(double) * c->N, hipMemcpyHostToDevice, c->stream));
        HIPC(c, stream_wait(c->stream));
    }
    return 0;
}

int smcn_get_state(smcn_ctx* c, double* x, double* logw, double* wn) {
    CHECK_CTX(c);
    int rc = 0;
    if (x && (rc = download_nd(c, c->x, x))) return rc;
    if (logw && (rc = download_n(c, c->logw, logw, sizeof(double)))) return rc;
    if (wn && (rc = download_n(c, c->wn, wn, sizeof(double)))) return rc;
    return 0;
}

int smcn_get_proposal(smcn_ctx* c, double* r, double* x_new, double* r_new, double* logw_new) {
    CHECK_CTX(c);
    int rc = 0;
    if (r && (rc = download_momentum(c, c->r, c->r_pm, r))) return rc;
    if (x_new && (rc = download_nd(c, c->x_new, x_new))) return rc;
    if (r_new && (rc = download_momentum(c, c->r_new, c->r_new_pm, r_new))) return rc;
    if (logw_new && (rc = download_n(c, c->logw_new, logw_new, sizeof(double)))) return rc;
    return 0;
}

// A proposal computed elsewhere (tests: the reference's recorded r, x', r'): uploads it and evaluates the
// density parts at x (-> lpri0/llik0) and x' (-> lpri1/llik1), i.e. the state smcn_propose_nuts leaves behind.
int smcn_set_proposal(smcn_ctx* c, const double* r, const double* x_new, const double* r_new) {
    CHECK_CTX(c);
    if (!r || !x_new || !r_new) FAIL(c, "smcn_set_proposal: null");
    int rc = 0;
    if ((rc = upload_nd(c, r, c->r))) return rc;
    if ((rc = upload_nd(c, x_new, c->x_new))) return rc;
    if ((rc = upload_nd(c, r_new, c->r_new))) return rc;
    c->r_pm = c->r_new_pm = false;
    if ((rc = eval_resident(c, c->x, 1.0, nullptr, c->lpri0, c->llik0))) return rc;
    if ((rc = eval_resident(c, c->x_new, 1.0, nullptr, c->lpri1, c->llik1))) return rc;
    HIPC(c, hipMemsetAsync(c->nleap, 0, sizeof(int32_t) * c->N, c->stream));
    HIPC(c, stream_wait(c->stream));
    c->momentum_set = false;
    c->lg_set = false;
    c->kin_valid = false;       // |r|^2, |r'|^2 and the moved flags of the last NUTS launch describe another proposal
    return 0;
}

int smcn_set_momentum(smcn_ctx* c, const double* r) {
    CHECK_CTX(c);
    if (!r) FAIL(c, "smcn_set_momentum: null");
    int rc = upload_nd(c, r, c->r);
    if (rc) return rc;
    c->r_pm = false;
    c->momentum_set = true;
    c->kin_valid = false;
    return 0;
}

// ---- target --------------------------------------------------------------------------
}  // extern "C"
template <class LaneModel>
static int launch_lane_eval(smcn_ctx* c, const double* x, int64_t M, int64_t rs, int64_t cs, double phi, double* logp,
                            double* grad, int64_t grs, int64_t gcs, double* lpri, double* llik) {
    int64_t blocks = (M + 255) / 256;
    const int64_t cap = (int64_t)c->num_cu * 8;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    lane_eval_kernel<LaneModel><<<(int)blocks, 256, 0, c->stream>>>(c->mdata, x, M, rs, cs, phi, logp, grad, grs, gcs, lpri,
                                                                    llik);
    HIPC(c, hipGetLastError());
    return 0;
}

template <class Model>
static int launch_eval(smcn_ctx* c, Model, const double* x, int64_t M, int64_t rs, int64_t cs, double phi,
                       double* logp, double* grad, int64_t grs, int64_t gcs, double* lpri, double* llik) {
    constexpr int gpb = 256 / Model::G;
    int64_t blocks = (M + gpb - 1) / gpb;
    const int64_t cap = (int64_t)c->num_cu * 8;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    eval_kernel<Model><<<(int)blocks, 256, sizeof(double) * Model::SHARED, c->stream>>>(
        c->mdata, x, M, rs, cs, phi, logp, grad, grs, gcs, lpri, llik);
    HIPC(c, hipGetLastError());
    return 0;
}

extern "C" {
int smcn_target_eval(smcn_ctx* c, const double* x, int64_t M, double phi, double* logp, double* grad, double* lpri,
                     double* llik) {
    CHECK_CTX(c);
    if (!x || M < 1) FAIL(c, "smcn_target_eval: bad arguments");
    HIPC(c, hipSetDevice(c->device));
    const int64_t n = M * c->D;
    int rc = ensure_stage(c, n);
    if (rc) return rc;
    if ((rc = ensure_stage2(c, n + 3 * M))) return rc;
    HIPC(c, hipMemcpyAsync(c->stage, x, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    double* d_grad = c->stage2;
    double* d_lp = c->stage2 + n;
    double* d_pri = d_lp + M;
    double* d_lik = d_pri + M;
    if (c->lane_kernel)
        rc = launch_lane_eval<ArmaLaneModel>(c, c->stage, M, c->D, 1, phi, d_lp, grad ? d_grad : nullptr, c->D, 1, d_pri, d_lik);
    else
        rc = with_model(c, [&](auto m) {
            return launch_eval(c, m, c->stage, M, c->D, 1, phi, d_lp, grad ? d_grad : nullptr, c->D, 1, d_pri, d_lik);
        });
    if (rc) return rc;
    if (logp) HIPC(c, hipMemcpyAsync(logp, d_lp, sizeof(double) * M, hipMemcpyDeviceToHost, c->stream));
    if (grad) HIPC(c, hipMemcpyAsync(grad, d_grad, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
    if (lpri) HIPC(c, hipMemcpyAsync(lpri, d_pri, sizeof(double) * M, hipMemcpyDeviceToHost, c->stream));
    if (llik) HIPC(c, hipMemcpyAsync(llik, d_lik, sizeof(double) * M, hipMemcpyDeviceToHost, c->stream));
    HIPC(c, stream_wait(c->stream));
    return 0;
}

int smcn_target_constrain(smcn_ctx* c, const double* x, int64_t M, double* out) {
    CHECK_CTX(c);
    if (!x || !out || M < 1) FAIL(c, "smcn_target_constrain: bad arguments");
    const int64_t n = M * c->D;
    int rc = ensure_stage(c, n);
    if (rc) return rc;
    if ((rc = ensure_stage2(c, n))) return rc;
    HIPC(c, hipMemcpyAsync(c->stage, x, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    constrain_kernel<<<grid_for(n, 256), 256, 0, c->stream>>>(c->stage, c->stage2, M, c->D, c->model);
    HIPC(c, hipGetLastError());
    HIPC(c, hipMemcpyAsync(out, c->stage2, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
    HIPC(c, stream_wait(c->stream));
    return 0;
}

// log pi_phi at the resident particles -> work[], parts -> lpri1/llik1
// SMCN_MODEL_HOST: value (and gradient parts, [D][N] on the device) of the caller's target at the
// resident positions xdev [D][N]
static int host_eval(smcn_ctx* c, const double* xdev, bool want_grad, double* lpri, double* llik, double* gpri,
                     double* glik) {
    if (!c->host_fn) FAIL(c, "host target: call smcn_set_host_target first");
    const int64_t N = c->N, ND = N * c->D;
    int rc = ensure_stage(c, ND);
    if (rc) return rc;
    c->hx.resize(ND); c->hlp.resize(N); c->hll.resize(N);
    if (want_grad) { c->hgp.resize(ND); c->hgl.resize(ND); }
    transpose_kernel<<<grid_for(ND, 256), 256, 0, c->stream>>>(xdev, c->stage, c->D, N);   // -> [N][D]
    HIPC(c, hipGetLastError());
    HIPC(c, hipMemcpyAsync(c->hx.data(), c->stage, sizeof(double) * ND, hipMemcpyDeviceToHost, c->stream));
    HIPC(c, stream_wait(c->stream));
    if (c->host_fn(c->host_user, N, c->D, c->hx.data(), want_grad ? 1 : 0, c->hlp.data(), c->hll.data(),
                   want_grad ? c->hgp.data() : nullptr, want_grad ? c->hgl.data() : nullptr) != 0)
        FAIL(c, "host target: the callback reported an error");
    HIPC(c, hipMemcpyAsync(lpri, c->hlp.data(), sizeof(double) * N, hipMemcpyHostToDevice, c->stream));
    HIPC(c, hipMemcpyAsync(llik, c->hll.data(), sizeof(double) * N, hipMemcpyHostToDevice, c->stream));
    if (want_grad) {
        for (int w = 0; w < 2; ++w) {
            HIPC(c, hipMemcpyAsync(c->stage, (w ? c->hgl : c->hgp).data(), sizeof(double) * ND, hipMemcpyHostToDevice,
                                   c->stream));
            transpose_kernel<<<grid_for(ND, 256), 256, 0, c->stream>>>(c->stage, w ? glik : gpri, N, c->D);   // -> [D][N]
            HIPC(c, hipGetLastError());
        }
    }
    return 0;
}
__global__ void host_logp_kernel(const double* lpri, const double* llik, double phi, double* logp, int64_t N) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) logp[i] = combine_lp(lpri[i], llik[i], phi);
}

static int eval_resident(smcn_ctx* c, const double* xdev, double phi, double* logp, double* lpri, double* llik) {
    if (c->model == SMCN_MODEL_HOST) {
        double* lp_ = lpri ? lpri : c->lpri1;
        double* ll_ = llik ? llik : c->llik1;
        int rc = host_eval(c, xdev, false, lp_, ll_, nullptr, nullptr);
        if (rc) return rc;
        if (logp) host_logp_kernel<<<grid_for(c->N, 256), 256, 0, c->stream>>>(lp_, ll_, phi, logp, c->N);
        HIPC(c, hipGetLastError());
        return 0;
    }
    if (c->lane_kernel) return launch_lane_eval<ArmaLaneModel>(c, xdev, c->N, 1, c->N, phi, logp, nullptr, 0, 0, lpri, llik);
    return with_model(c, [&](auto m) {
        return launch_eval(c, m, xdev, c->N, 1, c->N, phi, logp, nullptr, 0, 0, lpri, llik);
    });
}

int smcn_init_weights(smcn_ctx* c, double phi, const double* logq0) {
    CHECK_CTX(c);
    HIPC(c, hipSetDevice(c->device));
    const int64_t N = c->N;
    int rc = eval_resident(c, c->x, phi, c->work, c->lpri1, c->llik1);
    if (rc) return rc;
    double* lq = nullptr;
    if (logq0) {
        HIPC(c, hipMemcpyAsync(c->Lg, logq0, sizeof(double) * N, hipMemcpyHostToDevice, c->stream));
        lq = c->Lg;
    }
    init_logw_kernel<<<grid_for(N, 256), 256, 0, c->stream>>>(c->work, lq, c->x, c->logw, N, c->D);
    HIPC(c, hipGetLastError());
    HIPC(c, stream_wait(c->stream));
    return 0;
}

int smcn_init_particles_std_normal(smcn_ctx* c, double phi) {
    CHECK_CTX(c);
    HIPC(c, hipSetDevice(c->device));
    const int64_t n = c->N * ((c->D + 1) / 2);
    normals_kernel<<<grid_for(n, 256), 256, 0, c->stream>>>(c->x, c->N, c->D, c->base, c->seed, 0u, kStreamInit);
    HIPC(c, hipGetLastError());
    return smcn_init_weights(c, phi, nullptr);
}

// ---- normalise / ESS -------------------------------------------------------------------
static int lse_partials(smcn_ctx* c, const double* a, double out[4]) {
    const int64_t N = c->N;
    const int g = red_grid(N);
    max_partial_kernel<<<g, kRedBlock, 0, c->stream>>>(a, N, c->part);
    max_final_kernel<<<1, kRedBlock, 0, c->stream>>>(c->part, g, c->scal);
    lse_partial_kernel<<<g, kRedBlock, 0, c->stream>>>(a, N, c->scal, c->part);
    sum_final_kernel<<<final_grid(3), kRedBlock, 0, c->stream>>>(c->part, g, 3, c->scal + 1);
    HIPC(c, hipGetLastError());
    HIPC(c, hipMemcpyAsync(out, c->scal, sizeof(double) * 4, hipMemcpyDeviceToHost, c->stream));
    HIPC(c, stream_wait(c->stream));
    return 0;
}

int smcn_normalise_partials(smcn_ctx* c, double out[4]) {
    CHECK_CTX(c);
    Range roctx_range("smcn:normalise");
    HIPC(c, hipSetDevice(c->device));
    return lse_partials(c, c->logw, out);
}

int smcn_normalise_apply(smcn_ctx* c, double loglik) {
    CHECK_CTX(c);
    Range roctx_range("smcn:normalise");
    wn_kernel<<<grid_for(c->N, 256), 256, 0, c->stream>>>(c->logw, c->wn, c->N, loglik);
    HIPC(c, hipGetLastError());
    return 0;
}

int smcn_normalise(smcn_ctx* c, double* loglik, double* ess) {
    CHECK_CTX(c);
    double p[4];
    int rc = smcn_normalise_partials(c, p);
    if (rc) return rc;
    // scipy logsumexp: log1p(s/m) + log(m) + max   (s = 0 stays 0)
    const double mx = p[0], m = p[1], s1 = p[2], s2 = p[3];
    const double s = (s1 == 0.0) ? s1 : s1 / m;
    const double ll = log1p(s) + log(m) + mx;
    const double shift = std::isfinite(mx) ? mx : 0.0;
    if (loglik) *loglik = ll;
    if (ess) *ess = 1.0 / (s2 * exp(2.0 * (shift - ll)));
    return smcn_normalise_apply(c, ll);
}

int smcn_temper_partials(smcn_ctx* c, double phi_old, double phi_new, double out[4]) {
    CHECK_CTX(c);
    Range roctx_range("smcn:tempering");
    HIPC(c, hipSetDevice(c->device));
    temper_logw_kernel<<<grid_for(c->N, 256), 256, 0, c->stream>>>(c->lpri1, c->llik1, c->work, c->N, phi_old,
                                                                   phi_new);
    HIPC(c, hipGetLastError());
    return lse_partials(c, c->work, out);
}

// ---- ESSTempering.calculate_phi on the device (smcn_temper.hpp) ------------------------------------------------
static int tb_ensure(smcn_ctx* c, int world) {
    if (world < 1 || world > 64) FAIL(c, "smcn_temper_bisect: 1..64 shards");
    if (!c->tb_state) {
        c->tb_blocks = (int)std::min<int64_t>(256, std::max<int64_t>(1, (c->N + 511) / 512));
        HIPC(c, dalloc(&c->tb_state, TB_STATE));
        HIPC(c, dalloc(&c->tb_part, (int64_t)c->tb_blocks * kTbNodes * 4));
        HIPC(c, dalloc(&c->tb_local, kTbNodes * 4));
        HIPC(c, hipMemsetAsync(c->tb_state, 0, sizeof(double) * TB_STATE, c->stream));
    }
    if (c->tb_world < world) {
        HIPC(c, stream_wait(c->stream));
        if (c->tb_gath) (void)cached_free(c->tb_gath);
        c->tb_gath = nullptr;
        HIPC(c, dalloc(&c->tb_gath, (int64_t)world * kTbNodes * 4));
        c->tb_world = world;
    }
    return 0;
}
// one pass, this shard's part: partials of the trial points, blocks merged -> the local buffer (asynchronous)
int smcn_temper_bisect_pass(smcn_ctx* c, int pass, double phi_old) {
    CHECK_CTX(c);
    Range roctx_range("smcn:tempering");
    if (pass < 0 || pass >= kTbPasses) FAIL(c, "smcn_temper_bisect_pass: bad pass");
    int rc = tb_ensure(c, 1);
    if (rc) return rc;
    temper_multi_partial_kernel<<<c->tb_blocks, 256, 0, c->stream>>>(c->lpri1, c->llik1, c->N, phi_old, c->tb_state, pass,
                                                                    c->tb_part);
    temper_multi_local_kernel<<<kTbNodes, 64, 0, c->stream>>>(c->tb_part, c->tb_blocks, c->tb_state, pass, c->tb_local);
    HIPC(c, hipGetLastError());
    return 0;
}
// device buffers of the exchange between shards: this shard's [15][4] partials, all shards' [world][15][4]
int smcn_temper_bisect_buffers(smcn_ctx* c, int world, void** local, void** gathered) {
    CHECK_CTX(c);
    int rc = tb_ensure(c, world);
    if (rc) return rc;
    if (local) *local = c->tb_local;
    if (gathered) *gathered = c->tb_gath;
    return 0;
}
// after the all-gather (world == 1: straight from the local buffer): f at the trial points, bisect.c's steps (asynchronous)
int smcn_temper_bisect_decide(smcn_ctx* c, int pass, int world, double target, double phi_old) {
    CHECK_CTX(c);
    if (pass < 0 || pass >= kTbPasses) FAIL(c, "smcn_temper_bisect_decide: bad pass");
    int rc = tb_ensure(c, world);
    if (rc) return rc;
    temper_multi_decide_kernel<<<1, 64, 0, c->stream>>>(world == 1 ? c->tb_local : c->tb_gath, world, target, phi_old, pass,
                                                       c->tb_state);
    HIPC(c, hipGetLastError());
    return 0;
}
// status: 0 the root (or 1.0) is in *phi; 1 still bisecting (enqueue more passes); 2 f(phi_old) and f(1) have the same
// sign (scipy's ValueError); 3 no convergence in 100 steps (scipy's RuntimeError)
int smcn_temper_bisect_result(smcn_ctx* c, double* phi, int* status) {
    CHECK_CTX(c);
    if (!phi || !status || !c->tb_state) FAIL(c, "smcn_temper_bisect_result: nothing to read");
    double st[TB_STATE];
    HIPC(c, hipMemcpyAsync(st, c->tb_state, sizeof(double) * TB_STATE, hipMemcpyDeviceToHost, c->stream));
    HIPC(c, stream_wait(c->stream));
    *phi = st[TB_RESULT];
    *status = st[TB_DONE] == 0.0 ? 1 : (st[TB_ERROR] == 1.0 ? 2 : (st[TB_ERROR] == 2.0 ? 3 : 0));
    return 0;
}
// one shard: the whole of ESSTempering.calculate_phi with ONE host synchronisation
int smcn_temper_bisect(smcn_ctx* c, double phi_old, double target, double* phi, int* status) {
    CHECK_CTX(c);
    if (!phi || !status) FAIL(c, "smcn_temper_bisect: null");
    int pass = 0, rc = 0;
    // the opening alone first (adaptive_tempering.py:58: ESS(1) >= target returns 1 -- every iteration once the ladder has
    // reached 1): one pass, one wait; a bisection that has to run is then enqueued whole (bisect.c ends within 40 steps
    // on brackets inside [0, 1]) behind a second wait
    if ((rc = smcn_temper_bisect_pass(c, 0, phi_old))) return rc;
    if ((rc = smcn_temper_bisect_decide(c, 0, 1, target, phi_old))) return rc;
    if ((rc = smcn_temper_bisect_result(c, phi, status))) return rc;
    if (*status != 1) return 0;
    pass = 1;
    const int first = 1 + (40 + kTbLevels - 1) / kTbLevels;
    for (int upto = first;; upto += 4) {
        for (; pass < upto && pass < kTbPasses; ++pass) {
            if ((rc = smcn_temper_bisect_pass(c, pass, phi_old))) return rc;
            if ((rc = smcn_temper_bisect_decide(c, pass, 1, target, phi_old))) return rc;
        }
        if ((rc = smcn_temper_bisect_result(c, phi, status))) return rc;
        if (*status != 1 || pass >= kTbPasses) return 0;
    }
}

// ---- moments ------------------------------------------------------------------------------
int smcn_moment_sums(smcn_ctx* c, const double* mean, double* sums) {
    CHECK_CTX(c);
    Range roctx_range("smcn:estimate");
    if (!sums) FAIL(c, "smcn_moment_sums: null");
    HIPC(c, hipSetDevice(c->device));
    const int g = red_grid(c->N);
    double* dmean = nullptr;
    if (mean) {
        dmean = c->scal + 16;
        HIPC(c, hipMemcpyAsync(dmean, mean, sizeof(double) * c->Dc, hipMemcpyHostToDevice, c->stream));
    }
    moment_partial_kernel<<<g, kRedBlock, 0, c->stream>>>(c->x, c->wn, c->N, c->D, c->model, dmean, c->part);
    sum_final_kernel<<<final_grid(c->D), kRedBlock, 0, c->stream>>>(c->part, g, c->D, c->scal + 16 + c->D);
    HIPC(c, hipGetLastError());
    HIPC(c, hipMemcpyAsync(sums, c->scal + 16 + c->D, sizeof(double) * c->Dc, hipMemcpyDeviceToHost, c->stream));
    HIPC(c, stream_wait(c->stream));
    return 0;
}

int smcn_moment_sums_of(smcn_ctx* c, const double* v, int Dc, const double* shift, double* out) {
    CHECK_CTX(c);
    if (!v || !out || Dc < 1 || Dc > c->D * c->D + c->D + 8) FAIL(c, "smcn_moment_sums_of: bad arguments");
    HIPC(c, hipSetDevice(c->device));
    const int64_t N = c->N, n = N * Dc;
    int rc = ensure_stage(c, n);
    if (rc) return rc;
    if ((rc = ensure_stage2(c, n))) return rc;
    HIPC(c, hipMemcpyAsync(c->stage, v, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    transpose_kernel<<<grid_for(n, 256), 256, 0, c->stream>>>(c->stage, c->stage2, N, Dc);   // -> [Dc][N]
    const int g = red_grid(N);
    double* dshift = nullptr;
    if (shift) {
        dshift = c->scal + 16;
        HIPC(c, hipMemcpyAsync(dshift, shift, sizeof(double) * Dc, hipMemcpyHostToDevice, c->stream));
    }
    moment_partial_kernel<<<g, kRedBlock, 0, c->stream>>>(c->stage2, c->wn, N, Dc, SMCN_MODEL_HOST, dshift, c->part);
    sum_final_kernel<<<final_grid(Dc), kRedBlock, 0, c->stream>>>(c->part, g, Dc, c->scal + 16 + Dc);
    HIPC(c, hipGetLastError());
    HIPC(c, hipMemcpyAsync(out, c->scal + 16 + Dc, sizeof(double) * Dc, hipMemcpyDeviceToHost, c->stream));
    HIPC(c, stream_wait(c->stream));
    return 0;
}

// ---- resampling -------------------------------------------------------------------------------
// forward-L re-weighting and the acceptance count of the device-resident loop: from the NUTS kernel's own per-particle
// |r|^2, |r'|^2 and moved flags when it wrote them (wave-per-particle kernels), else by passes over r, r', x, x'
static void enqueue_reweight_forward(smcn_ctx* c) {
    const int64_t N = c->N;
    if (c->kin_valid)
        reweight_kin_kernel<<<grid_for(N, 256), 256, 0, c->stream>>>(c->logw, c->lpri0, c->llik0, c->lpri1, c->llik1, c->kin0,
                                                                     c->kin1, c->logw_new, N, c->D);
    else {
        (void)momentum_dn(c);
        reweight_kernel<<<grid_for(N, 256), 256, 0, c->stream>>>(c->logw, c->lpri0, c->llik0, c->lpri1, c->llik1, c->r,
                                                                 c->r_new, nullptr, nullptr, c->logw_new, N, c->D);
    }
}
static void enqueue_moved_count(smcn_ctx* c, int g, double* part) {
    if (c->kin_valid) isum_partial_kernel<<<g, kRedBlock, 0, c->stream>>>(c->moved_i, c->N, part);
    else moved_partial_kernel<<<g, kRedBlock, 0, c->stream>>>(c->x, c->x_new, c->N, c->D, part);
}
// second stage of a resampling for wide particles: rows gathered one at a time, dealt to the XCDs (gather_rows_kernel)
static void enqueue_gather_rows(smcn_ctx* c, const double* flag, const int64_t* idx, int64_t n_src_total,
                                int64_t n_src_local, const double* x, double* x_out) {
    const int nchunks = grid_for(c->N, 256 * kGatherPerThread);
    const int64_t blocks = (int64_t)8 * ((c->D + 7) / 8) * nchunks;
    gather_rows_kernel<<<(unsigned)blocks, 256, 0, c->stream>>>(flag, idx, n_src_total, n_src_local, x, x_out, c->N, c->D,
                                                                 nchunks);
}
// Samples._resample decided on the device (ss[SS_FLAG]): scan, search, gather into x_tmp, copy back
static void enqueue_resample_if(smcn_ctx* c, const double* u, uint32_t iter) {
    const int64_t N = c->N;
    const int nt = grid_for(N, kScanTile);
    const int wide = c->D >= kGatherRowsMinD;
    const bool fused = nt <= kFusedOffsetsMaxTiles;      // the search kernel sums the tile totals itself: two launches
    scan_tile_if_kernel<<<nt, 256, 0, c->stream>>>(c->ss, c->wn, N, c->scan_local, c->ttot);
    if (!fused) scan_offsets_if_kernel<<<1, 64, 0, c->stream>>>(c->ss, c->ttot, nt, c->toff);
    search_gather_if_kernel<<<grid_for(N, 256), 256, 0, c->stream>>>(c->ss, c->scan_local, c->toff, N, u, c->seed, iter,
                                                                     c->base, c->x, c->x_tmp, c->D, c->logw,
                                                                     c->resample_scheme, wide ? c->idx : nullptr, !wide,
                                                                     fused ? c->ttot : nullptr, nt);
    if (wide) enqueue_gather_rows(c, c->ss + SS_FLAG, c->idx, N, N, c->x, c->x_tmp);
    copy_if_kernel<<<grid_for(N * c->D, 256), 256, 0, c->stream>>>(c->ss, c->x_tmp, c->x, N * c->D);
}

int smcn_resample_multinomial(smcn_ctx* c, const double* u, double loglik, double log_n_total, int64_t iteration,
                              int64_t* idx_out) {
    CHECK_CTX(c);
    Range roctx_range("smcn:resample");
    HIPC(c, hipSetDevice(c->device));
    const int64_t N = c->N;
    const int nt = grid_for(N, kScanTile);
    double* du = nullptr;
    if (u) {
        HIPC(c, hipMemcpyAsync(c->work, u, sizeof(double) * N, hipMemcpyHostToDevice, c->stream));
        du = c->work;
    }
    const bool fused = nt <= kFusedOffsetsMaxTiles;      // the search kernel sums the tile totals itself: two launches
    scan_tile_kernel<<<nt, 256, 0, c->stream>>>(c->wn, N, c->scan_local, c->ttot);
    if (!fused) scan_offsets_kernel<<<1, 64, 0, c->stream>>>(c->ttot, nt, c->toff);
    const int wide = c->D >= kGatherRowsMinD;
    search_gather_kernel<<<grid_for(N, 256), 256, 0, c->stream>>>(c->scan_local, c->toff, nt, N, du, c->seed,
                                                                  (uint32_t)iteration, c->base, c->x, c->x_tmp, c->D,
                                                                  c->logw, loglik - log_n_total,
                                                                  (idx_out || wide) ? c->idx : nullptr,
                                                                  c->resample_scheme, !wide, fused ? c->ttot : nullptr);
    if (wide) enqueue_gather_rows(c, nullptr, c->idx, N, N, c->x, c->x_tmp);
    HIPC(c, hipGetLastError());
    std::swap(c->x, c->x_tmp);
    if (idx_out) HIPC(c, hipMemcpyAsync(idx_out, c->idx, sizeof(int64_t) * N, hipMemcpyDeviceToHost, c->stream));
    HIPC(c, stream_wait(c->stream));
    return 0;
}

// Measurement aid (bench.py, roofline of the resampling kernels): `reps` times the three kernels of
// Samples._resample (samples.py:124-146: blocked scan of wn, tile offsets, search + gather) on the
// resident weights and particles, timed with HIP events on this context's stream.  The particle
// state afterwards is that of `reps` successive resamplings (weights are not renormalised in between:
// the work per repetition is identical, the result is not a sample of anything).
int smcn_bench_resample(smcn_ctx* c, int reps, int64_t iteration, double* ms_total) {
    CHECK_CTX(c);
    if (reps < 1 || !ms_total) FAIL(c, "smcn_bench_resample: bad arguments");
    const int64_t N = c->N;
    const int nt = grid_for(N, kScanTile);
    const bool fused = nt <= kFusedOffsetsMaxTiles;
    // the particles are permuted on a COPY (x_tmp <-> stage): the resident state is the same before and after
    int rc = ensure_stage(c, N * c->D);
    if (rc) return rc;
    struct Events {     // destroyed on every exit path
        hipEvent_t e0 = nullptr, e1 = nullptr;
        ~Events() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
    } ev;
    HIPC(c, hipEventCreate(&ev.e0));
    HIPC(c, hipEventCreate(&ev.e1));
    HIPC(c, hipMemcpyAsync(c->x_tmp, c->x, sizeof(double) * N * c->D, hipMemcpyDeviceToDevice, c->stream));
    double *src = c->x_tmp, *dst = c->stage;
    HIPC(c, hipEventRecord(ev.e0, c->stream));
    for (int r = 0; r < reps; ++r) {
        scan_tile_kernel<<<nt, 256, 0, c->stream>>>(c->wn, N, c->scan_local, c->ttot);
        if (!fused) scan_offsets_kernel<<<1, 64, 0, c->stream>>>(c->ttot, nt, c->toff);
        const int wide = c->D >= kGatherRowsMinD;
        search_gather_kernel<<<grid_for(N, 256), 256, 0, c->stream>>>(c->scan_local, c->toff, nt, N, nullptr, c->seed,
                                                                      (uint32_t)(iteration + r), c->base, src, dst, c->D,
                                                                      c->work, 0.0, c->idx, c->resample_scheme, !wide,
                                                                      fused ? c->ttot : nullptr);
        if (wide) enqueue_gather_rows(c, nullptr, c->idx, N, N, src, dst);
        std::swap(src, dst);
    }
    HIPC(c, hipGetLastError());
    HIPC(c, hipEventRecord(ev.e1, c->stream));
    HIPC(c, hipEventSynchronize(ev.e1));
    float ms = 0.f;
    HIPC(c, hipEventElapsedTime(&ms, ev.e0, ev.e1));
    *ms_total = ms;
    return 0;
}

// ---- NUTS ---------------------------------------------------------------------------------------
}  // extern "C"
// the model that finishes the trees a two-phase launch parks: the same particle on more lanes where there is such a functor
template <class M>
struct resume_model { using type = M; };
template <int NOBS, int C_, int RED, int LEVELS, bool FAST>
struct resume_model<PrmwcdDistModel<8, NOBS, C_, RED, LEVELS, FAST>> { using type = PrmwcdDistModel<64, NOBS, C_, 2, 5, FAST>; };
template <int NOBS, int C_, int RED, int LEVELS, bool FAST, int WAVES>
struct resume_model<PrmwcdDistModel<4, NOBS, C_, RED, LEVELS, FAST, WAVES>> { using type = PrmwcdDistModel<64, NOBS, C_, 2, 5, FAST>; };
#ifdef SMCN_VARIANTS
template <int NOBS, int C_, int LEVELS, bool LK>
struct resume_model<PrmwcdLaneModel<NOBS, C_, LEVELS, LK>> { using type = PrmwcdDistModel<64, NOBS, C_, 2, 5, true>; };
// one lane per particle in a kernel of its own (smcn_nuts_lane.hpp)
template <class Model>
static int launch_nuts_lane(smcn_ctx* c, NutsArgs a, int64_t items) {
    const size_t lds = sizeof(double) * ((size_t)kNutsBlock * lane_lds_doubles(Model::DL) + ((Model::SHARED + 1) & ~1));
    const void* kern = (const void*)nuts_lane_kernel<Model>;
    HIPC(c, hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int64_t blocks = (items + kNutsBlock - 1) / kNutsBlock;
    if (blocks > c->num_cu) blocks = c->num_cu;
    const int64_t need = blocks * kNutsBlock * (int64_t)lane_hbm_doubles(Model::DL);
    if (need > c->nuts_scratch_len) {
        HIPC(c, stream_wait(c->stream));
        if (c->nuts_scratch) (void)cached_free(c->nuts_scratch);
        c->nuts_scratch = nullptr;
        HIPC(c, dalloc(&c->nuts_scratch, need));
        c->nuts_scratch_len = need;
    }
    a.scratch = c->nuts_scratch;
    c->kin_valid = false;
    a.kin0 = nullptr; a.kin1 = nullptr; a.moved = nullptr;
    const int k = c->ev_n < kTimerRing ? c->ev_n : -1;
    if (k >= 0) HIPC(c, hipEventRecord(c->ev0[k], c->stream));
    nuts_lane_kernel<Model><<<(int)blocks, kNutsBlock, lds, c->stream>>>(a);
    HIPC(c, hipGetLastError());
    if (k >= 0) {
        HIPC(c, hipEventRecord(c->ev1[k], c->stream));
        c->ev_n++;
    }
    return 0;
}
#endif

template <class Model, bool TP = false>
static int launch_nuts_phase(smcn_ctx* c, Model, NutsArgs a, int64_t items);

// A/B builds: SMCN_GAUSS_OLD_KERNEL=1 runs Gaussians of 65..512 dimensions in the generic kernel (round 4)
static bool wave_old_kernel() {
#ifdef SMCN_VARIANTS
    static const bool v = getenv("SMCN_GAUSS_OLD_KERNEL") && atoi(getenv("SMCN_GAUSS_OLD_KERNEL")) != 0;
    return v;
#else
    return false;
#endif
}
// A/B builds: SMCN_FIN_OLD=1 finishes parked trees with the generic kernel's wave-per-particle instantiation (round 4)
static bool fin_old_kernel() {
#ifdef SMCN_VARIANTS
    static const bool v = getenv("SMCN_FIN_OLD") && atoi(getenv("SMCN_FIN_OLD")) != 0;
    return v;
#else
    return false;
#endif
}

// One wavefront per PRMwCD tree (smcn_nuts_fin.hpp): the parked trees of a two-phase launch, or -- widen = 2 -- every tree.
template <class Model>
static int launch_nuts_fin(smcn_ctx* c, NutsArgs a, int64_t items) {
    constexpr int wpb = kNutsBlock / 64;
    const size_t lds = sizeof(double) * ((size_t)((Model::SHARED + 1) & ~1) + (size_t)wpb * fin_lds_doubles());
    const void* kern = (const void*)nuts_fin_kernel<Model>;
    HIPC(c, hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));   // per device: on every launch
    int per_cu = 0;
    HIPC(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, nuts_fin_kernel<Model>, kNutsBlock, lds));
    if (per_cu < 1) FAIL(c, "nuts finisher kernel does not fit on a CU");
    if (const char* e = getenv("SMCN_FIN_BLOCKS_PER_CU")) {   // tuning knob
        const int v = atoi(e);
        if (v >= 1 && v < per_cu) per_cu = v;
    }
    int64_t blocks = (items + wpb - 1) / wpb;
    const int64_t cap = (int64_t)c->num_cu * per_cu;
    if (blocks > cap) blocks = cap;
    c->kin_valid = false;
    if (a.resume_in) {
        a.kin0 = nullptr; a.kin1 = nullptr; a.moved = nullptr;   // (a resumed tree's start statistics were not kept)
    } else {
        if (!c->kin0) {
            HIPC(c, dalloc(&c->kin0, c->N));
            HIPC(c, dalloc(&c->kin1, c->N));
            HIPC(c, cached_malloc((void**)&c->moved_i, sizeof(int32_t) * c->N));
        }
        a.kin0 = c->kin0; a.kin1 = c->kin1; a.moved = c->moved_i;
        c->kin_valid = true;
    }
    HIPC(c, hipMemsetAsync(c->queue, 0, sizeof(unsigned int) * 16, c->stream));
    const int k = c->ev_n < kTimerRing ? c->ev_n : -1;
    if (k >= 0) HIPC(c, hipEventRecord(c->ev0[k], c->stream));
    nuts_fin_kernel<Model><<<(int)blocks, kNutsBlock, lds, c->stream>>>(a);
    HIPC(c, hipGetLastError());
    if (k >= 0) {
        HIPC(c, hipEventRecord(c->ev1[k], c->stream));
        c->ev_n++;
    }
    return 0;
}

// One wavefront per particle, candidates by leaf index (smcn_nuts_wave.hpp): Gaussians of 65..512 dimensions.
template <class Model, bool FULL, bool HAS, int SLOTS, int WAVES>
static int launch_nuts_wave_t(smcn_ctx* c, NutsArgs a) {
    constexpr int wpb = kNutsBlock / 64;
    const size_t lds = sizeof(double) * (size_t)wpb * SLOTS * wave_slot_doubles(Model::DL);
    const void* kern = (const void*)nuts_wave_kernel<Model, FULL, HAS, SLOTS, WAVES>;
    HIPC(c, hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));   // per device: on every launch
    int per_cu = 0;
    HIPC(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, nuts_wave_kernel<Model, FULL, HAS, SLOTS, WAVES>, kNutsBlock, lds));
    if (per_cu < 1) FAIL(c, "nuts wave kernel does not fit on a CU");
    if (const char* e = getenv("SMCN_NUTS_BLOCKS_PER_CU")) {   // tuning knob
        const int v = atoi(e);
        if (v >= 1 && v < per_cu) per_cu = v;
    }
    int64_t blocks = (a.N + wpb - 1) / wpb;
    const int64_t cap = (int64_t)c->num_cu * per_cu;
    if (blocks > cap) blocks = cap;
    // first leaves of the sub-trees above the LDS slots: one area per resident wavefront
    const int64_t need = blocks * wpb * (int64_t)kMaxLevels * wave_slot_doubles(Model::DL);
    if (need > c->nuts_scratch_len) {
        HIPC(c, stream_wait(c->stream));
        if (c->nuts_scratch) (void)cached_free(c->nuts_scratch);
        c->nuts_scratch = nullptr;
        HIPC(c, dalloc(&c->nuts_scratch, need));
        c->nuts_scratch_len = need;
    }
    a.scratch = c->nuts_scratch;
    if (!c->kin0) {
        HIPC(c, dalloc(&c->kin0, c->N));
        HIPC(c, dalloc(&c->kin1, c->N));
        HIPC(c, cached_malloc((void**)&c->moved_i, sizeof(int32_t) * c->N));
    }
    a.kin0 = c->kin0; a.kin1 = c->kin1; a.moved = c->moved_i;
    c->kin_valid = true;
    HIPC(c, hipMemsetAsync(c->queue, 0, sizeof(unsigned int) * 16, c->stream));
    const int k = c->ev_n < kTimerRing ? c->ev_n : -1;
    if (k >= 0) HIPC(c, hipEventRecord(c->ev0[k], c->stream));
    nuts_wave_kernel<Model, FULL, HAS, SLOTS, WAVES><<<(int)blocks, kNutsBlock, lds, c->stream>>>(a);
    HIPC(c, hipGetLastError());
    if (k >= 0) {
        HIPC(c, hipEventRecord(c->ev1[k], c->stream));
        c->ev_n++;
    }
    return 0;
}
template <class Model, int SLOTS, int WAVES>
static int launch_nuts_wave_sw(smcn_ctx* c, NutsArgs a) {
    const bool full = c->D == 64 * Model::DL, has = c->mdata_h[2] != 0.0;
    if (full) return has ? launch_nuts_wave_t<Model, true, true, SLOTS, WAVES>(c, a) : launch_nuts_wave_t<Model, true, false, SLOTS, WAVES>(c, a);
    return has ? launch_nuts_wave_t<Model, false, true, SLOTS, WAVES>(c, a) : launch_nuts_wave_t<Model, false, false, SLOTS, WAVES>(c, a);
}
template <class Model>
static int launch_nuts_wave(smcn_ctx* c, NutsArgs a) {
    if constexpr (Model::DL == 4) {
        // three wavefronts per SIMD (<= 168 VGPRs), three LDS slots of 4 KB each per wavefront = 144 KB per CU: the third
        // wavefront hides what the round trips at a tree's two ends and the HBM stack levels cost the other two
        // (profiles/r05_c5_ab.txt: 1.74 -> 2.02 G leapfrog/s at step 0.1 against two wavefronts with four slots)
#ifdef SMCN_VARIANTS
        static const int cfg = getenv("SMCN_WAVE_CFG") ? atoi(getenv("SMCN_WAVE_CFG")) : 0;    // A/B: slots x wavefronts
        if (cfg == 42) return launch_nuts_wave_sw<Model, 4, 2>(c, a);
        if (cfg == 32) return launch_nuts_wave_sw<Model, 3, 2>(c, a);
#endif
        // (the masked / with-likelihood instantiations need a few registers more: two wavefronts, four slots, no scratch)
        if (c->D == 64 * Model::DL && c->mdata_h[2] == 0.0) return launch_nuts_wave_t<Model, true, false, 3, 3>(c, a);
        return launch_nuts_wave_sw<Model, 4, 2>(c, a);
    } else {
        return launch_nuts_wave_sw<Model, kWaveLdsSlots, Model::MIN_WAVES>(c, a);
    }
}

template <class Model>
static int launch_nuts(smcn_ctx* c, Model, NutsArgs a) {
    if constexpr (model_wave_kernel<Model>::value) {
        c->nuts_parked = 0;
        if (!wave_old_kernel()) return launch_nuts_wave<Model>(c, a);
    }
    constexpr int VS0 = Model::DIST ? Model::G * Model::DL : Model::DL;
    constexpr bool HBM0 = model_hybrid_always<Model>::value ||
                          sizeof(double) * (size_t)(kNutsBlock / Model::G) * nuts_slot_doubles(VS0) > 150 * 1024;
    // (nuts_kernel's REGE_K: edges in registers, park and take up; or PARK_SLOT: one lane per particle, park only)
    constexpr bool REGE0 = HBM0 && Model::DIST && model_two_phase<Model>::value && (Model::DL <= 4 || Model::G == 1);
    c->nuts_parked = 0;
    a.step_align = Model::G < 64 ? model_step_align<Model>::value : 1;
    using Model2 = typename resume_model<Model>::type;
    if constexpr (REGE0 && !std::is_same<Model2, Model>::value) {
        // widen == 2 (tests, A/B): EVERY tree from its start in the kernel instantiation that otherwise finishes the parked
        // ones -- the finisher's functor, its LDS and HBM stack levels -- so that the parity tests reach it on whole trees
        if (c->nuts_wide2 == 2) {
            a.step_align = 1;
            if constexpr (model_fin_kernel<Model2>::value) {
                if (!fin_old_kernel()) return launch_nuts_fin<Model2>(c, a, a.N);
            }
            return launch_nuts_phase<Model2, REGE0>(c, Model2{}, a, a.N);
        }
    }
    if (!REGE0 || c->nuts_jcap <= 0 || c->nuts_jcap >= a.max_depth + 1) return launch_nuts_phase<Model, false>(c, Model{}, a, a.N);
    // ---- two phases: trees that want more than jcap doublings are parked and finished by a second launch ----------
    const int64_t rsz = 8 * (int64_t)c->D + 8;
    if (!c->nuts_resume) {
        HIPC(c, dalloc(&c->nuts_resume, c->N * rsz));
        HIPC(c, cached_malloc((void**)&c->nuts_pend, sizeof(unsigned int) * (c->N + 1)));
    }
    HIPC(c, hipMemsetAsync(c->nuts_pend, 0, sizeof(unsigned int), c->stream));
    a.jcap = c->nuts_jcap; a.resume = c->nuts_resume; a.pend = c->nuts_pend; a.resume_in = 0;
    if constexpr (Model::G < 64) {
        // an inner park level, taken up again by this launch's own groups (smcn_set_nuts_requeue; NutsArgs::mq)
        static const int mq_env = getenv("SMCN_NUTS_REQUEUE") ? atoi(getenv("SMCN_NUTS_REQUEUE")) : -1;   // (A/B: overrides the setting)
        const int mq_b = mq_env >= 0 ? mq_env : c->nuts_mq_b;
        c->nuts_mq_used = false;
        if (mq_b > 0 && mq_b < c->nuts_jcap && c->nuts_mq_ok) {
            if (8 * c->D + 8 > 128) FAIL(c, "nuts: the inner park level holds records of up to 128 doubles (D <= 15)");
            if (!c->nuts_mq) {
                HIPC(c, cached_malloc((void**)&c->nuts_mq, sizeof(unsigned int) * (c->N + 16)));
                HIPC(c, dalloc(&c->nuts_mq_rec, c->N * 128));
            }
            HIPC(c, hipMemsetAsync(c->nuts_mq, 0, sizeof(unsigned int) * (c->N + 16), c->stream));
            a.mq = c->nuts_mq; a.mq_rec = c->nuts_mq_rec; a.mq_b = mq_b;
            c->nuts_mq_used = true;
        }
    }
    int rc = launch_nuts_phase<Model, REGE0>(c, Model{}, a, a.N);
    if (rc) return rc;
    if (a.mq && getenv("SMCN_MQ_DEBUG")) {      // diagnostics of the inner level: allocated, taken, entries never seen, unclaimed
        unsigned int h[8];
        HIPC(c, hipMemcpyAsync(h, a.mq, sizeof(h), hipMemcpyDeviceToHost, c->stream));
        HIPC(c, stream_wait(c->stream));
        fprintf(stderr, "[smcn] requeue level %d: parked %u taken %u lost %u unclaimed %d\n", a.mq_b, h[0], h[1], h[2], (int)h[5]);
    }
    if (c->nuts_wide2 && !std::is_same<Model2, Model>::value) {
        if constexpr (model_fin_kernel<Model2>::value) {
            if (!fin_old_kernel()) {
                // the finisher takes its trees from the device-side list (count included): nothing to wait for -- a full grid is
                // launched and wavefronts without a ticket leave at once; smcn_nuts_parked reads the count when asked
                c->nuts_parked = -1;
                a.jcap = 0; a.resume_in = 1; a.mq = nullptr; a.mq_b = 0;
                return launch_nuts_fin<Model2>(c, a, a.N);
            }
        }
    }
    unsigned int parked = 0;
    HIPC(c, hipMemcpyAsync(&parked, c->nuts_pend, sizeof(unsigned int), hipMemcpyDeviceToHost, c->stream));
    HIPC(c, stream_wait(c->stream));
    c->nuts_parked = parked;
    if (parked == 0) return 0;
    a.jcap = 0; a.resume_in = 1; a.mq = nullptr; a.mq_b = 0;
    if (c->nuts_wide2 && !std::is_same<Model2, Model>::value) return launch_nuts_phase<Model2, REGE0>(c, Model2{}, a, (int64_t)parked);
    if constexpr (Model::G == 1) FAIL(c, "two-phase launches of the one-lane-per-particle kernel need the finisher (widen != 0)");
    return launch_nuts_phase<Model, REGE0>(c, Model{}, a, (int64_t)parked);
}

template <class Model, bool TP>
static int launch_nuts_phase(smcn_ctx* c, Model, NutsArgs a, int64_t items) {
#ifdef SMCN_VARIANTS
    if constexpr (model_lane_kernel<Model>::value) return launch_nuts_lane<Model>(c, a, items);
#endif
    constexpr int G = Model::G;
    constexpr int VS = Model::DIST ? G * Model::DL : Model::DL;
    constexpr int gpb = kNutsBlock / G;
    constexpr bool HBM = model_hybrid_always<Model>::value ||
                         sizeof(double) * (size_t)gpb * nuts_slot_doubles(VS) > 150 * 1024;   // does not fit LDS
    const size_t lds = HBM ? sizeof(double) * ((size_t)gpb * nuts_hybrid_lds_doubles(VS, Model::LDS_LEVELS) +
                                               ((Model::SHARED + 1) & ~1))
                           : sizeof(double) * ((size_t)gpb * nuts_slot_doubles(VS) + ((Model::SHARED + 1) & ~1));
    const void* kern = (const void*)nuts_kernel<Model, HBM, TP>;
    // per DEVICE attribute (a process may hold contexts on several devices): set on every launch
    if (lds > 0) HIPC(c, hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = 0;
    HIPC(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, nuts_kernel<Model, HBM, TP>, kNutsBlock, lds));
    if (per_cu < 1) FAIL(c, "nuts kernel does not fit on a CU");
    if (const char* e = getenv("SMCN_NUTS_BLOCKS_PER_CU")) {   // tuning knob
        const int v = atoi(e);
        if (v >= 1 && v < per_cu) per_cu = v;
    }
    int64_t blocks = (items + gpb - 1) / gpb;
    const int64_t cap = (int64_t)c->num_cu * per_cu;
    if (blocks > cap) blocks = cap;
    if (HBM) {
        const int64_t need = blocks * gpb * (int64_t)nuts_slot_doubles(VS);
        if (need > c->nuts_scratch_len) {
            HIPC(c, stream_wait(c->stream));
            if (c->nuts_scratch) (void)cached_free(c->nuts_scratch);
            c->nuts_scratch = nullptr;
            HIPC(c, dalloc(&c->nuts_scratch, need));
            c->nuts_scratch_len = need;
        }
        a.scratch = c->nuts_scratch;
    }
    c->kin_valid = false;
    if (a.resume_in) {
        a.kin0 = nullptr; a.kin1 = nullptr; a.moved = nullptr;   // (a resumed tree's start statistics were not kept)
    } else if constexpr (nuts_kernel_writes_stats<Model, HBM>()) {
        if (!c->kin0) {
            HIPC(c, dalloc(&c->kin0, c->N));
            HIPC(c, dalloc(&c->kin1, c->N));
            HIPC(c, cached_malloc((void**)&c->moved_i, sizeof(int32_t) * c->N));
        }
        a.kin0 = c->kin0; a.kin1 = c->kin1; a.moved = c->moved_i;
        c->kin_valid = true;
    }
    HIPC(c, hipMemsetAsync(c->queue, 0, sizeof(unsigned int) * 16, c->stream));
    const int k = c->ev_n < kTimerRing ? c->ev_n : -1;
    if (k >= 0) HIPC(c, hipEventRecord(c->ev0[k], c->stream));
    nuts_kernel<Model, HBM, TP><<<(int)blocks, kNutsBlock, lds, c->stream>>>(a);
    HIPC(c, hipGetLastError());
    if (k >= 0) {
        HIPC(c, hipEventRecord(c->ev1[k], c->stream));
        c->ev_n++;
    }
    return 0;
}

#ifdef SMCN_VARIANTS
template <class Model, bool TAPE>
static int launch_nuts2(smcn_ctx* c, Model, Nuts2Args a, const double* tape_d, const int64_t* tape_off_d,
                        bool fuse_reweight, int B = 1, double* gen_x = nullptr, double* gen_logw = nullptr,
                        double* cnt = nullptr, int phase = 0 /* 0: all, 1: prep + kernel, 2: post */) {
    constexpr int G = Model::G, DL = Model::DL, VP = n2_vp(DL);
    constexpr int gpb = kNutsBlock / G;
    const int64_t N = c->N;
    if (N * (int64_t)B * n2_out_doubles(DL) * 8 >= (int64_t)1 << 32)
        FAIL(c, "nuts2: shard too large for 32-bit record offsets (split over more shards)");
    if (TAPE && B != 1) FAIL(c, "nuts2: recorded tapes replay one transition at a time");
    if (phase == 2 && B > c->rec_cap) FAIL(c, "nuts2: post without a launch");
    if (B > c->rec_cap) {   // once: sized for the longest block the caller announced (smcn_fuse_begin)
        const int cap = (c->fuse_max > B && (int64_t)N * c->fuse_max * n2_out_doubles(DL) * 8 < ((int64_t)1 << 32))
                            ? c->fuse_max : B;
        HIPC(c, stream_wait(c->stream));
        if (c->in_rec) (void)cached_free(c->in_rec);
        if (c->out_rec) (void)cached_free(c->out_rec);
        c->in_rec = c->out_rec = nullptr;
        HIPC(c, dalloc(&c->in_rec, N * cap * n2_in_doubles(DL)));
        HIPC(c, dalloc(&c->out_rec, N * cap * n2_out_doubles(DL)));
        c->rec_cap = cap;
    }
    constexpr int NL = Model::N2_LDS_LEVELS;
    const size_t lds = sizeof(double) * ((size_t)gpb * n2_slot_doubles(DL, NL) + ((Model::SHARED + 1) & ~1));
    HIPC(c, hipFuncSetAttribute((const void*)nuts2_kernel<Model, TAPE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds));   // per device: on every launch
    int per_cu = 0;
    HIPC(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, nuts2_kernel<Model, TAPE>, kNutsBlock, lds));
    if (per_cu < 1) FAIL(c, "nuts2 kernel does not fit on a CU");
    if (const char* e = getenv("SMCN_NUTS_BLOCKS_PER_CU")) {   // tuning knob
        const int v = atoi(e);
        if (v >= 1 && v < per_cu) per_cu = v;
    }
    int64_t blocks = (N + gpb - 1) / gpb;
    const int64_t cap = (int64_t)c->num_cu * per_cu;
    if (blocks > cap) blocks = cap;
    if (NL < 10) {   // overflow tree-stack levels, one area per resident group
        const int64_t need = blocks * gpb * (int64_t)n2_ovf_doubles(DL, NL);
        if (need > c->n2_ovf_len) {
            HIPC(c, stream_wait(c->stream));
            if (c->n2_ovf) (void)cached_free(c->n2_ovf);
            c->n2_ovf = nullptr;
            HIPC(c, dalloc(&c->n2_ovf, need));
            c->n2_ovf_len = need;
        }
        a.ovf = c->n2_ovf;
    }
    if (phase != 2) {
    // momentum draw + slice exponential + input records (samples.py:155, nuts.py:69)
    if (c->momentum_set && B != 1) FAIL(c, "nuts2: caller-supplied momenta go with single transitions");
    nuts2_prep_kernel<<<grid_for(N * B, 256), 256, sizeof(double) * 256 * n2_in_doubles(DL), c->stream>>>(c->x, c->momentum_set ? c->r : nullptr, c->r,
                                                                   c->in_rec, N, c->D, VP, c->base, c->seed, a.iter,
                                                                   B, tape_d, tape_off_d);
    c->momentum_set = false;
    HIPC(c, hipMemsetAsync(c->queue, 0, sizeof(unsigned int) * 4, c->stream));
    a.in = c->in_rec;
    a.out = c->out_rec;
    a.B = B;
    const int k = c->ev_n < kTimerRing ? c->ev_n : -1;
    if (k >= 0) HIPC(c, hipEventRecord(c->ev0[k], c->stream));
    nuts2_kernel<Model, TAPE><<<(int)blocks, kNutsBlock, lds, c->stream>>>(a);
    HIPC(c, hipGetLastError());
    if (k >= 0) {
        HIPC(c, hipEventRecord(c->ev1[k], c->stream));
        c->ev_n++;
    }
    }
    if (phase == 1) return 0;
    nuts2_post_kernel<0><<<grid_for(N, 256), 256, 0, c->stream>>>(
        c->out_rec, c->in_rec, c->x, fuse_reweight ? c->logw : nullptr, c->x_new, c->r_new, c->lpri0, c->llik0,
        c->lpri1, c->llik1, c->nleap, c->depth, c->ndraws, c->flags, fuse_reweight ? c->logw_new : nullptr, gen_x,
        gen_logw, cnt, N, c->D, VP, B);
    HIPC(c, hipGetLastError());
    return 0;
}

#endif

// One lane per particle (smcn_nuts3.hpp): one wavefront per block, no work queue; the record buffers,
// the prep kernel (momentum draw, slice exponential) and the post kernel (unpack + forward-L re-weight)
// are those of the v2 kernel.
template <class Model, bool TAPE, int LC, int LF, int WPE = 1>
static int launch_nuts3(smcn_ctx* c, Nuts2Args a, const double* tape_d, const int64_t* tape_off_d, bool fuse_reweight,
                        int B, double* gen_x, double* gen_logw, double* cnt, int phase) {
    constexpr int D = Model::D, VP = n2_vp(D);
    const int64_t N = c->N;
    if (TAPE && B != 1) FAIL(c, "nuts3: recorded tapes replay one transition at a time");
    if (phase == 2 && B > c->rec_cap) FAIL(c, "nuts3: post without a launch");
    if (B > c->rec_cap) {   // once: sized for the longest block the caller announced (smcn_fuse_begin)
        const int cap = c->fuse_max > B ? c->fuse_max : B;
        HIPC(c, stream_wait(c->stream));
        if (c->in_rec) (void)cached_free(c->in_rec);
        if (c->out_rec) (void)cached_free(c->out_rec);
        c->in_rec = c->out_rec = nullptr;
        HIPC(c, dalloc(&c->in_rec, N * cap * n2_in_doubles(D)));
        HIPC(c, dalloc(&c->out_rec, N * cap * n2_out_doubles(D)));
        c->rec_cap = cap;
    }
    const size_t lds = (size_t)n3_lds_bytes<Model>(LC, LF);
    // one wavefront per SIMD is all the chip holds of this kernel: larger populations are NOT more wavefronts (a second
    // round of them would last a whole longest chain again) -- the resident lanes take the particles beyond 64 * blocks
    // from a queue as they finish their own (smcn_nuts3.hpp)
    int64_t blocks = (N + kN3Block - 1) / kN3Block;
    const int64_t resident = (int64_t)c->num_cu * 4 * WPE;
    if (c->lane_grid_cap > 0 && blocks > c->lane_grid_cap) blocks = c->lane_grid_cap;
    else if (c->lane_grid_cap == 0 && blocks > resident) blocks = resident;
    // (the ready bits of a wavefront's particles are one 64-bit word per lane: beyond 4 096 particles a wavefront the
    //  launch is the plain one -- a wavefront per 64 particles, in rounds)
    if (((N + blocks - 1) / blocks + 63) / 64 > kN3ReadyWords) blocks = (N + kN3Block - 1) / kN3Block;
    const int64_t need = blocks * kN3Block * 2 * (int64_t)n3_ovf_pairs(D, LC, LF);   // doubles
    if (need > c->n2_ovf_len) {
        HIPC(c, stream_wait(c->stream));
        if (c->n2_ovf) (void)cached_free(c->n2_ovf);
        c->n2_ovf = nullptr;
        HIPC(c, dalloc(&c->n2_ovf, need));
        c->n2_ovf_len = need;
    }
    a.ovf = c->n2_ovf;
    if (phase != 2) {
        if (c->momentum_set && B != 1) FAIL(c, "nuts3: caller-supplied momenta go with single transitions");
        nuts3_prep_kernel<D><<<grid_for(N * B, 256), 256, 0, c->stream>>>(c->x, c->momentum_set ? c->r : nullptr, c->r,
                                                                         c->in_rec, N, c->base, c->seed, a.iter, B, tape_d,
                                                                         tape_off_d);
        c->momentum_set = false;
        a.in = c->in_rec;
        a.out = c->out_rec;
        a.B = B;
        a.logw0 = (fuse_reweight && B > 1) ? c->logw : nullptr;   // compact records for the transitions before the last
        a.wide = (c->wide_eval ? 1 : 0) | 2;    // the helper draws do not touch the bits: on in both modes
        const bool queued = blocks * kN3Block < N;     // fewer lanes than particles: every wavefront works through a run of them
        a.seg_len = 0;
        a.seg_tail = 0;
        a.seg_align = 4;     // (measured at N = 131 072: 2 / 4 / 8 gave 7.57 / 7.95 / 7.63 G leapfrog/s; no such rule 6.75)
        if (queued) {
            // A block is handed on in SEGMENTS (a quarter of it each): a wavefront's makespan is (its particles' work / 64) +
            // (about one job), and a job is then a quarter of a particle's chain.  (Measured, B = 20: 4 or 5 segments beat
            // 1, 2, 10 and 20 -- profiles/r04_n_sweep.md.)  The ready bits of [segments][particles of a wavefront] are one
            // 64-bit word per lane: fewer segments for very large populations.
            const int64_t per_wave = (N + blocks - 1) / blocks, words = (per_wave + 63) / 64;
            int nseg = c->lane_segments > 0 ? c->lane_segments : 4;
            if (nseg > B) nseg = (int)B;
            while (nseg > 1 && nseg * words > kN3ReadyWords) --nseg;
            a.seg_tail = 0;
            if (nseg > 1) a.seg_len = (int)((B + nseg - 1) / nseg);
            int64_t segs = a.seg_len > 0 ? (B + a.seg_len - 1) / a.seg_len : 1;
            // ... and the last segment once more, 2/5 of it before the block's end (5 5 5 3 2 for 20 transitions): a wavefront
            // ends on short jobs
            const int64_t last_len = B - (segs - 1) * a.seg_len;
            // (measured at N = 131 072, 20 transitions: no cut 7.47, the last 1 / 2 / 3 transitions 7.62 / 7.69 / 7.57 G leapfrog/s)
            if (segs > 1 && last_len >= 2 && (segs + 1) * words <= kN3ReadyWords) {
                a.seg_tail = (int)((2 * last_len + 2) / 5 > 0 ? (2 * last_len + 2) / 5 : 1);
                ++segs;
            }
            const int64_t hw = N * (segs - 1) * 2 * (n2_vp(D) / 2 + 1);     // 8-byte words: [N][segments - 1][VH + 1 pairs]
            if (hw > c->handover_len) {
                HIPC(c, stream_wait(c->stream));
                if (c->handover) (void)cached_free(c->handover);
                c->handover = nullptr;
                HIPC(c, cached_malloc((void**)&c->handover, sizeof(unsigned long long) * (size_t)hw));
                c->handover_len = hw;
            }
            a.handover = c->handover;
        }
        if (WPE > 1 && !queued) FAIL(c, "nuts3: the two-wavefront instantiation is the queue kernel's");
        const void* const kfn = (queued || WPE > 1) ? (const void*)nuts3_kernel<Model, TAPE, LC, LF, true, WPE>
                                                    : (const void*)nuts3_kernel<Model, TAPE, LC, LF, false, 1>;
        HIPC(c, hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));   // per device
        const int k = c->ev_n < kTimerRing ? c->ev_n : -1;
        if (k >= 0) HIPC(c, hipEventRecord(c->ev0[k], c->stream));
        if (queued || WPE > 1) nuts3_kernel<Model, TAPE, LC, LF, true, WPE><<<(int)blocks, kN3Block, lds, c->stream>>>(a);
        else nuts3_kernel<Model, TAPE, LC, LF, false, 1><<<(int)blocks, kN3Block, lds, c->stream>>>(a);
        HIPC(c, hipGetLastError());
        if (k >= 0) {
            HIPC(c, hipEventRecord(c->ev1[k], c->stream));
            c->ev_n++;
        }
    }
    if (phase == 1) return 0;
    nuts2_post_kernel<D><<<grid_for(N, 256), 256, 0, c->stream>>>(
        c->out_rec, c->in_rec, c->x, fuse_reweight ? c->logw : nullptr, c->x_new, c->r_new, c->lpri0, c->llik0,
        c->lpri1, c->llik1, c->nleap, c->depth, c->ndraws, c->flags, fuse_reweight ? c->logw_new : nullptr, gen_x,
        gen_logw, cnt, N, c->D, VP, B, (fuse_reweight && B > 1) ? 1 : 0, 1);
    HIPC(c, hipGetLastError());
    return 0;
}

// NUTSProposal.rvs for a host-evaluated target: the tree state machine on the device, one callback
// per lock-step leapfrog (smcn_nuts_host.hpp).  Leaves x_new, r_new, density parts, tree statistics.
static int propose_host(smcn_ctx* c, double step_size, double phi, int max_depth, double delta_max, int64_t iteration,
                        const double* tape_d, const int64_t* tape_off_d) {
    const int64_t N = c->N;
    const int D = c->D;
    if (!c->hc_vec) {
        HIPC(c, dalloc(&c->hc_vec, (int64_t)HV_COUNT * D * N));
        HIPC(c, dalloc(&c->hc_sc, (int64_t)HS_COUNT * N));
        HIPC(c, dalloc(&c->hc_st, (int64_t)HI_COUNT * N));
        HIPC(c, dalloc(&c->hc_gp, (int64_t)D * N));
        HIPC(c, dalloc(&c->hc_gl, (int64_t)D * N));
    }
    if (!c->momentum_set) {  // samples.py:155 with the N(0, I) momentum proposal
        const int64_t n = N * ((D + 1) / 2);
        normals_kernel<<<grid_for(n, 256), 256, 0, c->stream>>>(c->r, N, D, c->base, c->seed, (uint32_t)iteration,
                                                               kStreamMomentum);
        HIPC(c, hipGetLastError());
    }
    c->momentum_set = false;
    nuts_host_begin_kernel<<<grid_for(N, 256), 256, 0, c->stream>>>(c->x, c->r, c->hc_vec, c->hc_st, N, D);
    HIPC(c, hipGetLastError());
    NutsHostArgs a;
    a.N = N; a.particle_base = c->base; a.D = D; a.vec = c->hc_vec; a.sc = c->hc_sc; a.st = c->hc_st;
    a.lpri = c->lpri1; a.llik = c->llik1; a.gpri = c->hc_gp; a.glik = c->hc_gl;
    a.eps = step_size; a.phi = phi; a.delta_max = delta_max; a.max_depth = max_depth;
    a.seed = c->seed; a.iter = (uint32_t)iteration; a.tape = tape_d; a.tape_off = tape_off_d; a.n_active = c->queue;
    // every tree ends after at most 2^(max_depth+1) - 1 leapfrogs (+ the initial evaluation)
    const int64_t max_rounds = ((int64_t)1 << (max_depth + 1)) + 1;
    for (int64_t round = 0; round < max_rounds; ++round) {
        // pending positions = hc_vec[HV_X]; finished particles are evaluated too (their values are ignored)
        int rc = host_eval(c, c->hc_vec + (int64_t)HV_X * D * N, true, c->lpri1, c->llik1, c->hc_gp, c->hc_gl);
        if (rc) return rc;
        HIPC(c, hipMemsetAsync(c->queue, 0, sizeof(unsigned int) * 4, c->stream));
        nuts_host_advance_kernel<<<grid_for(N, 256), 256, 0, c->stream>>>(a);
        HIPC(c, hipGetLastError());
        unsigned int active = 0;
        HIPC(c, hipMemcpyAsync(&active, c->queue, sizeof(unsigned int), hipMemcpyDeviceToHost, c->stream));
        HIPC(c, stream_wait(c->stream));
        if (active == 0) break;
    }
    nuts_host_finish_kernel<<<grid_for(N, 256), 256, 0, c->stream>>>(c->hc_vec, c->hc_sc, c->hc_st, N, D, c->x_new,
                                                                    c->r_new, c->lpri0, c->llik0, c->lpri1, c->llik1,
                                                                    c->nleap, c->depth, c->ndraws, c->flags);
    HIPC(c, hipGetLastError());
    c->lg_set = false;
    return 0;
}

static int propose_async(smcn_ctx* c, double step_size, double phi, int max_depth, double delta_max, int64_t iteration,
                         const double* tape, const int64_t* tape_off, bool fuse_reweight = false,
                         bool* reweighted = nullptr, int B = 1, double* gen_x = nullptr, double* gen_logw = nullptr,
                         double* cnt = nullptr, int phase = 0) {
    if (reweighted) *reweighted = false;
    c->kin_valid = false;       // (set again by a launch whose kernel writes the per-particle statistics)
    if (max_depth < 0 || max_depth > kMaxLevels) FAIL(c, "smcn_propose_nuts: max_depth must be in 0..10");
    if ((tape == nullptr) != (tape_off == nullptr)) FAIL(c, "smcn_propose_nuts: tape and tape_off go together");
    const int64_t N = c->N;
    const double* tape_d = nullptr;
    const int64_t* tape_off_d = nullptr;
    if (tape) {
        const int64_t len = tape_off[N];
        if (len < 0) FAIL(c, "smcn_propose_nuts: bad tape offsets");
        if (len + 1 > c->tape_cap) {
            HIPC(c, stream_wait(c->stream));
            if (c->tape_d) (void)cached_free(c->tape_d);
            c->tape_d = nullptr;
            HIPC(c, dalloc(&c->tape_d, len + 1));
            c->tape_cap = len + 1;
        }
        if (!c->tape_off_d) HIPC(c, dalloc(&c->tape_off_d, N + 1));
        HIPC(c, hipMemcpyAsync(c->tape_d, tape, sizeof(double) * len, hipMemcpyHostToDevice, c->stream));
        HIPC(c, hipMemcpyAsync(c->tape_off_d, tape_off, sizeof(int64_t) * (N + 1), hipMemcpyHostToDevice, c->stream));
        tape_d = c->tape_d;
        tape_off_d = c->tape_off_d;
    }
    if (c->model == SMCN_MODEL_HOST) {
        if (B != 1 || phase != 0) FAIL(c, "host target: one transition per call (no fused blocks)");
        if (reweighted) *reweighted = false;
        return propose_host(c, step_size, phi, max_depth, delta_max, iteration, tape_d, tape_off_d);
    }
    // arma: one lane per particle (any series length)
    if (c->model == SMCN_MODEL_ARMA && c->lane_kernel) {
        Nuts2Args b;
        b.N = N; b.particle_base = c->base; b.mdata = c->mdata; b.in = nullptr; b.out = nullptr;
        b.queue = c->queue; b.eps = step_size; b.phi = phi; b.delta_max = delta_max; b.max_depth = max_depth;
        b.seed = c->seed; b.iter = (uint32_t)iteration; b.tape = tape_d; b.tape_off = tape_off_d;
        b.prof = c->prof; b.ovf = nullptr; b.B = B;
        int rc3;
#ifdef SMCN_VARIANTS
        // A/B: two wavefronts per SIMD for populations of at least 128 particles per SIMD (SMCN_LANE_WPE=2)
        static const int wpe = getenv("SMCN_LANE_WPE") ? atoi(getenv("SMCN_LANE_WPE")) : 1;
        if (wpe == 2 && !tape_d && N > (int64_t)c->num_cu * 4 * 2 * kN3Block && c->lane_grid_cap == 0)
            rc3 = launch_nuts3<ArmaLaneModel, false, SMCN_WPE2_LC, SMCN_WPE2_LF, 2>(c, b, tape_d, tape_off_d, fuse_reweight, B, gen_x,
                                                                                   gen_logw, cnt, phase);
        else
#endif
        rc3 = tape_d ? launch_nuts3<ArmaLaneModel, true, 3, 3>(c, b, tape_d, tape_off_d, fuse_reweight, B, gen_x,
                                                               gen_logw, cnt, phase)
                     : launch_nuts3<ArmaLaneModel, false, 3, 3>(c, b, tape_d, tape_off_d, fuse_reweight, B, gen_x,
                                                                gen_logw, cnt, phase);
        if (rc3) return rc3;
        if (reweighted) *reweighted = fuse_reweight;
        c->lg_set = false;
        return 0;
    }
#ifdef SMCN_VARIANTS
    // A/B builds only: the second-generation kernel for replicated-state group functors (arma on 4 + 4 lanes
    // with SMCN_ARMA_NUTS2=1, PRMwCD replicated with SMCN_PRMWCD_DIST=0); the product runs arma in nuts3_kernel
    static const bool force_v1 = getenv("SMCN_NUTS_V1") != nullptr;
    bool used_v2 = false;
    int rc2 = with_model(c, [&](auto m) {
        using M = decltype(m);
        if constexpr (!M::DIST) {
            if (!force_v1) {
                Nuts2Args b;
                b.N = N; b.particle_base = c->base; b.mdata = c->mdata; b.in = nullptr; b.out = nullptr;
                b.queue = c->queue; b.eps = step_size; b.phi = phi; b.delta_max = delta_max; b.max_depth = max_depth;
                b.seed = c->seed; b.iter = (uint32_t)iteration; b.tape = tape_d; b.tape_off = tape_off_d;
                b.prof = c->prof;
                b.ovf = nullptr;
                used_v2 = true;
                return tape_d ? launch_nuts2<M, true>(c, m, b, tape_d, tape_off_d, fuse_reweight, B, gen_x, gen_logw, cnt,
                                                      phase)
                              : launch_nuts2<M, false>(c, m, b, tape_d, tape_off_d, fuse_reweight, B, gen_x, gen_logw,
                                                       cnt, phase);
            }
        }
        return 0;
    });
    if (rc2) return rc2;
    if (used_v2) {
        if (reweighted) *reweighted = fuse_reweight;
        c->lg_set = false;
        return 0;
    }
#endif
    if (B != 1 || phase != 0) FAIL(c, "fused transitions: this model's kernel runs one transition per launch");
    // (nuts_wave_kernel's targets take and leave the momentum one contiguous row per particle: smcn_ctx::r_pm)
    const bool wave_target = c->model == SMCN_MODEL_GAUSS && c->D > 64 && !wave_old_kernel();
    if (!c->momentum_set) {  // samples.py:155 with the N(0, I) momentum proposal
        const int64_t n = N * ((c->D + 1) / 2);
        if (wave_target)
            normals_pm_kernel<<<grid_for(n, 256), 256, 0, c->stream>>>(c->r, N, c->D, c->base, c->seed, (uint32_t)iteration,
                                                                      kStreamMomentum);
        else
            normals_kernel<<<grid_for(n, 256), 256, 0, c->stream>>>(c->r, N, c->D, c->base, c->seed, (uint32_t)iteration,
                                                                   kStreamMomentum);
        HIPC(c, hipGetLastError());
        c->r_pm = wave_target;
    }
    c->momentum_set = false;
    c->r_new_pm = wave_target;
    NutsArgs a;
    a.r_pm = c->r_pm ? 1 : 0;
    a.r_new_pm = c->r_new_pm ? 1 : 0;
    a.N = N; a.particle_base = c->base; a.mdata = c->mdata; a.x = c->x; a.r = c->r;
    a.x_new = c->x_new; a.r_new = c->r_new;
    a.lpri0 = c->lpri0; a.llik0 = c->llik0; a.lpri1 = c->lpri1; a.llik1 = c->llik1;
    a.nleap = c->nleap; a.depth = c->depth; a.ndraws = c->ndraws; a.flags = c->flags;
    a.queue = c->queue; a.eps = step_size; a.phi = phi; a.delta_max = delta_max; a.max_depth = max_depth;
    a.seed = c->seed; a.iter = (uint32_t)iteration; a.tape = nullptr; a.tape_off = nullptr;
    a.prof = c->prof;
    a.scratch = nullptr;
    a.tape = tape_d;
    a.tape_off = tape_off_d;
    int rc = with_model(c, [&](auto m) { return launch_nuts(c, m, a); });
    if (rc) return rc;
    c->lg_set = false;
    return 0;
}
extern "C" {
// Two-phase NUTS launches for the group kernels with register-resident edges (PRMwCD, Gaussians of 129..256 dimensions):
// doublings <= 0 switches them off.  widen != 0: the parked trees are finished by the wavefront-per-particle functor of
// the model where one exists (PRMwCD), else (and for widen == 0) by the kernel that parked them -- bit for bit the
// one-launch result then.  widen == 2 (tests): no parking -- every tree runs from its start in the finisher's kernel.
int smcn_set_nuts_cap(smcn_ctx* c, int doublings, int widen) {
    CHECK_CTX(c);
    c->nuts_jcap = doublings > 0 ? doublings : 0;
    c->nuts_wide2 = widen == 2 ? 2 : (widen != 0);
    return 0;
}
// ... and an INNER park level of the first launch (0 < doublings < the cap above; 0 = none): trees that want more than
// `doublings` doublings are parked once more there and taken up again by the same launch's groups when the fresh
// particles have run out -- the launch then ends on pieces of trees, not on whole ones.  Same trees, same results.
int smcn_set_nuts_requeue(smcn_ctx* c, int doublings) {
    CHECK_CTX(c);
    c->nuts_mq_b = doublings > 0 ? doublings : 0;
    return 0;
}
int smcn_nuts_parked(smcn_ctx* c, int64_t* parked) {
    CHECK_CTX(c);
    if (!parked) FAIL(c, "smcn_nuts_parked: null");
    if (c->nuts_parked < 0) {        // (the last two-phase launch left the count on the device)
        unsigned int n = 0;
        HIPC(c, hipSetDevice(c->device));
        HIPC(c, hipMemcpyAsync(&n, c->nuts_pend, sizeof(unsigned int), hipMemcpyDeviceToHost, c->stream));
        HIPC(c, stream_wait(c->stream));
        c->nuts_parked = n;
    }
    *parked = c->nuts_parked;
    return 0;
}

int smcn_propose_nuts(smcn_ctx* c, double step_size, double phi, int max_depth, double delta_max, int64_t iteration,
                      const double* tape, const int64_t* tape_off) {
    CHECK_CTX(c);
    Range roctx_range("smcn:nuts");
    c->nuts_mq_ok = true;          // (this entry point waits for the launch and checks the inner level's hand-overs)
    int rc = propose_async(c, step_size, phi, max_depth, delta_max, iteration, tape, tape_off);
    c->nuts_mq_ok = false;
    if (rc) return rc;
    if (c->nuts_mq_used) {
        // the inner park level's hand-overs: parked == taken, none lost (smcn_nuts.hpp bounds the wait for an entry so that a
        // launch always ends; a tree behind an entry never seen would be missing from the proposal)
        unsigned int h[6];
        HIPC(c, hipMemcpyAsync(h, c->nuts_mq, sizeof(h), hipMemcpyDeviceToHost, c->stream));
        HIPC(c, stream_wait(c->stream));
        if (h[2] != 0u || h[0] != h[1] || h[5] != 0u) FAIL(c, "nuts: the inner park level lost a tree (parked != taken)");
        return 0;
    }
    HIPC(c, stream_wait(c->stream));
    return 0;
}

int smcn_get_tree_stats(smcn_ctx* c, int32_t* nleap, int32_t* depth, int32_t* ndraws, int32_t* flags) {
    CHECK_CTX(c);
    int rc = 0;
    if (nleap && (rc = download_n(c, c->nleap, nleap, sizeof(int32_t)))) return rc;
    if (depth && (rc = download_n(c, c->depth, depth, sizeof(int32_t)))) return rc;
    if (ndraws && (rc = download_n(c, c->ndraws, ndraws, sizeof(int32_t)))) return rc;
    if (flags && (rc = download_n(c, c->flags, flags, sizeof(int32_t)))) return rc;
    return 0;
}

int smcn_last_leapfrogs(smcn_ctx* c, int64_t* total) {
    CHECK_CTX(c);
    if (!total) FAIL(c, "smcn_last_leapfrogs: null");
    const int g = red_grid(c->N);
    isum_partial_kernel<<<g, kRedBlock, 0, c->stream>>>(c->nleap, c->N, c->part);
    sum_final_kernel<<<1, kRedBlock, 0, c->stream>>>(c->part, g, 1, c->scal + 8);
    HIPC(c, hipGetLastError());
    double v = 0.0;
    HIPC(c, hipMemcpyAsync(&v, c->scal + 8, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPC(c, stream_wait(c->stream));
    *total = (int64_t)v;
    return 0;
}

int smcn_get_density_parts(smcn_ctx* c, double* lpri0, double* llik0, double* lpri1, double* llik1) {
    CHECK_CTX(c);
    int rc = 0;
    if (lpri0 && (rc = download_n(c, c->lpri0, lpri0, sizeof(double)))) return rc;
    if (llik0 && (rc = download_n(c, c->llik0, llik0, sizeof(double)))) return rc;
    if (lpri1 && (rc = download_n(c, c->lpri1, lpri1, sizeof(double)))) return rc;
    if (llik1 && (rc = download_n(c, c->llik1, llik1, sizeof(double)))) return rc;
    return 0;
}

// ---- reweight / commit --------------------------------------------------------------------------
int smcn_set_lkernel_values(smcn_ctx* c, const double* L, const double* q) {
    CHECK_CTX(c);
    if (L) {
        HIPC(c, hipMemcpyAsync(c->Lg, L, sizeof(double) * c->N, hipMemcpyHostToDevice, c->stream));
        c->lg_set = true;
    }
    if (q) {
        HIPC(c, hipMemcpyAsync(c->qv, q, sizeof(double) * c->N, hipMemcpyHostToDevice, c->stream));
        c->q_set = true;
    }
    HIPC(c, stream_wait(c->stream));
    return 0;
}

int smcn_reweight(smcn_ctx* c, int lkernel) {
    CHECK_CTX(c);
    Range roctx_range("smcn:reweight");
    if (lkernel != SMCN_LKERNEL_FORWARD && lkernel != SMCN_LKERNEL_GAUSSIAN) FAIL(c, "Unknown L-kernel supplied");
    if (lkernel == SMCN_LKERNEL_GAUSSIAN && !c->lg_set)
        FAIL(c, "smcn_reweight: call smcn_gauss_lkernel_logpdf first");
    { int rcm = momentum_dn(c); if (rcm) return rcm; }
    reweight_kernel<<<grid_for(c->N, 256), 256, 0, c->stream>>>(
        c->logw, c->lpri0, c->llik0, c->lpri1, c->llik1, c->r, c->r_new, c->lg_set ? c->Lg : nullptr,
        c->q_set ? c->qv : nullptr, c->logw_new, c->N, c->D);
    HIPC(c, hipGetLastError());
    HIPC(c, stream_wait(c->stream));
    c->lg_set = false;
    c->q_set = false;
    return 0;
}

int smcn_gauss_lkernel_sums(smcn_ctx* c, const double* shift, double* sums) {
    CHECK_CTX(c);
    Range roctx_range("smcn:gauss_lkernel");
    if (!shift || !sums) FAIL(c, "smcn_gauss_lkernel_sums: null");
    HIPC(c, hipSetDevice(c->device));
    const int D = c->D, E = 2 * D, nq = E + E * (E + 1) / 2;
    if (D > 64) FAIL(c, "smcn_gauss_lkernel_sums: D > 64 not supported");
    const int TP = D <= 16 ? 256 : 64;
    const size_t lds = sizeof(double) * ((size_t)E * TP + nq) + sizeof(unsigned short) * (size_t)((nq + 3) & ~3);
    HIPC(c, hipFuncSetAttribute((const void*)glk_sums_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    int64_t nb = (c->N + TP - 1) / TP;
    if (nb > kMaxPart) nb = kMaxPart;
    double* dshift = c->scal + 16;
    HIPC(c, hipMemcpyAsync(dshift, shift, sizeof(double) * E, hipMemcpyHostToDevice, c->stream));
    glk_sums_kernel<<<(int)nb, kGlkSumsBlock, lds, c->stream>>>(c->r_new, c->x_new, c->N, D, dshift, TP, c->part);
    double* dout = c->scal + 16 + E;
    sum_final_kernel<<<final_grid(nq), kRedBlock, 0, c->stream>>>(c->part, (int)nb, nq, dout);
    HIPC(c, hipGetLastError());
    HIPC(c, hipMemcpyAsync(sums, dout, sizeof(double) * nq, hipMemcpyDeviceToHost, c->stream));
    HIPC(c, stream_wait(c->stream));
    return 0;
}

int smcn_gauss_lkernel_logpdf(smcn_ctx* c, const double* mu_x, const double* m0, const double* B, const double* U,
                              double c0) {
    CHECK_CTX(c);
    Range roctx_range("smcn:gauss_lkernel");
    if (!mu_x || !m0 || !B || !U) FAIL(c, "smcn_gauss_lkernel_logpdf: null");
    HIPC(c, hipSetDevice(c->device));
    const int D = c->D;
    if (D > 32) FAIL(c, "smcn_gauss_lkernel_logpdf: D > 32 not supported");
    double* par = c->scal + 16;
    HIPC(c, hipMemcpyAsync(par, mu_x, sizeof(double) * D, hipMemcpyHostToDevice, c->stream));
    HIPC(c, hipMemcpyAsync(par + D, m0, sizeof(double) * D, hipMemcpyHostToDevice, c->stream));
    HIPC(c, hipMemcpyAsync(par + 2 * D, B, sizeof(double) * D * D, hipMemcpyHostToDevice, c->stream));
    HIPC(c, hipMemcpyAsync(par + 2 * D + D * D, U, sizeof(double) * D * D, hipMemcpyHostToDevice, c->stream));
    const size_t lds = sizeof(double) * ((size_t)2 * D + 2 * D * D + (size_t)D * 256);
    HIPC(c, hipFuncSetAttribute((const void*)glk_logpdf_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    glk_logpdf_kernel<<<grid_for(c->N, 256), 256, lds, c->stream>>>(c->r_new, c->x_new, c->N, D, par, c0, c->Lg);
    HIPC(c, hipGetLastError());
    HIPC(c, stream_wait(c->stream));
    c->lg_set = true;
    return 0;
}

// The whole Gaussian L-kernel for ONE shard without a host round trip in between: both moment passes, the D x D algebra
// (smcn_glk.hpp) and the conditional log-density, one wait at the end for the status.  info = [status, c0, cond(c_xx),
// cond(cov)]; status 0: the L values are set; 1 / 2: a matrix is not positive definite / too ill-conditioned for the
// Cholesky route -- nothing is set and the caller runs the reference's own pinv / eigh on the host
// (lkernel/gaussian_lkernel.py: smcn_gauss_lkernel_sums + smcn_gauss_lkernel_logpdf).
int smcn_gauss_lkernel_device(smcn_ctx* c, double info[4]) {
    CHECK_CTX(c);
    Range roctx_range("smcn:gauss_lkernel");
    if (!info) FAIL(c, "smcn_gauss_lkernel_device: null");
    HIPC(c, hipSetDevice(c->device));
    const int D = c->D, E = 2 * D, nq = E + E * (E + 1) / 2;
    if (D > kGlkMaxD) FAIL(c, "smcn_gauss_lkernel_device: D > 32 not supported (use the host algebra)");
    if (c->N < 2) FAIL(c, "smcn_gauss_lkernel_device: needs at least two particles");
    const size_t npar = (size_t)2 * D + 2 * D * D + 4;
    if (!c->glk_buf) HIPC(c, cached_malloc((void**)&c->glk_buf, sizeof(double) * (E + 2 * (size_t)nq + npar)));
    double *dmu = c->glk_buf, *ds1 = dmu + E, *ds2 = ds1 + nq, *par = ds2 + nq;
    const int TP = D <= 16 ? 256 : 64;
    const size_t lds = sizeof(double) * ((size_t)E * TP + nq) + sizeof(unsigned short) * (size_t)((nq + 3) & ~3);
    HIPC(c, hipFuncSetAttribute((const void*)glk_sums_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    int64_t nb = (c->N + TP - 1) / TP;
    if (nb > kMaxPart) nb = kMaxPart;
    HIPC(c, hipMemsetAsync(dmu, 0, sizeof(double) * E, c->stream));
    glk_sums_kernel<<<(int)nb, kGlkSumsBlock, lds, c->stream>>>(c->r_new, c->x_new, c->N, D, dmu, TP, c->part);
    sum_final_kernel<<<final_grid(nq), kRedBlock, 0, c->stream>>>(c->part, (int)nb, nq, ds1);
    glk_mean_kernel<<<1, 64, 0, c->stream>>>(ds1, E, (double)c->N, dmu);
    glk_sums_kernel<<<(int)nb, kGlkSumsBlock, lds, c->stream>>>(c->r_new, c->x_new, c->N, D, dmu, TP, c->part);
    sum_final_kernel<<<final_grid(nq), kRedBlock, 0, c->stream>>>(c->part, (int)nb, nq, ds2);
    glk_algebra_kernel<<<1, 64, sizeof(double) * 6 * D * D, c->stream>>>(ds2, dmu, D, (double)c->N, par);
    HIPC(c, hipGetLastError());
    double tail[4];
    HIPC(c, hipMemcpyAsync(tail, par + 2 * D + 2 * D * D, sizeof(double) * 4, hipMemcpyDeviceToHost, c->stream));
    // the log-density is enqueued behind the algebra before its status is known: the values only COUNT if it is 0
    const size_t lds2 = sizeof(double) * ((size_t)2 * D + 2 * D * D + (size_t)D * 256);
    HIPC(c, hipFuncSetAttribute((const void*)glk_logpdf_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    glk_logpdf_kernel<<<grid_for(c->N, 256), 256, lds2, c->stream>>>(c->r_new, c->x_new, c->N, D, par, 0.0, c->Lg,
                                                                     par + 2 * D + 2 * D * D);
    HIPC(c, hipGetLastError());
    HIPC(c, stream_wait(c->stream));
    info[0] = tail[1]; info[1] = tail[0]; info[2] = tail[2]; info[3] = tail[3];
    c->lg_set = tail[1] == 0.0;
    return 0;
}

// The same over SHARDS (all-gather of 2D + (2D)^2-ish sums): three stages with the caller's all-gather of
// nq = E + E (E + 1) / 2 doubles (E = 2 D) between them -- smcn_gauss_lkernel_buffers names the two buffers --
//   stage 0: un-shifted sums of this shard -> local            [all-gather local -> gathered]
//   stage 1: rows added in rank order, mean over n_total, centred sums of this shard -> local      [all-gather]
//   stage 2: rows added, the D x D algebra (every rank computes the same bits), the conditional log-density of this
//            shard's particles; waits and returns info as smcn_gauss_lkernel_device does.
// world == 1 skips nothing but the all-gather: stage 1 and 2 then read the local row.
int smcn_gauss_lkernel_buffers(smcn_ctx* c, int world, void** local, void** gathered) {
    CHECK_CTX(c);
    if (world < 1 || world > 64 || !local || !gathered) FAIL(c, "smcn_gauss_lkernel_buffers: bad arguments");
    const int D = c->D, E = 2 * D, nq = E + E * (E + 1) / 2;
    if (D > kGlkMaxD) FAIL(c, "smcn_gauss_lkernel_buffers: D > 32 not supported (use the host algebra)");
    if (c->glk_world < world) {
        HIPC(c, stream_wait(c->stream));
        if (c->glk_xchg) (void)cached_free(c->glk_xchg);
        c->glk_xchg = nullptr;
        HIPC(c, dalloc(&c->glk_xchg, (int64_t)(world + 1) * nq));
        c->glk_world = world;
    }
    *local = c->glk_xchg;
    *gathered = c->glk_xchg + nq;
    return 0;
}

int smcn_gauss_lkernel_stage(smcn_ctx* c, int stage, int world, double n_total, double info[4]) {
    CHECK_CTX(c);
    Range roctx_range("smcn:gauss_lkernel");
    HIPC(c, hipSetDevice(c->device));
    const int D = c->D, E = 2 * D, nq = E + E * (E + 1) / 2;
    if (stage < 0 || stage > 2 || world < 1 || world > c->glk_world || !c->glk_xchg)
        FAIL(c, "smcn_gauss_lkernel_stage: call smcn_gauss_lkernel_buffers(world) first; stages 0, 1, 2");
    if (!(n_total >= 2.0)) FAIL(c, "smcn_gauss_lkernel_stage: needs at least two particles");
    const size_t npar = (size_t)2 * D + 2 * D * D + 4;
    if (!c->glk_buf) HIPC(c, cached_malloc((void**)&c->glk_buf, sizeof(double) * (E + 2 * (size_t)nq + npar)));
    double *dmu = c->glk_buf, *ds1 = dmu + E, *ds2 = ds1 + nq, *par = ds2 + nq;
    double *loc = c->glk_xchg, *gat = loc + nq;
    const double* rows = world > 1 ? gat : loc;
    const int TP = D <= 16 ? 256 : 64;
    const size_t lds = sizeof(double) * ((size_t)E * TP + nq) + sizeof(unsigned short) * (size_t)((nq + 3) & ~3);
    HIPC(c, hipFuncSetAttribute((const void*)glk_sums_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    int64_t nb = (c->N + TP - 1) / TP;
    if (nb > kMaxPart) nb = kMaxPart;
    if (stage == 0) {
        HIPC(c, hipMemsetAsync(dmu, 0, sizeof(double) * E, c->stream));
        glk_sums_kernel<<<(int)nb, kGlkSumsBlock, lds, c->stream>>>(c->r_new, c->x_new, c->N, D, dmu, TP, c->part);
        sum_final_kernel<<<final_grid(nq), kRedBlock, 0, c->stream>>>(c->part, (int)nb, nq, loc);
        HIPC(c, hipGetLastError());
        return 0;
    }
    if (stage == 1) {
        glk_combine_kernel<<<grid_for(nq, 256), 256, 0, c->stream>>>(rows, world, nq, ds1);
        glk_mean_kernel<<<1, 64, 0, c->stream>>>(ds1, E, n_total, dmu);
        glk_sums_kernel<<<(int)nb, kGlkSumsBlock, lds, c->stream>>>(c->r_new, c->x_new, c->N, D, dmu, TP, c->part);
        sum_final_kernel<<<final_grid(nq), kRedBlock, 0, c->stream>>>(c->part, (int)nb, nq, loc);
        HIPC(c, hipGetLastError());
        return 0;
    }
    if (!info) FAIL(c, "smcn_gauss_lkernel_stage: null");
    glk_combine_kernel<<<grid_for(nq, 256), 256, 0, c->stream>>>(rows, world, nq, ds2);
    glk_algebra_kernel<<<1, 64, sizeof(double) * 6 * D * D, c->stream>>>(ds2, dmu, D, n_total, par);
    HIPC(c, hipGetLastError());
    double tail[4];
    HIPC(c, hipMemcpyAsync(tail, par + 2 * D + 2 * D * D, sizeof(double) * 4, hipMemcpyDeviceToHost, c->stream));
    const size_t lds2 = sizeof(double) * ((size_t)2 * D + 2 * D * D + (size_t)D * 256);
    HIPC(c, hipFuncSetAttribute((const void*)glk_logpdf_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    glk_logpdf_kernel<<<grid_for(c->N, 256), 256, lds2, c->stream>>>(c->r_new, c->x_new, c->N, D, par, 0.0, c->Lg,
                                                                     par + 2 * D + 2 * D * D);
    HIPC(c, hipGetLastError());
    HIPC(c, stream_wait(c->stream));
    info[0] = tail[1]; info[1] = tail[0]; info[2] = tail[2]; info[3] = tail[3];
    c->lg_set = tail[1] == 0.0;
    return 0;
}

int smcn_accept_reject(smcn_ctx* c, double phi, const double* u, int64_t iteration) {
    CHECK_CTX(c);
    Range roctx_range("smcn:accept_reject");
    const int64_t N = c->N;
    double* du = nullptr;
    if (u) {
        HIPC(c, hipMemcpyAsync(c->work, u, sizeof(double) * N, hipMemcpyHostToDevice, c->stream));
        du = c->work;
    }
    { int rcm = momentum_dn(c); if (rcm) return rcm; }
    accept_reject_kernel<<<grid_for(N, 256), 256, 0, c->stream>>>(c->x, c->r, c->x_new, c->r_new, c->lpri0, c->llik0,
                                                                  c->lpri1, c->llik1, du, c->seed, (uint32_t)iteration,
                                                                  c->base, phi, N, c->D);
    HIPC(c, hipGetLastError());
    HIPC(c, stream_wait(c->stream));
    return 0;
}

int smcn_reweight_asymptotic(smcn_ctx* c, double phi_old, double phi_new) {
    CHECK_CTX(c);
    Range roctx_range("smcn:reweight");
    reweight_asymptotic_kernel<<<grid_for(c->N, 256), 256, 0, c->stream>>>(c->logw, c->lpri0, c->llik0, c->logw_new,
                                                                           c->N, phi_old, phi_new);
    HIPC(c, hipGetLastError());
    HIPC(c, stream_wait(c->stream));
    return 0;
}

int smcn_set_logw_density_ratio(smcn_ctx* c, double phi_num, double phi_den) {
    CHECK_CTX(c);
    int rc = eval_resident(c, c->x, 1.0, nullptr, c->lpri1, c->llik1);
    if (rc) return rc;
    density_ratio_kernel<<<grid_for(c->N, 256), 256, 0, c->stream>>>(c->lpri1, c->llik1, c->logw, c->N, phi_num,
                                                                     phi_den);
    HIPC(c, hipGetLastError());
    HIPC(c, stream_wait(c->stream));
    return 0;
}

int smcn_eval_proposed_parts(smcn_ctx* c, int which) {
    CHECK_CTX(c);
    HIPC(c, hipSetDevice(c->device));
    int rc = eval_resident(c, which == 0 ? c->x : c->x_new, 1.0, nullptr, c->lpri1, c->llik1);
    if (rc) return rc;
    HIPC(c, stream_wait(c->stream));
    return 0;
}

int smcn_commit(smcn_ctx* c, int64_t* n_moved) {
    CHECK_CTX(c);
    Range roctx_range("smcn:commit");
    if (n_moved) {
        const int g = red_grid(c->N);
        moved_partial_kernel<<<g, kRedBlock, 0, c->stream>>>(c->x, c->x_new, c->N, c->D, c->part);
        sum_final_kernel<<<1, kRedBlock, 0, c->stream>>>(c->part, g, 1, c->scal + 9);
        HIPC(c, hipGetLastError());
        double v = 0.0;
        HIPC(c, hipMemcpyAsync(&v, c->scal + 9, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPC(c, stream_wait(c->stream));
        *n_moved = (int64_t)v;
    }
    std::swap(c->x, c->x_new);        // samples.py:221
    std::swap(c->logw, c->logw_new);  // samples.py:222
    return 0;
}

// ---- device-resident loop ---------------------------------------------------------------------------
int smcn_fast_begin(smcn_ctx* c, int64_t K, int save_history, int world) {
    CHECK_CTX(c);
    if (K < 0 || world < 1 || world > 64) FAIL(c, "smcn_fast_begin: bad arguments");
    SETUP_TRACE_DECL;
    HIPC(c, stream_wait(c->stream));
    SETUP_TRACE("fast_begin: wait");
    const int HS = hist_stride(c->Dc), NQ = 4 + 2 * c->Dc;
    for (double** p : {&c->hist, &c->ss, &c->lp, &c->gath, &c->hist_x, &c->hist_logw, &c->u_res}) {
        if (*p) (void)cached_free(*p);
        *p = nullptr;
    }
    HIPC(c, dalloc(&c->hist, (K + 1) * HS));
    HIPC(c, dalloc(&c->ss, SS_SHIFT + c->Dc + 8));
    HIPC(c, dalloc(&c->lp, NQ));
    HIPC(c, dalloc(&c->gath, (int64_t)world * NQ));
    HIPC(c, dalloc(&c->u_res, c->N));
    HIPC(c, hipMemsetAsync(c->hist, 0, sizeof(double) * (K + 1) * HS, c->stream));   // (the context's stream: see smcn_ctx_create)
    HIPC(c, hipMemsetAsync(c->ss, 0, sizeof(double) * (SS_SHIFT + c->Dc + 8), c->stream));
    SETUP_TRACE("fast_begin: small buffers");
    c->fast_K = K;
    c->fast_hist = save_history != 0;
    if (c->hist_h) { (void)hipHostFree(c->hist_h); c->hist_h = nullptr; }
    c->hist_h_valid = false;
    if (c->fast_hist) {
        HIPC(c, dalloc(&c->hist_x, (K + 1) * c->N * c->D));
        HIPC(c, dalloc(&c->hist_logw, (K + 1) * c->N));
        HIPC(c, hipMemcpyAsync(c->hist_x, c->x, sizeof(double) * c->N * c->D, hipMemcpyDeviceToDevice, c->stream));
        HIPC(c, hipMemcpyAsync(c->hist_logw, c->logw, sizeof(double) * c->N, hipMemcpyDeviceToDevice, c->stream));
        SETUP_TRACE("fast_begin: history buffers");
        if (c->N * c->D * kDlChunk <= ((int64_t)1 << 28)) {   // (wide particles: the staging is made when first asked for)
            int rc = dl_prepare(c);
            if (rc) return rc;
            SETUP_TRACE("fast_begin: download stream");
        }
    }
    return 0;
}

int smcn_fast_buffers(smcn_ctx* c, void** local_partials, void** gathered, int* nq) {
    CHECK_CTX(c);
    if (c->fast_K < 0) FAIL(c, "smcn_fast_buffers: call smcn_fast_begin first");
    if (local_partials) *local_partials = c->lp;
    if (gathered) *gathered = c->gath;
    if (nq) *nq = 4 + 2 * c->Dc;
    return 0;
}

int smcn_set_resample_uniforms(smcn_ctx* c, const double* u) {
    CHECK_CTX(c);
    if (c->fast_K < 0 || !u) FAIL(c, "smcn_set_resample_uniforms: needs smcn_fast_begin and a buffer");
    HIPC(c, hipMemcpyAsync(c->u_res, u, sizeof(double) * c->N, hipMemcpyHostToDevice, c->stream));
    c->u_set = true;
    return 0;
}

/* host-side exchange for communicators without a device path */
int smcn_partials_get(smcn_ctx* c, double* out) {
    CHECK_CTX(c);
    if (c->fast_K < 0 || !out) FAIL(c, "smcn_partials_get: no smcn_fast_begin");
    HIPC(c, hipMemcpyAsync(out, c->lp, sizeof(double) * (4 + 2 * c->Dc), hipMemcpyDeviceToHost, c->stream));
    HIPC(c, stream_wait(c->stream));
    return 0;
}
int smcn_partials_set_gathered(smcn_ctx* c, const double* in, int world) {
    CHECK_CTX(c);
    if (c->fast_K < 0 || !in || world < 1 || world > 64) FAIL(c, "smcn_partials_set_gathered: bad arguments");
    HIPC(c, hipMemcpyAsync(c->gath, in, sizeof(double) * world * (4 + 2 * c->Dc), hipMemcpyHostToDevice, c->stream));
    HIPC(c, stream_wait(c->stream));
    return 0;
}

}  // extern "C"
// shard partials [max, cnt, s1, s2, sum e c(x), sum e (c(x)-shift)^2] of one generation -> out (device)
static int enqueue_partials(smcn_ctx* c, const double* logw, const double* x, double* out,
                            const double* shift = nullptr, int ngen = 1) {
    // ngen consecutive generations ([ngen][N] weights, [ngen][D][N] particles) in two launches
    const int64_t N = c->N;
    int g = red_grid(N);
    // many generations at once: fewer, fatter blocks (the kernel is a chain of block reductions; at one element per
    // thread it was latency-bound: 46 us for 20 generations of 65 536 particles)
    if (ngen >= 4 && g > 64) g = 64;
    if (c->D >= 64 && g > 128) g = 128;   // wide particles: 2 block reductions per coordinate -- four particles per thread
                                          // between them (measured at D = 256, N = 131 072: 155 us against 197 at 512 blocks)
    const int NQ = 4 + 2 * c->Dc;
    while ((int64_t)g * NQ * ngen > (int64_t)kMaxPart * (4 * c->D * c->D + 2 * c->D + 8) && g > 1) g /= 2;
    const int nz = c->D >= 64 ? 8 : 1;
    gen_partials_kernel<<<dim3(g, ngen, nz), kRedBlock, 0, c->stream>>>(logw, x, N, c->D, c->model,
                                                                        shift ? shift : c->ss + SS_SHIFT, c->part, N,
                                                                        N * c->D, (c->D > 8 && ngen == 1 && nz == 1) ? c->work : nullptr);
    const int qb = c->Dc <= 16 ? 1 : (2 * c->Dc + 7) / 8 > 64 ? 64 : (2 * c->Dc + 7) / 8;
    gen_reduce_blocks_kernel<<<dim3(ngen, qb), kRedBlock, 0, c->stream>>>(c->part, g, c->Dc, out);
    HIPC(c, hipGetLastError());
    return 0;
}
extern "C" {
int smcn_step_begin(smcn_ctx* c, int64_t k) {
    CHECK_CTX(c);
    c->hist_h_valid = false;
    Range roctx_range("smcn:normalise");
    if (c->fast_K < 0 || k < 0 || k > c->fast_K) FAIL(c, "smcn_step_begin: bad iteration / no smcn_fast_begin");
    int rc = enqueue_partials(c, c->logw, c->x, c->lp);
    if (rc) return rc;
    // single shard: the "gathered" block is this shard's partials (several shards: the caller's all-gather fills it)
    HIPC(c, hipMemcpyAsync(c->gath, c->lp, sizeof(double) * (4 + 2 * c->Dc), hipMemcpyDeviceToDevice, c->stream));
    return 0;
}

int smcn_step_finish(smcn_ctx* c, int64_t k, int world, int rank, double n_total, double step_size, double phi,
                     int max_depth, double delta_max, int lkernel, int last, const double* tape,
                     const int64_t* tape_off) {
    CHECK_CTX(c);
    Range roctx_range("smcn:step");
    if (c->fast_K < 0 || k < 0 || k > c->fast_K) FAIL(c, "smcn_step_finish: bad iteration / no smcn_fast_begin");
    if (world < 1 || rank < 0 || rank >= world) FAIL(c, "smcn_step_finish: bad world/rank");
    if (lkernel != SMCN_LKERNEL_FORWARD) FAIL(c, "smcn_step_finish: only the forward L-kernel runs device-resident");
    const int64_t N = c->N;
    const int HS = hist_stride(c->Dc);
    double* hk = c->hist + k * HS;
    combine_ranks_kernel<<<1, 64, 0, c->stream>>>(c->gath, world, rank, c->Dc, n_total,
                                                  log((double)N), c->ss + SS_SHIFT, phi, hk, c->ss);
    wn_dev_kernel<<<grid_for(N, 256), 256, 0, c->stream>>>(c->logw, c->wn, N, c->ss);
    HIPC(c, hipGetLastError());
    if (last) return 0;
    // samples.py:116-146, decided on the device
    enqueue_resample_if(c, c->u_set ? c->u_res : nullptr, (uint32_t)k);
    c->u_set = false;
    HIPC(c, hipGetLastError());
    bool reweighted = false;
    int rc = propose_async(c, step_size, phi, max_depth, delta_max, k, tape, tape_off, true, &reweighted);
    if (rc) return rc;
    if (!reweighted) enqueue_reweight_forward(c);
    const int g = red_grid(N);
    isum_partial_kernel<<<g, kRedBlock, 0, c->stream>>>(c->nleap, N, c->part);
    sum_to_kernel<<<1, kRedBlock, 0, c->stream>>>(c->part, g, hk + H_LEAPS);
    enqueue_moved_count(c, g, c->part + g);
    sum_to_kernel<<<1, kRedBlock, 0, c->stream>>>(c->part + g, g, hk + H_MOVED);
    c->kin_valid = false;
    HIPC(c, hipGetLastError());
    std::swap(c->x, c->x_new);        // samples.py:221
    std::swap(c->logw, c->logw_new);  // samples.py:222
    if (c->fast_hist) {               // smc_sampler.py:139-140
        HIPC(c, hipMemcpyAsync(c->hist_x + (k + 1) * N * c->D, c->x, sizeof(double) * N * c->D,
                               hipMemcpyDeviceToDevice, c->stream));
        HIPC(c, hipMemcpyAsync(c->hist_logw + (k + 1) * N, c->logw, sizeof(double) * N, hipMemcpyDeviceToDevice,
                               c->stream));
    }
    return 0;
}

// ---- fused transitions: B SMC iterations per NUTS launch, speculating "no resampling" ----------
// Between two resampling events a particle's next transition depends only on its own sample
// (Philox draws are keyed by iteration and particle), so B iterations of one particle can run
// back to back inside one launch; the per-generation weights, ESS and estimates follow from the
// recorded transitions.  If some generation's ESS turns out below N/2 (samples.py:120) the
// generations after it are discarded and the loop resumes there -- results are those of the
// one-iteration-per-launch schedule, bit for bit.  This amortises the tail of a launch (the
// longest tree) over B iterations.
int smcn_fuse_begin(smcn_ctx* c, int Bmax, int world) {
    CHECK_CTX(c);
    if (c->fast_K < 0) FAIL(c, "smcn_fuse_begin: call smcn_fast_begin first");
    if (Bmax < 1 || Bmax > 64 || world < 1 || world > 64) FAIL(c, "smcn_fuse_begin: bad arguments");
    HIPC(c, stream_wait(c->stream));
    const int NQ = 4 + 2 * c->Dc;
    for (double** p : {&c->lpB, &c->gathB, &c->gen_x, &c->gen_logw, &c->cnt, &c->shiftB}) {
        if (*p) (void)cached_free(*p);
        *p = nullptr;
    }
    HIPC(c, dalloc(&c->lpB, (int64_t)Bmax * NQ));
    HIPC(c, dalloc(&c->gathB, (int64_t)Bmax * world * NQ));
    HIPC(c, dalloc(&c->cnt, 2 * Bmax));
    HIPC(c, dalloc(&c->shiftB, c->Dc));
    if (c->ss_scratch) (void)cached_free(c->ss_scratch);
    c->ss_scratch = nullptr;
    HIPC(c, dalloc(&c->ss_scratch, (int64_t)Bmax * (SS_SHIFT + c->Dc + 8)));
    if (c->rows_h) (void)hipHostFree(c->rows_h);
    c->rows_h = nullptr;
    HIPC(c, hipHostMalloc((void**)&c->rows_h, sizeof(double) * (size_t)Bmax * hist_stride(c->Dc)));
    if (!c->ev_rows) HIPC(c, hipEventCreateWithFlags(&c->ev_rows, hipEventDisableTiming));
    if (!c->fast_hist) {
        HIPC(c, dalloc(&c->gen_x, (int64_t)Bmax * c->N * c->D));
        HIPC(c, dalloc(&c->gen_logw, (int64_t)Bmax * c->N));
    }
    c->fuse_max = Bmax;
    // lane kernel: the per-transition record buffers and the overflow tree-stack area for blocks of up to Bmax
    // transitions -- here, once, rather than at the first launch that needs them (a cold sample() spent its first
    // milliseconds in hipMalloc)
    if (c->lane_kernel && Bmax > c->rec_cap) {
        constexpr int D = ArmaLaneModel::D;
        if (c->in_rec) (void)cached_free(c->in_rec);
        if (c->out_rec) (void)cached_free(c->out_rec);
        c->in_rec = c->out_rec = nullptr;
        HIPC(c, dalloc(&c->in_rec, c->N * Bmax * n2_in_doubles(D)));
        HIPC(c, dalloc(&c->out_rec, c->N * Bmax * n2_out_doubles(D)));
        c->rec_cap = Bmax;
        const int64_t blocks = (c->N + kN3Block - 1) / kN3Block;
        const int64_t need = blocks * kN3Block * 2 * (int64_t)n3_ovf_pairs(D, 3, 3);
        if (need > c->n2_ovf_len) {
            if (c->n2_ovf) (void)cached_free(c->n2_ovf);
            c->n2_ovf = nullptr;
            HIPC(c, dalloc(&c->n2_ovf, need));
            c->n2_ovf_len = need;
        }
    }
    return 0;
}

int smcn_fuse_buffers(smcn_ctx* c, void** local_partials, void** gathered, int* nq) {
    CHECK_CTX(c);
    if (c->fuse_max < 1) FAIL(c, "smcn_fuse_buffers: call smcn_fuse_begin first");
    if (local_partials) *local_partials = c->lpB;
    if (gathered) *gathered = c->gathB;
    if (nq) *nq = 4 + 2 * c->Dc;
    return 0;
}

}  // extern "C"
__global__ void store_counts_kernel(const double* cnt, int B, double* hist, int HS) {
    const int b = threadIdx.x;
    if (b < B) { hist[b * HS + H_LEAPS] = cnt[2 * b]; hist[b * HS + H_MOVED] = cnt[2 * b + 1]; }
}
static double* gen_x_ptr(smcn_ctx* c, int64_t k0) { return c->fast_hist ? c->hist_x + (k0 + 1) * c->N * c->D : c->gen_x; }
static double* gen_logw_ptr(smcn_ctx* c, int64_t k0) { return c->fast_hist ? c->hist_logw + (k0 + 1) * c->N : c->gen_logw; }
extern "C" {

// Generation k0's scalars from the gathered partials (after smcn_step_begin + exchange); returns
// whether the population has to resample (samples.py:120).  Needed by the caller only with several
// shards, where resampling is a GLOBAL operation; one shard decides on the device (smcn_fuse_run).
int smcn_fuse_decide(smcn_ctx* c, int64_t k0, int world, int rank, double n_total, double phi, int* resample) {
    CHECK_CTX(c);
    c->hist_h_valid = false;
    Range roctx_range("smcn:normalise+ess");
    if (c->fast_K < 0 || k0 < 0 || k0 > c->fast_K || !resample) FAIL(c, "smcn_fuse_decide: bad arguments");
    const int64_t N = c->N;
    const int HS = hist_stride(c->Dc);
    combine_ranks_kernel<<<1, 64, 0, c->stream>>>(c->gath, world, rank, c->Dc, n_total, log((double)N),
                                                  c->ss + SS_SHIFT, phi, c->hist + k0 * HS, c->ss);
    wn_dev_kernel<<<grid_for(N, 256), 256, 0, c->stream>>>(c->logw, c->wn, N, c->ss);
    HIPC(c, hipGetLastError());
    double flag = 0.0;
    HIPC(c, hipMemcpyAsync(&flag, c->ss + SS_FLAG, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPC(c, stream_wait(c->stream));
    *resample = flag != 0.0;
    return 0;
}

// after smcn_step_begin(k0) + exchange: generation k0's scalars, conditional resampling, then B
// transitions per particle in one launch and the shard partials of generations k0+1 .. k0+B-1.
// decided != 0: smcn_fuse_decide (and a global resampling, if needed) already ran for k0;
// decided == 0: the decision and a shard-LOCAL resampling happen on the device, no host wait.
int smcn_fuse_run(smcn_ctx* c, int64_t k0, int B, int world, int rank, double n_total, double step_size, double phi,
                  int max_depth, double delta_max, int decided) {
    CHECK_CTX(c);
    c->hist_h_valid = false;
    Range roctx_range("smcn:nuts");
    if (c->fuse_max < 1 || B < 1 || B > c->fuse_max || k0 < 0 || k0 + B > c->fast_K)
        FAIL(c, "smcn_fuse_run: bad iteration range / no smcn_fuse_begin");
    const int64_t N = c->N;
    const int HS = hist_stride(c->Dc), NQ = 4 + 2 * c->Dc;
    double* hk = c->hist + k0 * HS;
    if (!decided) {
        combine_ranks_kernel<<<1, 64, 0, c->stream>>>(c->gath, world, rank, c->Dc, n_total, log((double)N),
                                                      c->ss + SS_SHIFT, phi, hk, c->ss);
        wn_dev_kernel<<<grid_for(N, 256), 256, 0, c->stream>>>(c->logw, c->wn, N, c->ss);
        enqueue_resample_if(c, nullptr, (uint32_t)k0);
        HIPC(c, hipGetLastError());
    }
    HIPC(c, hipMemsetAsync(c->cnt, 0, sizeof(double) * 2 * B, c->stream));
    bool reweighted = false;
    int rc = propose_async(c, step_size, phi, max_depth, delta_max, k0, nullptr, nullptr, true, &reweighted, B,
                           gen_x_ptr(c, k0), gen_logw_ptr(c, k0), c->cnt);
    if (rc) return rc;
    if (reweighted) {
        store_counts_kernel<<<1, 64, 0, c->stream>>>(c->cnt, B, hk, HS);
    } else {
        // models without the fused-transition kernel: one transition, generation k0+1 written here
        if (B != 1) FAIL(c, "smcn_fuse_run: this model runs one iteration per launch");
        enqueue_reweight_forward(c);
        const int g = red_grid(N);
        isum_partial_kernel<<<g, kRedBlock, 0, c->stream>>>(c->nleap, N, c->part);
        sum_to_kernel<<<1, kRedBlock, 0, c->stream>>>(c->part, g, hk + H_LEAPS);
        enqueue_moved_count(c, g, c->part + g);
        sum_to_kernel<<<1, kRedBlock, 0, c->stream>>>(c->part + g, g, hk + H_MOVED);
        c->kin_valid = false;
        // generation k0+1 = (x_new, logw_new): committed by a pointer swap in smcn_fuse_finish; only a kept
        // history needs a copy (at D = 256 the two 268 MB copies per iteration were 18 % of the step)
        if (c->fast_hist) {
            HIPC(c, hipMemcpyAsync(gen_x_ptr(c, k0), c->x_new, sizeof(double) * N * c->D, hipMemcpyDeviceToDevice,
                                   c->stream));
            HIPC(c, hipMemcpyAsync(gen_logw_ptr(c, k0), c->logw_new, sizeof(double) * N, hipMemcpyDeviceToDevice,
                                   c->stream));
        }
    }
    c->plain_block = !reweighted;
    // one variance shift (the mean of generation k0) for every generation of the block
    HIPC(c, hipMemcpyAsync(c->shiftB, c->ss + SS_SHIFT, sizeof(double) * c->Dc, hipMemcpyDeviceToDevice, c->stream));
    if (B > 1) {   // generations k0+1 .. k0+B-1 in one batch (the last one opens the next call)
        rc = enqueue_partials(c, gen_logw_ptr(c, k0), gen_x_ptr(c, k0), c->lpB, c->shiftB, B - 1);
        if (rc) return rc;
    }
    if (B > 1 && world == 1)
        HIPC(c, hipMemcpyAsync(c->gathB, c->lpB, sizeof(double) * (B - 1) * NQ, hipMemcpyDeviceToDevice, c->stream));
    HIPC(c, hipGetLastError());
    return 0;
}

#ifdef SMCN_LEGACY_ABI
int smcn_fuse_partials_get(smcn_ctx* c, int B, double* out) {
    CHECK_CTX(c);
    if (c->fuse_max < 1 || B < 2 || B > c->fuse_max || !out) FAIL(c, "smcn_fuse_partials_get: bad arguments");
    HIPC(c, hipMemcpyAsync(out, c->lpB, sizeof(double) * (B - 1) * (4 + 2 * c->Dc), hipMemcpyDeviceToHost, c->stream));
    HIPC(c, stream_wait(c->stream));
    return 0;
}
int smcn_fuse_partials_set(smcn_ctx* c, int B, int world, const double* in) {
    CHECK_CTX(c);
    if (c->fuse_max < 1 || B < 2 || B > c->fuse_max || !in) FAIL(c, "smcn_fuse_partials_set: bad arguments");
    HIPC(c, hipMemcpyAsync(c->gathB, in, sizeof(double) * world * (B - 1) * (4 + 2 * c->Dc), hipMemcpyHostToDevice,
                           c->stream));
    HIPC(c, stream_wait(c->stream));
    return 0;
}
#endif

// combine generations k0+1 .. k0+B-1 (gathered partials, rank-major [world][B-1][nq]), wait, and commit up to
// the first generation that has to resample: *n_ok in 1..B transitions were valid.
int smcn_fuse_finish(smcn_ctx* c, int64_t k0, int B, int world, int rank, double n_total, double phi, int* n_ok) {
    CHECK_CTX(c);
    if (c->fuse_max < 1 || B < 1 || B > c->fuse_max || !n_ok) FAIL(c, "smcn_fuse_finish: bad arguments");
    const int64_t N = c->N;
    const int HS = hist_stride(c->Dc), NQ = 4 + 2 * c->Dc;
    for (int g = 1; g < B; ++g)   // gathered block is rank-major: [world][B-1][nq]
        combine_ranks_kernel<<<1, 64, 0, c->stream>>>(c->gathB + (int64_t)(g - 1) * NQ, world, rank, c->Dc, n_total,
                                                      log((double)N), c->shiftB, phi, c->hist + (k0 + g) * HS,
                                                      c->ss, (B - 1) * NQ);
    HIPC(c, hipGetLastError());
    std::vector<double> rows((size_t)B * HS);
    HIPC(c, hipMemcpyAsync(rows.data(), c->hist + k0 * HS, sizeof(double) * B * HS, hipMemcpyDeviceToHost, c->stream));
    HIPC(c, stream_wait(c->stream));
    int ok = B;
    for (int g = 1; g < B; ++g)
        if (rows[(size_t)g * HS + H_RESAMPLED] != 0.0) { ok = g; break; }
    // the committed state is generation k0 + ok
    if (c->plain_block && B == 1) {
        std::swap(c->x, c->x_new);        // samples.py:221
        std::swap(c->logw, c->logw_new);  // samples.py:222
    } else {
        HIPC(c, hipMemcpyAsync(c->x, gen_x_ptr(c, k0) + (int64_t)(ok - 1) * N * c->D, sizeof(double) * N * c->D,
                               hipMemcpyDeviceToDevice, c->stream));
        HIPC(c, hipMemcpyAsync(c->logw, gen_logw_ptr(c, k0) + (int64_t)(ok - 1) * N, sizeof(double) * N,
                               hipMemcpyDeviceToDevice, c->stream));
    }
    *n_ok = ok;
    return 0;
}

// ---- pipelined fused blocks ---------------------------------------------------------------
// The same arithmetic as smcn_fuse_run / smcn_fuse_finish, cut so that the host never idles the
// device: the statistics of ALL B generations of a block are produced at its end (one combine
// launch, rows read back through pinned memory behind an event), and the caller may enqueue the
// next block's prep + NUTS launch (speculating "no resampling") before it waits for that event.
// Sequence per block: [smcn_block_resample_local] smcn_block_launch -> smcn_block_post ->
// exchange of smcn_fuse_buffers ([B][nq]) -> smcn_block_stats -> (smcn_block_commit +
// smcn_block_launch of the next block) -> smcn_block_wait.
int smcn_block_resample_local(smcn_ctx* c, int64_t k0) {   // after smcn_fuse_decide said "resample"
    CHECK_CTX(c);
    if (c->fast_K < 0 || k0 < 0 || k0 > c->fast_K) FAIL(c, "smcn_block_resample_local: bad iteration");
    enqueue_resample_if(c, nullptr, (uint32_t)k0);
    HIPC(c, hipGetLastError());
    return 0;
}
int smcn_block_launch(smcn_ctx* c, int64_t k0, int B, double step_size, double phi, int max_depth, double delta_max) {
    CHECK_CTX(c);
    Range roctx_range("smcn:nuts");
    if (c->fuse_max < 1 || B < 1 || B > c->fuse_max || k0 < 0 || k0 + B > c->fast_K)
        FAIL(c, "smcn_block_launch: bad iteration range / no smcn_fuse_begin");
    // the leapfrog / moved counters of the block are cleared HERE, in front of the long NUTS launch, not between it and
    // the post kernel (the previous block's counts were copied into the history by its own smcn_block_post)
    HIPC(c, hipMemsetAsync(c->cnt, 0, sizeof(double) * 2 * B, c->stream));
    bool reweighted = false;
    return propose_async(c, step_size, phi, max_depth, delta_max, k0, nullptr, nullptr, true, &reweighted, B,
                         gen_x_ptr(c, k0), gen_logw_ptr(c, k0), c->cnt, 1);
}
int smcn_block_post(smcn_ctx* c, int64_t k0, int B, int world) {
    CHECK_CTX(c);
    c->hist_h_valid = false;
    Range roctx_range("smcn:reweight+estimate");
    if (c->fuse_max < 1 || B < 1 || B > c->fuse_max || k0 < 0 || k0 + B > c->fast_K)
        FAIL(c, "smcn_block_post: bad iteration range");
    const int HS = hist_stride(c->Dc), NQ = 4 + 2 * c->Dc;
    bool reweighted = false;
    int rc = propose_async(c, 0.0, 1.0, 0, 0.0, k0, nullptr, nullptr, true, &reweighted, B, gen_x_ptr(c, k0),
                           gen_logw_ptr(c, k0), c->cnt, 2);
    if (rc) return rc;
    store_counts_kernel<<<1, 64, 0, c->stream>>>(c->cnt, B, c->hist + k0 * HS, HS);
    // all B generations; the variance shift of the block is the mean of generation k0 (its history row)
    rc = enqueue_partials(c, gen_logw_ptr(c, k0), gen_x_ptr(c, k0), c->lpB, c->hist + k0 * HS + H_MEAN, B);
    if (rc) return rc;
    if (world == 1)
        HIPC(c, hipMemcpyAsync(c->gathB, c->lpB, sizeof(double) * B * NQ, hipMemcpyDeviceToDevice, c->stream));
    return 0;
}
int smcn_block_partials_get(smcn_ctx* c, int B, double* out) {
    CHECK_CTX(c);
    if (c->fuse_max < 1 || B < 1 || B > c->fuse_max || !out) FAIL(c, "smcn_block_partials_get: bad arguments");
    HIPC(c, hipMemcpyAsync(out, c->lpB, sizeof(double) * B * (4 + 2 * c->Dc), hipMemcpyDeviceToHost, c->stream));
    HIPC(c, stream_wait(c->stream));
    return 0;
}
int smcn_block_partials_set(smcn_ctx* c, int B, int world, const double* in) {
    CHECK_CTX(c);
    if (c->fuse_max < 1 || B < 1 || B > c->fuse_max || !in) FAIL(c, "smcn_block_partials_set: bad arguments");
    HIPC(c, hipMemcpyAsync(c->gathB, in, sizeof(double) * world * B * (4 + 2 * c->Dc), hipMemcpyHostToDevice,
                           c->stream));
    HIPC(c, stream_wait(c->stream));
    return 0;
}
int smcn_block_stats(smcn_ctx* c, int64_t k0, int B, int world, int rank, double n_total, double phi, int final_block) {
    CHECK_CTX(c);
    Range roctx_range("smcn:normalise+ess");
    if (c->fuse_max < 1 || B < 1 || B > c->fuse_max || k0 < 0 || k0 + B > c->fast_K)
        FAIL(c, "smcn_block_stats: bad iteration range");
    if (world < 1 || rank < 0 || rank >= world) FAIL(c, "smcn_block_stats: bad world/rank");
    const int HS = hist_stride(c->Dc);
    combine_ranks_gens_kernel<<<B, 64, 0, c->stream>>>(c->gathB, world, rank, c->Dc, n_total, log((double)c->N),
                                                       c->hist + k0 * HS + H_MEAN, phi, c->hist + (k0 + 1) * HS, HS,
                                                       c->ss, c->ss_scratch, SS_SHIFT + c->Dc + 8);
    HIPC(c, hipGetLastError());
    HIPC(c, hipMemcpyAsync(c->rows_h, c->hist + (k0 + 1) * HS, sizeof(double) * B * HS, hipMemcpyDeviceToHost,
                           c->stream));
    c->hist_h_valid = false;
    if (final_block) {   // the run ends with this block: the whole scalar history comes along behind the same event,
                         // so that smcn_fast_read needs no further round trip if the block turns out valid
        if (!c->hist_h) HIPC(c, hipHostMalloc((void**)&c->hist_h, sizeof(double) * (size_t)(c->fast_K + 1) * HS));
        HIPC(c, hipMemcpyAsync(c->hist_h, c->hist, sizeof(double) * (c->fast_K + 1) * HS, hipMemcpyDeviceToHost, c->stream));
        c->hist_h_valid = true;
    }
    HIPC(c, hipEventRecord(c->ev_rows, c->stream));
    return 0;
}
// *n_ok in 1..B: transitions valid before a generation that has to resample; *resample_next: whether
// generation k0 + *n_ok has to (always 1 if *n_ok < B).
int smcn_block_wait(smcn_ctx* c, int B, int* n_ok, int* resample_next) {
    CHECK_CTX(c);
    if (c->fuse_max < 1 || B < 1 || B > c->fuse_max || !n_ok || !resample_next)
        FAIL(c, "smcn_block_wait: bad arguments");
    HIPC(c, hipEventSynchronize(c->ev_rows));
    const int HS = hist_stride(c->Dc);
    int ok = B;
    for (int g = 1; g < B; ++g)
        if (c->rows_h[(size_t)(g - 1) * HS + H_RESAMPLED] != 0.0) { ok = g; break; }
    *n_ok = ok;
    *resample_next = c->rows_h[(size_t)(ok - 1) * HS + H_RESAMPLED] != 0.0;
    return 0;
}
// ESS of the B generations of the block just waited for (host copy; drives the block-size policy)
int smcn_block_ess(smcn_ctx* c, int B, double* out) {
    CHECK_CTX(c);
    if (c->fuse_max < 1 || B < 1 || B > c->fuse_max || !out) FAIL(c, "smcn_block_ess: bad arguments");
    const int HS = hist_stride(c->Dc);
    for (int g = 0; g < B; ++g) out[g] = c->rows_h[(size_t)g * HS + H_ESS];
    return 0;
}
int smcn_block_commit(smcn_ctx* c, int64_t k0, int ok) {   // the committed state becomes generation k0 + ok
    CHECK_CTX(c);
    if (c->fuse_max < 1 || ok < 1 || ok > c->fuse_max || k0 < 0 || k0 + ok > c->fast_K)
        FAIL(c, "smcn_block_commit: bad arguments");
    const int64_t N = c->N;
    HIPC(c, hipMemcpyAsync(c->x, gen_x_ptr(c, k0) + (int64_t)(ok - 1) * N * c->D, sizeof(double) * N * c->D,
                           hipMemcpyDeviceToDevice, c->stream));
    HIPC(c, hipMemcpyAsync(c->logw, gen_logw_ptr(c, k0) + (int64_t)(ok - 1) * N, sizeof(double) * N,
                           hipMemcpyDeviceToDevice, c->stream));
    return 0;
}

static int dl_prepare(smcn_ctx* c) {  // the download stream and its staging buffer (smcn_fast_begin with a history: up front)
    const int64_t ND = c->N * c->D;
    if (!c->dl_stream) HIPC(c, pool_take(c->device, &c->dl_stream));
    if (c->dl_stage_len < kDlChunk * ND) {
        if (c->dl_stage) (void)cached_free(c->dl_stage);
        c->dl_stage = nullptr;
        HIPC(c, dalloc(&c->dl_stage, kDlChunk * ND));
        c->dl_stage_len = kDlChunk * ND;
    }
    return 0;
}

// x_saved / logw_saved (smc_sampler.py:139-140) of generations k_from .. k_to -- which the caller knows to be final: every
// block up to k_to has been waited for -- into the caller's [K+1][N][D] / [K+1][N] arrays, on a stream of its own, so that
// the copies run beside the NUTS launch that is already enqueued instead of behind the whole loop.  Returns when the
// rows have landed (pageable destination memory: the copies are synchronous to the calling thread anyway).
int smcn_history_download(smcn_ctx* c, int64_t k_from, int64_t k_to, double* x_saved, double* logw_saved) {
    CHECK_CTX(c);
    Range roctx_range("smcn:history");
    if (c->fast_K < 0 || !c->fast_hist) FAIL(c, "smcn_history_download: no device history (smcn_fast_begin with save_history)");
    if (k_from < 0 || k_to > c->fast_K || !x_saved || !logw_saved) FAIL(c, "smcn_history_download: bad arguments");
    if (k_to < k_from) return 0;
    const int64_t N = c->N, ND = N * c->D;
    constexpr int64_t kChunk = kDlChunk;
    int rc0 = dl_prepare(c);
    if (rc0) return rc0;
    HIPC(c, hipMemcpyAsync(logw_saved + k_from * N, c->hist_logw + k_from * N, sizeof(double) * (k_to - k_from + 1) * N,
                           hipMemcpyDeviceToHost, c->dl_stream));
    for (int64_t k = k_from; k <= k_to; k += kChunk) {
        const int64_t n = (k_to - k + 1 < kChunk) ? (k_to - k + 1) : kChunk;
        for (int64_t g = 0; g < n; ++g)   // [D][N] -> [N][D] per generation
            transpose_kernel<<<grid_for(ND, 256), 256, 0, c->dl_stream>>>(c->hist_x + (k + g) * ND, c->dl_stage + g * ND, c->D, N);
        HIPC(c, hipGetLastError());
        HIPC(c, hipMemcpyAsync(x_saved + k * ND, c->dl_stage, sizeof(double) * n * ND, hipMemcpyDeviceToHost, c->dl_stream));
    }
    HIPC(c, hipStreamSynchronize(c->dl_stream));
    return 0;
}

int smcn_fast_read(smcn_ctx* c, double* hist, double* x_saved, double* logw_saved) {
    return smcn_fast_read_from(c, hist, x_saved, logw_saved, 0);
}

// ... generations k_from .. K only (the earlier ones are already in the caller's arrays: smcn_history_download)
int smcn_fast_read_from(smcn_ctx* c, double* hist, double* x_saved, double* logw_saved, int64_t k_from) {
    CHECK_CTX(c);
    Range roctx_range("smcn:history");
    if (c->fast_K < 0) FAIL(c, "smcn_fast_read: no smcn_fast_begin");
    if (k_from < 0 || k_from > c->fast_K + 1) FAIL(c, "smcn_fast_read_from: bad first generation");
    const int64_t K1 = c->fast_K + 1, N = c->N;
    const int HS = hist_stride(c->Dc);
    const bool cached = hist && c->hist_h_valid && !x_saved && !logw_saved;
    if (cached) {        // downloaded behind the last block's statistics (smcn_block_stats, final_block) and waited for
        HIPC(c, hipEventSynchronize(c->ev_rows));
        memcpy(hist, c->hist_h, sizeof(double) * K1 * HS);
        return 0;
    }
    if (hist) HIPC(c, hipMemcpyAsync(hist, c->hist, sizeof(double) * K1 * HS, hipMemcpyDeviceToHost, c->stream));
    if ((x_saved || logw_saved) && !c->fast_hist) FAIL(c, "smcn_fast_read: history was not enabled");
    if (logw_saved && k_from < K1)
        HIPC(c, hipMemcpyAsync(logw_saved + k_from * N, c->hist_logw + k_from * N, sizeof(double) * (K1 - k_from) * N,
                               hipMemcpyDeviceToHost, c->stream));
    if (x_saved) {
        int rc = ensure_stage(c, N * c->D);
        if (rc) return rc;
        for (int64_t k = k_from; k < K1; ++k) {   // [D][N] -> [N][D] per generation
            transpose_kernel<<<grid_for(N * c->D, 256), 256, 0, c->stream>>>(c->hist_x + k * N * c->D, c->stage, c->D, N);
            HIPC(c, hipGetLastError());
            HIPC(c, hipMemcpyAsync(x_saved + k * N * c->D, c->stage, sizeof(double) * N * c->D, hipMemcpyDeviceToHost,
                                   c->stream));
        }
    }
    HIPC(c, stream_wait(c->stream));
    return 0;
}

// ---- in-library communicator (RCCL) ------------------------------------------------------------
#define NCCLC(c, call)                                                                           \
    do {                                                                                         \
        ncclResult_t r_ = (call);                                                                \
        if (r_ != ncclSuccess) {                                                                 \
            (c)->err = std::string(#call) + ": " + rccl().GetErrorString(r_);                    \
            return -4;                                                                           \
        }                                                                                        \
    } while (0)

int smcn_comm_unique_id(char out[128]) {
    if (!out) return -1;
    if (!rccl().ok) { g_create_error = "smcn_comm_unique_id: " + rccl().why; return -4; }
    ncclUniqueId id;
    if (rccl().GetUniqueId(&id) != ncclSuccess) { g_create_error = "ncclGetUniqueId failed"; return -4; }
    memcpy(out, id.internal, NCCL_UNIQUE_ID_BYTES);
    return 0;
}

int smcn_comm_init(smcn_ctx* c, int rank, int world, const char id_bytes[128]) {
    CHECK_CTX(c);
    if (!id_bytes || world < 1 || rank < 0 || rank >= world) FAIL(c, "smcn_comm_init: bad arguments");
    if (!rccl().ok) FAIL(c, "smcn_comm_init: " + rccl().why);
    if (c->comm) { (void)rccl().CommDestroy(c->comm); c->comm = nullptr; }
    ncclUniqueId id;
    memcpy(id.internal, id_bytes, NCCL_UNIQUE_ID_BYTES);
    {
        const ncclResult_t r_ = rccl().CommInitRank(&c->comm, world, id, rank);
        if (r_ != ncclSuccess) {
            c->comm = nullptr;       // whatever a failed init left there is not a communicator (never pass it to CommDestroy)
            FAIL(c, std::string("smcn_comm_init: ncclCommInitRank: ") + rccl().GetErrorString(r_));
        }
    }
    c->comm_rank = rank;
    c->comm_world = world;
    return 0;
}

// What the COMMUNICATOR says about itself (not what the launcher's environment says): out = [ranks in the communicator
// (ncclCommCount), this rank in it (ncclCommUserRank), RCCL's version code]; -1 each without a communicator.
int smcn_comm_info(smcn_ctx* c, int out[3]) {
    CHECK_CTX(c);
    if (!out) FAIL(c, "smcn_comm_info: null");
    out[0] = out[1] = out[2] = -1;
    if (!c->comm) return 0;
    NCCLC(c, rccl().CommCount(c->comm, &out[0]));
    NCCLC(c, rccl().CommUserRank(c->comm, &out[1]));
    NCCLC(c, rccl().GetVersion(&out[2]));
    return 0;
}

int smcn_comm_destroy(smcn_ctx* c) {
    CHECK_CTX(c);
    if (c->comm) {
        HIPC(c, stream_wait(c->stream));
        NCCLC(c, rccl().CommDestroy(c->comm));
        c->comm = nullptr;
    }
    return 0;
}

// dst[world][n] <- all-gather(src[n]) (fp64, device pointers), in this context's stream
int smcn_comm_allgather(smcn_ctx* c, const void* src, void* dst, int64_t n) {
    CHECK_CTX(c);
    if (!c->comm) FAIL(c, "smcn_comm_allgather: no communicator (smcn_comm_init)");
    NCCLC(c, rccl().AllGather(src, dst, (size_t)n, ncclFloat64, c->comm, c->stream));
    return 0;
}

// all-to-all with per-peer counts (items of `elem` doubles), device pointers, send / recv segments in rank order
int smcn_comm_alltoallv(smcn_ctx* c, const void* send, const int64_t* send_counts, void* recv, const int64_t* recv_counts,
                        int elem) {
    CHECK_CTX(c);
    if (!c->comm) FAIL(c, "smcn_comm_alltoallv: no communicator (smcn_comm_init)");
    const double* sp = (const double*)send;
    double* rp = (double*)recv;
    if (!send_counts || !recv_counts || elem < 1) FAIL(c, "smcn_comm_alltoallv: bad arguments");
    for (int p = 0; p < c->comm_world; ++p)
        if (send_counts[p] < 0 || recv_counts[p] < 0 || (send_counts[p] > 0 && !send) || (recv_counts[p] > 0 && !recv))
            FAIL(c, "smcn_comm_alltoallv: negative count or missing buffer");
    NCCLC(c, rccl().GroupStart());
    // an error inside the group must still CLOSE it: an open group would swallow every later collective of this thread
    ncclResult_t bad = ncclSuccess;
    const char* what = "";
    for (int p = 0; p < c->comm_world && bad == ncclSuccess; ++p) {
        if (send_counts[p] > 0) {
            bad = rccl().Send(sp, (size_t)(send_counts[p] * elem), ncclFloat64, p, c->comm, c->stream);
            what = "ncclSend";
        }
        if (bad == ncclSuccess && recv_counts[p] > 0) {
            bad = rccl().Recv(rp, (size_t)(recv_counts[p] * elem), ncclFloat64, p, c->comm, c->stream);
            what = "ncclRecv";
        }
        sp += send_counts[p] * elem;
        rp += recv_counts[p] * elem;
    }
    const ncclResult_t ge = rccl().GroupEnd();
    if (bad != ncclSuccess) FAIL(c, std::string("smcn_comm_alltoallv: ") + what + ": " + rccl().GetErrorString(bad));
    if (ge != ncclSuccess) FAIL(c, std::string("smcn_comm_alltoallv: ncclGroupEnd: ") + rccl().GetErrorString(ge));
    return 0;
}

// all-gather of a few host doubles (counts, tile totals, step partials of the step-by-step strategies)
int smcn_comm_allgather_host(smcn_ctx* c, const double* src, int64_t n, double* dst) {
    CHECK_CTX(c);
    if (!c->comm) FAIL(c, "smcn_comm_allgather_host: no communicator (smcn_comm_init)");
    if (!src || !dst || n < 1) FAIL(c, "smcn_comm_allgather_host: bad arguments");
    int rc = ensure_stage2(c, n * (c->comm_world + 1));
    if (rc) return rc;
    double* d_src = c->stage2;
    double* d_dst = c->stage2 + n;
    HIPC(c, hipMemcpyAsync(d_src, src, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    NCCLC(c, rccl().AllGather(d_src, d_dst, (size_t)n, ncclFloat64, c->comm, c->stream));
    HIPC(c, hipMemcpyAsync(dst, d_dst, sizeof(double) * n * c->comm_world, hipMemcpyDeviceToHost, c->stream));
    HIPC(c, stream_wait(c->stream));
    return 0;
}

// raw device buffer <-> host (communicators without a device path move the exchange buffers through the host)
int smcn_buf_get(smcn_ctx* c, const void* dev, int64_t n, double* host) {
    CHECK_CTX(c);
    HIPC(c, hipMemcpyAsync(host, dev, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
    HIPC(c, stream_wait(c->stream));
    return 0;
}
// device -> device, in the context's stream, not waited for (in-process shards on one GPU exchange their buffers this way:
// smcnuts_amd.parallel.InProcessComm)
int smcn_buf_copy(smcn_ctx* c, void* dst, const void* src, int64_t n) {
    CHECK_CTX(c);
    if (n < 0 || (n > 0 && (!dst || !src))) FAIL(c, "smcn_buf_copy: bad arguments");
    if (n > 0) HIPC(c, hipMemcpyAsync(dst, src, sizeof(double) * n, hipMemcpyDeviceToDevice, c->stream));
    return 0;
}
int smcn_buf_set(smcn_ctx* c, void* dev, int64_t n, const double* host) {
    CHECK_CTX(c);
    HIPC(c, hipMemcpyAsync(dev, host, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    HIPC(c, stream_wait(c->stream));
    return 0;
}

// ---- Samples._resample (samples.py:124-146) over SHARDS without gathering the population ---------------
// Every shard scans its own (globally normalised) weights in the blocked order; the tile totals of all shards
// (N_local / 1024 doubles each, all-gathered) give every rank the same tile offsets as one shard of N_total
// particles computes.  A rank then draws its N_local keys (Philox keyed by GLOBAL slot), finds the tile -- hence
// the owner rank -- of each, and the shards exchange keys and ancestor rows point to point:
//   smcn_gres_begin  -> all-gather of tile totals -> smcn_gres_plan -> (host: order by owner, counts)
//   -> smcn_gres_set_order -> all-to-all of keys -> smcn_gres_serve -> all-to-all of rows -> smcn_gres_finish
// Shards of a multiple of the scan tile (1024) reproduce one shard's blocked scan exactly.  Any other shard size works
// the same way with the last tile of every shard partial: the cdf is then summed in a different association than one
// shard of N_total would use, so an ancestor can differ from the one-shard run only where a key falls within rounding
// of a cdf step (probability ~ N^2 2^-53 per resampling) -- the all-gather of the whole population this replaces
// (2.1 GB per rank and event at BASELINE config 5) bought nothing else.
int smcn_gres_begin(smcn_ctx* c, int world, double* ttot_host) {
    CHECK_CTX(c);
    Range roctx_range("smcn:resample");
    const int64_t n = c->N;
    if (world < 1 || world > 64) FAIL(c, "smcn_gres_begin: 1..64 shards");
    const int nt = grid_for(n, kScanTile);
    if (c->g_world != world) {
        HIPC(c, stream_wait(c->stream));
        for (void** q : {(void**)&c->g_ttot_all, (void**)&c->g_toff_all, (void**)&c->g_keys, (void**)&c->g_keys_send,
                         (void**)&c->g_rows_recv, (void**)&c->g_dest, (void**)&c->g_order}) {
            if (*q) (void)cached_free(*q);
            *q = nullptr;
        }
        HIPC(c, dalloc(&c->g_ttot_all, (int64_t)world * nt));
        HIPC(c, dalloc(&c->g_toff_all, (int64_t)world * nt + 1));
        HIPC(c, dalloc(&c->g_keys, n));
        HIPC(c, dalloc(&c->g_keys_send, n));
        HIPC(c, dalloc(&c->g_rows_recv, n * c->D));
        HIPC(c, dalloc(&c->g_dest, n));
        HIPC(c, dalloc(&c->g_order, n));
        c->g_world = world;
    }
    scan_tile_kernel<<<nt, 256, 0, c->stream>>>(c->wn, n, c->scan_local, c->ttot);
    HIPC(c, hipGetLastError());
    if (ttot_host) {
        HIPC(c, hipMemcpyAsync(ttot_host, c->ttot, sizeof(double) * nt, hipMemcpyDeviceToHost, c->stream));
        HIPC(c, stream_wait(c->stream));
    }
    return 0;
}

// device buffers of the exchange: this shard's tile totals [nt], all shards' [world][nt], keys in send order [n],
// keys to serve [cap], rows served [cap][D], rows received [n][D]
int smcn_gres_buffers(smcn_ctx* c, void** ttot_local, void** ttot_all, void** keys_send, void** keys_recv, void** rows_send,
                      void** rows_recv) {
    CHECK_CTX(c);
    if (c->g_world < 1) FAIL(c, "smcn_gres_buffers: call smcn_gres_begin first");
    if (ttot_local) *ttot_local = c->ttot;
    if (ttot_all) *ttot_all = c->g_ttot_all;
    if (keys_send) *keys_send = c->g_keys_send;
    if (keys_recv) *keys_recv = c->g_keys_recv;
    if (rows_send) *rows_send = c->g_rows_send;
    if (rows_recv) *rows_recv = c->g_rows_recv;
    return 0;
}

int smcn_gres_reserve(smcn_ctx* c, int64_t m) {   // room to serve m requests
    CHECK_CTX(c);
    if (m < 0) FAIL(c, "smcn_gres_reserve: negative request count");
    if (m < 1) m = 1;     // a rank that serves nothing still owns (tiny) buffers: communicators alias them by address
    if (m > c->g_serve_cap) {
        HIPC(c, stream_wait(c->stream));
        if (c->g_keys_recv) (void)cached_free(c->g_keys_recv);
        if (c->g_rows_send) (void)cached_free(c->g_rows_send);
        c->g_keys_recv = c->g_rows_send = nullptr;
        HIPC(c, dalloc(&c->g_keys_recv, m));
        HIPC(c, dalloc(&c->g_rows_send, m * c->D));
        c->g_serve_cap = m;
    }
    return 0;
}
}  // extern "C"

// key of every local slot and the GLOBAL tile that holds its ancestor (first tile whose last cdf value exceeds the key)
__global__ void gres_plan_kernel(const double* toff_all, int nt_all, int64_t n, int64_t n_total, uint64_t seed, uint32_t iter,
                                 int64_t particle_base, int scheme, double* keys, int32_t* tile) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double total = toff_all[nt_all];
    const double key = resample_key(scheme, nullptr, i, particle_base + i, n_total, 0, seed, iter);
    int lo = 0, hi = nt_all;
    while (lo < hi) {
        const int mid = lo + ((hi - lo) >> 1);
        const double cv = toff_all[mid + 1] / total;      // = cdf of the tile's last element
        if (key < cv) hi = mid;
        else lo = mid + 1;
    }
    keys[i] = key;
    tile[i] = lo < nt_all ? lo : nt_all - 1;
}
__global__ void gres_permute_kernel(const double* keys, const int32_t* order, int64_t n, double* keys_send) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) keys_send[k] = keys[order[k]];
}
// owner side: the ancestor of every received key inside this shard's tiles, and its row
__global__ void gres_serve_kernel(const double* keys, int64_t m, const double* toff_all, int nt_all, int tile0, int nt,
                                  const double* local, int64_t n, const double* x, int D, double* rows) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= m) return;
    const double total = toff_all[nt_all];
    const double key = keys[k];
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = lo + ((hi - lo) >> 1);
        const double cv = (toff_all[tile0 + mid / kScanTile] + local[mid]) / total;
        if (key < cv) hi = mid;
        else lo = mid + 1;
    }
    const int64_t src = lo < n ? lo : n - 1;
    for (int d = 0; d < D; ++d) rows[k * D + d] = x[(int64_t)d * n + src];
}
__global__ void gres_scatter_kernel(const double* rows, const int32_t* order, int64_t n, int D, double* x_out, double* logw,
                                    double logw_value) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int64_t i = order[k];
    for (int d = 0; d < D; ++d) x_out[(int64_t)d * n + i] = rows[k * D + d];
    logw[i] = logw_value;
}

extern "C" {
// after the all-gather of tile totals (ttot_all_host != NULL: uploaded here; NULL: already in the device buffer):
// tile offsets over all shards, this shard's keys, and for every local slot the rank that owns its ancestor
int smcn_gres_plan(smcn_ctx* c, int world, int rank, const double* ttot_all_host, int64_t iteration, int32_t* dest_host) {
    CHECK_CTX(c);
    if (c->g_world != world || !dest_host) FAIL(c, "smcn_gres_plan: call smcn_gres_begin first");
    const int64_t n = c->N;
    const int nt = grid_for(n, kScanTile), nt_all = nt * world;
    if (ttot_all_host)
        HIPC(c, hipMemcpyAsync(c->g_ttot_all, ttot_all_host, sizeof(double) * nt_all, hipMemcpyHostToDevice, c->stream));
    scan_offsets_kernel<<<1, 64, 0, c->stream>>>(c->g_ttot_all, nt_all, c->g_toff_all);
    gres_plan_kernel<<<grid_for(n, 256), 256, 0, c->stream>>>(c->g_toff_all, nt_all, n, n * world, c->seed, (uint32_t)iteration,
                                                              c->base, c->resample_scheme, c->g_keys, c->g_dest);
    HIPC(c, hipGetLastError());
    std::vector<int32_t> tile((size_t)n);
    HIPC(c, hipMemcpyAsync(tile.data(), c->g_dest, sizeof(int32_t) * n, hipMemcpyDeviceToHost, c->stream));
    HIPC(c, stream_wait(c->stream));
    for (int64_t i = 0; i < n; ++i) dest_host[i] = tile[(size_t)i] / nt;
    (void)rank;
    return 0;
}
int smcn_gres_set_order(smcn_ctx* c, const int32_t* order) {   // order[k] = local slot of the k-th key in send order
    CHECK_CTX(c);
    if (!order || c->g_world < 1) FAIL(c, "smcn_gres_set_order: bad arguments");
    const int64_t n = c->N;
    HIPC(c, hipMemcpyAsync(c->g_order, order, sizeof(int32_t) * n, hipMemcpyHostToDevice, c->stream));
    gres_permute_kernel<<<grid_for(n, 256), 256, 0, c->stream>>>(c->g_keys, c->g_order, n, c->g_keys_send);
    HIPC(c, hipGetLastError());
    HIPC(c, stream_wait(c->stream));
    return 0;
}
int smcn_gres_serve(smcn_ctx* c, int world, int rank, int64_t m) {
    CHECK_CTX(c);
    if (c->g_world != world || m < 0 || m > c->g_serve_cap) FAIL(c, "smcn_gres_serve: reserve first");
    const int64_t n = c->N;
    const int nt = grid_for(n, kScanTile);
    if (m > 0)
        gres_serve_kernel<<<grid_for(m, 256), 256, 0, c->stream>>>(c->g_keys_recv, m, c->g_toff_all, nt * world, rank * nt, nt,
                                                                   c->scan_local, n, c->x, c->D, c->g_rows_send);
    HIPC(c, hipGetLastError());
    return 0;
}
// x <- the received ancestor rows, logw <- loglik - log(N_total) (samples.py:140-143); loglik NULL: the device-resident
// loop's own value
int smcn_gres_finish(smcn_ctx* c, int world, const double* loglik) {
    CHECK_CTX(c);
    if (c->g_world != world) FAIL(c, "smcn_gres_finish: call smcn_gres_begin first");
    if (!loglik && c->fast_K < 0) FAIL(c, "smcn_gres_finish: loglik needed outside the device-resident loop");
    const int64_t n = c->N;
    double ll = 0.0;
    if (loglik) {
        ll = *loglik;
    } else {
        HIPC(c, hipMemcpyAsync(&ll, c->ss + SS_LL, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPC(c, stream_wait(c->stream));
    }
    gres_scatter_kernel<<<grid_for(n, 256), 256, 0, c->stream>>>(c->g_rows_recv, c->g_order, n, c->D, c->x_tmp, c->logw,
                                                                 ll - log((double)(n * world)));
    HIPC(c, hipGetLastError());
    std::swap(c->x, c->x_tmp);
    return 0;
}

int smcn_selftest_math(smcn_ctx* c, const double* x, int64_t n, double* out) {
    CHECK_CTX(c);
    if (!x || !out || n < 1) FAIL(c, "smcn_selftest_math: bad arguments");
    int rc = ensure_stage(c, n);
    if (rc) return rc;
    if ((rc = ensure_stage2(c, 11 * n))) return rc;
    HIPC(c, hipMemcpyAsync(c->stage, x, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    selftest_math_kernel<<<grid_for(n, 256), 256, 0, c->stream>>>(c->stage, n, c->stage2);
    HIPC(c, hipGetLastError());
    HIPC(c, hipMemcpyAsync(out, c->stage2, sizeof(double) * 11 * n, hipMemcpyDeviceToHost, c->stream));
    HIPC(c, stream_wait(c->stream));
    return 0;
}

int smcn_selftest_wide(smcn_ctx* c, int lanes, const double* x, int64_t n, double* out) {
    CHECK_CTX(c);
    if (c->model != SMCN_MODEL_ARMA || !c->lane_kernel) FAIL(c, "smcn_selftest_wide: arma contexts only");
    if (!x || !out || n < 1 || (lanes != 64 && lanes != 32 && lanes != 16 && lanes != 8 && lanes != 4)) FAIL(c, "smcn_selftest_wide: bad arguments");
    if (lanes >= 32 && c->arma_T < ArmaLaneModel::WIDE_MIN_T_ROWS) FAIL(c, "smcn_selftest_wide: 32 / 64 lanes per series take at least 130 observations");
    if (c->arma_T < ArmaLaneModel::WIDE_MIN_T || c->arma_T > ArmaLaneModel::YMAX)
        FAIL(c, "smcn_selftest_wide: the wide evaluation takes series of 64..384 observations");
    int rc = ensure_stage(c, 4 * n);
    if (rc) return rc;
    if ((rc = ensure_stage2(c, 8 * n))) return rc;
    HIPC(c, hipMemcpyAsync(c->stage, x, sizeof(double) * 4 * n, hipMemcpyHostToDevice, c->stream));
    const size_t lds = 8 * (ArmaLaneModel::YMAX + ArmaLaneModel::YPAD) + 16 * ArmaLaneModel::xch_pairs<4>();
    const int G = 64 / lanes;
    const int blocks = (int)((n + G - 1) / G);
    if (lanes == 64) selftest_wide_kernel<ArmaLaneModel, 64><<<blocks, 64, lds, c->stream>>>(c->mdata, c->stage, n, c->stage2);
    else if (lanes == 32) selftest_wide_kernel<ArmaLaneModel, 32><<<blocks, 64, lds, c->stream>>>(c->mdata, c->stage, n, c->stage2);
    else if (lanes == 16) selftest_wide_kernel<ArmaLaneModel, 16><<<blocks, 64, lds, c->stream>>>(c->mdata, c->stage, n, c->stage2);
    else if (lanes == 8) selftest_wide_kernel<ArmaLaneModel, 8><<<blocks, 64, lds, c->stream>>>(c->mdata, c->stage, n, c->stage2);
    else selftest_wide_kernel<ArmaLaneModel, 4><<<blocks, 64, lds, c->stream>>>(c->mdata, c->stage, n, c->stage2);
    HIPC(c, hipGetLastError());
    HIPC(c, hipMemcpyAsync(out, c->stage2, sizeof(double) * 8 * n, hipMemcpyDeviceToHost, c->stream));
    HIPC(c, stream_wait(c->stream));
    return 0;
}

}  // extern "C"
// ---- measured roofline denominators (bench.py: roofline.peak_measured) ---------------------------------------------
__global__ void __launch_bounds__(256) peak_copy_kernel(const double2* __restrict__ a, double2* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
// 16 independent fp64 FMA chains per lane, one wavefront per block; `waves` blocks share a SIMD through the LDS they ask for
__global__ void __launch_bounds__(64) peak_fma_kernel(double* out, int iters, double a, double b) {
    extern __shared__ double peak_pad[];
    double v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = a + i + threadIdx.x * 1e-9;
    if (iters < 0) peak_pad[threadIdx.x] = a;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = fma(v[i], a, b);
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    if (s == 12345.678) out[blockIdx.x * 64 + threadIdx.x] = s;
}
extern "C" {
// Measured on THIS device, now: out[0] = copy bandwidth (GB/s, 1 read + 1 write over 2 x 1 GiB), out[1] = fp64 FMA rate
// (TFLOP/s) with ONE wavefront per SIMD (the occupancy of the lane-per-particle NUTS kernel), out[2] = the same with four.
// ~0.3 s of device time after a short clock ramp.
int smcn_measure_peaks(smcn_ctx* c, double out[3]) {
    CHECK_CTX(c);
    if (!out) FAIL(c, "smcn_measure_peaks: null");
    struct Events {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        ~Events() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
    } ev;
    HIPC(c, hipEventCreate(&ev.e0));
    HIPC(c, hipEventCreate(&ev.e1));
    const size_t bytes = (size_t)1 << 30, n = bytes / sizeof(double2);
    struct Buf { void* p = nullptr; ~Buf() { if (p) (void)cached_free(p, false, /*keep=*/false); } } a, b;   // 2 x 1 GiB of scratch: back to the driver
    HIPC(c, cached_malloc((void**)&a.p, bytes));
    HIPC(c, cached_malloc((void**)&b.p, bytes));
    HIPC(c, hipMemsetAsync(a.p, 0, bytes, c->stream));
    float best = 1e30f;
    for (int rep = 0; rep < 8; ++rep) {
        HIPC(c, hipEventRecord(ev.e0, c->stream));
        peak_copy_kernel<<<c->num_cu * 8, 256, 0, c->stream>>>((const double2*)a.p, (double2*)b.p, n);
        HIPC(c, hipEventRecord(ev.e1, c->stream));
        HIPC(c, hipEventSynchronize(ev.e1));
        float ms = 0.f;
        HIPC(c, hipEventElapsedTime(&ms, ev.e0, ev.e1));
        if (rep > 1 && ms < best) best = ms;
    }
    out[0] = 2.0 * (double)bytes / best / 1e6;
    for (int wv = 0; wv < 2; ++wv) {
        const int waves = wv == 0 ? 1 : 4, iters = 40000;
        const size_t lds = waves == 1 ? 36 * 1024 : 8 * 1024;     // 4 / 16 one-wavefront blocks per CU
        HIPC(c, hipFuncSetAttribute((const void*)peak_fma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const int grid = c->num_cu * 4 * waves;
        best = 1e30f;
        for (int rep = 0; rep < 6; ++rep) {
            HIPC(c, hipEventRecord(ev.e0, c->stream));
            peak_fma_kernel<<<grid, 64, lds, c->stream>>>((double*)b.p, iters, 1.0000001, 1e-9);
            HIPC(c, hipEventRecord(ev.e1, c->stream));
            HIPC(c, hipEventSynchronize(ev.e1));
            float ms = 0.f;
            HIPC(c, hipEventElapsedTime(&ms, ev.e0, ev.e1));
            if (rep > 1 && ms < best) best = ms;
        }
        out[1 + wv] = (double)grid * iters * 16.0 * 64.0 * 2.0 / best / 1e9;
    }
    HIPC(c, hipGetLastError());
    return 0;
}

int smcn_debug_profile(smcn_ctx* c, uint64_t out[16], int reset) {
    CHECK_CTX(c);
    HIPC(c, stream_wait(c->stream));
    if (out) {
        HIPC(c, hipMemcpy(out, c->prof, sizeof(uint64_t) * 16, hipMemcpyDeviceToHost));
        unsigned int qv[4];
        HIPC(c, hipMemcpy(qv, c->queue, sizeof qv, hipMemcpyDeviceToHost));
        if (!c->lane_kernel) out[6] = qv[2];   // residency census of the last launch: max blocks alive at once
    }
    if (reset) HIPC(c, hipMemsetAsync(c->prof, 0, sizeof(uint64_t) * 16, c->stream));
    return 0;
}

int smcn_timers(smcn_ctx* c, double out[6], int reset) {
    CHECK_CTX(c);
    HIPC(c, stream_wait(c->stream));
    for (int i = 0; i < c->ev_n; ++i) {
        float ms = 0.f;
        HIPC(c, hipEventElapsedTime(&ms, c->ev0[i], c->ev1[i]));
        c->nuts_ms += ms;
        c->nuts_launches++;
    }
    c->ev_n = 0;
    if (out) {
        for (int i = 0; i < 6; ++i) out[i] = 0.0;
        out[0] = c->nuts_ms;
        out[1] = (double)c->nuts_launches;
    }
    if (reset) {
        c->nuts_ms = 0.0;
        c->nuts_launches = 0;
    }
    return 0;
}

}  // extern "C"
