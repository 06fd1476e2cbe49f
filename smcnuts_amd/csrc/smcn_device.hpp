// Device-side building blocks for gfx950 (wave64): lane-group cross-lane ops
// on DPP, Philox4x32-10, small helpers.  A "group" is G consecutive lanes of a
// wavefront (G a power of two, 1..64) that together own one particle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace smcn {

constexpr double kLog2Pi = 1.8378770664093454835606594728112;
constexpr double kLogPi = 1.1447298858494001741434273513531;
constexpr double kInf = __builtin_huge_val();

// ---- DPP moves of a double (two dword moves) ------------------------------
// CTRL: 0x00-0xFF quad_perm, 0x101-0x10F row_shl, 0x111-0x11F row_shr,
//       0x140 row_mirror, 0x141 row_half_mirror.  Lanes whose source is out of
//       the 16-lane row read 0 (bound_ctrl).
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// Butterfly all-reduce over the G lanes of a group.  Every stage adds a lane's
// value and its partner's; fp add is commutative, so all lanes of the group
// end with bit-identical sums.
// v + (v of lane ^ 16) and v + (v of lane ^ 32) without the LDS crossbar: gfx950's v_permlane16_swap /
// v_permlane32_swap exchange the upper half-rows (rows) of one operand with the lower ones of the other; fed the same
// value twice, the two results hold (own, partner) in the lower lanes and (partner, own) in the upper ones, so their
// sum is own + partner in every lane -- bit for bit what `v += __shfl_xor(v, 16 | 32)` gives (fp add commutes), at the
// latency of two VALU moves instead of two ds_bpermute round trips.
__device__ __forceinline__ double xor16_add(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
__device__ __forceinline__ double xor32_add(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
template <int G>
__device__ __forceinline__ double group_sum(double v) {
    if constexpr (G >= 2) v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]  (xor 1)
    if constexpr (G >= 4) v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]  (xor 2)
    if constexpr (G >= 8) v += dpp_mov<0x141>(v);   // row_half_mirror      (i <-> 7-i)
    if constexpr (G >= 16) v += dpp_mov<0x140>(v);  // row_mirror           (i <-> 15-i)
    if constexpr (G >= 32) v = xor16_add(v);
    if constexpr (G >= 64) v = xor32_add(v);
    return v;
}
// Sums of TWO / FOUR values over the 64 lanes of a wavefront in one butterfly.  v_permlane32_swap(x, y) leaves
// (x's lower half, y's lower half) in x and (x's upper half, y's upper half) in y, so x + y holds a[l] + a[l + 32] in
// lanes 0-31 and b[l - 32] + b[l] in lanes 32-63: the first stage of both sums for the price of one.  The 16-lane
// stage does the same with rows, after which every row carries ONE value's partials through the four DPP stages:
//   wave_sum2: 3 + 3 + 12 = 18 instructions (two separate butterflies: 40);   wave_sum4: 6 + 3 + 12 = 21 (80).
// The totals come back wave-uniform (v_readlane of the row that holds them).
__device__ __forceinline__ double lane_value(double v, int lane) {   // lane: compile-time constant
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double swap32_add(double a, double b) {   // lanes 0-31: a[l] + a[l+32]; lanes 32-63: b[l-32] + b[l]
    const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ double swap16_add(double a, double b) {   // rows 0, 2: a's row + its odd neighbour; rows 1, 3: b's
    const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ double row_sum16(double v) {              // every lane: the sum over its 16-lane row
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    return v;
}
__device__ __forceinline__ void wave_sum2(double a, double b, double& A, double& B) {
    double v = swap32_add(a, b);          // halves: a | b
    v = xor16_add(v);
    v = row_sum16(v);
    A = lane_value(v, 0);
    B = lane_value(v, 32);
}
__device__ __forceinline__ void wave_sum4(double a, double b, double c, double d, double& A, double& B, double& C, double& D) {
    const double p = swap32_add(a, b);    // halves: a | b
    const double q = swap32_add(c, d);    // halves: c | d
    double v = swap16_add(p, q);          // rows: a, c, b, d
    v = row_sum16(v);
    A = lane_value(v, 0);
    C = lane_value(v, 16);
    B = lane_value(v, 32);
    D = lane_value(v, 48);
}
// the same butterfly with the last two stages through ds_bpermute (selftest reference for the swaps above)
__device__ __forceinline__ double wave_sum_bpermute(double v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// value of lane (lg - K) of the same group, 0.0 for lg < K
template <int G, int K>
__device__ __forceinline__ double group_shift_up(double v, int lg) {
    double s;
    if constexpr (G <= 16) s = dpp_mov<0x110 + K>(v);
    else s = __shfl_up(v, K, 64);
    return lg >= K ? s : 0.0;
}
// value of lane (lg + K) of the same group, 0.0 for lg + K >= G
template <int G, int K>
__device__ __forceinline__ double group_shift_down(double v, int lg) {
    double s;
    if constexpr (G <= 16) s = dpp_mov<0x100 + K>(v);
    else s = __shfl_down(v, K, 64);
    return lg + K < G ? s : 0.0;
}

// value held by lane `src` (0..G-1) of the caller's group
// (G == 64: the group is the wavefront and `src` is the same in every lane -- a v_readlane with a scalar lane index
// instead of a trip through the LDS crossbar)
template <int G>
__device__ __forceinline__ double group_read(double v, int src) {
    if constexpr (G == 1) return v;
    if constexpr (G == 64) {
        const int sl = __builtin_amdgcn_readfirstlane(src);
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), sl),
                                __builtin_amdgcn_readlane(__double2loint(v), sl));
    }
    const int lane = (int)(threadIdx.x & 63u);
    return __shfl(v, (lane & ~(G - 1)) + src, 64);
}
// value held by lane `src` (0..3, compile-time after unrolling) of the caller's quad: one DPP move per dword
__device__ __forceinline__ double quad_read(double v, int src) {
    switch (src & 3) {
        case 0: return dpp_mov<0x00>(v);
        case 1: return dpp_mov<0x55>(v);
        case 2: return dpp_mov<0xAA>(v);
        default: return dpp_mov<0xFF>(v);
    }
}
template <int G>
__device__ __forceinline__ int group_read_i(int v, int src) {
    if constexpr (G == 1) return v;
    if constexpr (G == 64) return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(src));
    const int lane = (int)(threadIdx.x & 63u);
    return __shfl(v, (lane & ~(G - 1)) + src, 64);
}

// Lanes of one wavefront exchanging data through LDS (or global memory): the hardware executes a
// wave's memory instructions in order, but the COMPILER orders accesses per thread only -- a read
// of what ANOTHER lane wrote may legally be scheduled above that write.  This makes the exchange
// explicit: release by the writers, acquire by the readers, no instruction reordering across it.
// dst = src as ONE v_mov_b64 that the optimiser cannot turn back into a select: for use under a branch, where a group
// of 64-bit values moves at one instruction each (a select is two v_cndmask_b32, and runs of those on VCC issue at ~17
// cycles apiece for a lone wavefront: tools/ubench/misc_issue).  The join's phi makes t and dst one register.
__device__ __forceinline__ void mov64_under_branch(double& dst, double src) {
    double t;
    asm volatile("v_mov_b64_e32 %0, %1" : "=v"(t) : "v"(src));
    dst = t;
}

__device__ __forceinline__ void wave_exchange_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- Philox4x32-10 --------------------------------------------------------
struct u32x4 { uint32_t a, b, c, d; };

__host__ __device__ __forceinline__ u32x4 philox4x32_10(u32x4 ctr, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * ctr.a;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * ctr.c;
        u32x4 n;
        n.a = (uint32_t)(p1 >> 32) ^ ctr.b ^ k0;
        n.b = (uint32_t)p1;
        n.c = (uint32_t)(p0 >> 32) ^ ctr.d ^ k1;
        n.d = (uint32_t)p0;
        ctr = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return ctr;
}

// 53-bit uniform in [0,1): ((a>>5)*2^26 + (b>>6)) / 2^53
__host__ __device__ __forceinline__ double u53(uint32_t a, uint32_t b) {
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}

// Philox streams (counter word 3)
enum : uint32_t { kStreamNuts = 0, kStreamMomentum = 1, kStreamResample = 2, kStreamInit = 3 };

// draw q of (seed, iteration, particle, stream): block q>>1, half q&1
__host__ __device__ __forceinline__ double philox_uniform(uint64_t seed, uint32_t iter, uint32_t particle,
                                                          uint32_t stream, uint32_t q) {
    const u32x4 o = philox4x32_10({q >> 1, particle, iter, stream}, (uint32_t)seed, (uint32_t)(seed >> 32));
    return (q & 1u) ? u53(o.c, o.d) : u53(o.a, o.b);
}

// Resampling key of output slot `gi` of `n` (samples.py:139 draws n uniforms: multinomial).
// scheme 0: multinomial, an independent uniform per slot (the reference's rng.choice);
// scheme 1: systematic, ONE uniform u0 (the draw of slot `u0_slot`) and the comb (gi + u0) / n.
// `u` (tests): recorded uniforms, indexed by local slot `i`.
__device__ __forceinline__ double resample_key(int scheme, const double* u, int64_t i, int64_t gi, int64_t n,
                                               int64_t u0_slot, uint64_t seed, uint32_t iter) {
    if (scheme == 0) return u ? u[i] : philox_uniform(seed, iter, (uint32_t)gi, kStreamResample, 0u);
    const double u0 = u ? u[0] : philox_uniform(seed, iter, (uint32_t)u0_slot, kStreamResample, 0u);
    return ((double)(gi - u0_slot) + u0) / (double)n;
}

__device__ __forceinline__ bool finite_d(double v) { return __builtin_isfinite(v); }

constexpr int kRedBlock = 256;
// ---- block reductions (fixed order => run-to-run deterministic) -------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
// sum over a 256-thread block; result valid in every thread
__device__ __forceinline__ double block_sum(double v, double* sh /*>=4*/) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    return ((sh[0] + sh[1]) + sh[2]) + sh[3];
}
__device__ __forceinline__ double block_max(double v, double* sh) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    return fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
}


// ---- lean fp64 elementary functions for the per-leaf density evaluation ----------
// The device libm's log1p costs ~135 VALU instructions and exp ~40; one arma
// evaluation needs exp, log1p and two reciprocals, replicated on every lane.  These
// versions keep <= ~2 ulp (tests/test_gpu_parity.py::test_device_math) at ~1/3 of
// the instructions.  Non-finite / out-of-range inputs give non-finite outputs, which
// the caller maps to -inf as the reference's target adapter does.
__device__ __forceinline__ double rcp_nr(double a) {       // 1/a: hardware seed + 2 Newton steps
    double y = __builtin_amdgcn_rcp(a);
    double e = fma(-a, y, 1.0);
    y = fma(y, e, y);
    e = fma(-a, y, 1.0);
    return fma(y, e, y);
}
__device__ __forceinline__ double rsqrt_nr(double a) {     // 1/sqrt(a), a > 0: hardware seed + 2 Newton steps
    double y = __builtin_amdgcn_rsq(a);
    const double h = 0.5 * a;
    y = y * fma(-h * y, y, 1.5);
    return y * fma(-h * y, y, 1.5);
}
__device__ __forceinline__ double exp_fast(double t) {     // e^t, Taylor degree 12 on |r| <= ln2/2
    const double k = __builtin_rint(t * 1.4426950408889634074);
    double r = fma(-k, 6.93147180369123816490e-01, t);
    r = fma(-k, 1.90821492927058770002e-10, r);
    double p = 2.08767569878680989792e-09;                 // 1/12!
    p = fma(p, r, 2.50521083854417187751e-08);
    p = fma(p, r, 2.75573192239858906526e-07);
    p = fma(p, r, 2.75573192239858906526e-06);
    p = fma(p, r, 2.48015873015873015873e-05);
    p = fma(p, r, 1.98412698412698412698e-04);
    p = fma(p, r, 1.38888888888888888889e-03);
    p = fma(p, r, 8.33333333333333333333e-03);
    p = fma(p, r, 4.16666666666666666667e-02);
    p = fma(p, r, 1.66666666666666666667e-01);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)k);
}
// The same with the Horner constants as SCALAR operands: the compiler keeps them in VGPRs and emits v_mov_b64 + v_fmac_f64
// per step (the fused form wants its addend in the destination); v_fma_f64 takes one SGPR pair, so a step is one instruction.
__device__ __forceinline__ double exp_fast_s(double t) {   // bit-identical to exp_fast
    const double k = __builtin_rint(t * 1.4426950408889634074);
    double r = fma(-k, 6.93147180369123816490e-01, t);
    r = fma(-k, 1.90821492927058770002e-10, r);
    double p;
    // nine Horner steps as ONE block (between separate asm statements the hazard recogniser pads an s_nop each)
    asm("v_fma_f64 %0, %1, %2, %3\n\t"
        "v_fma_f64 %0, %0, %2, %4\n\t"
        "v_fma_f64 %0, %0, %2, %5\n\t"
        "v_fma_f64 %0, %0, %2, %6\n\t"
        "v_fma_f64 %0, %0, %2, %7\n\t"
        "v_fma_f64 %0, %0, %2, %8\n\t"
        "v_fma_f64 %0, %0, %2, %9\n\t"
        "v_fma_f64 %0, %0, %2, %10\n\t"
        "v_fma_f64 %0, %0, %2, %11"
        : "=&v"(p)
        : "v"(2.08767569878680989792e-09), "v"(r), "s"(2.50521083854417187751e-08), "s"(2.75573192239858906526e-07),
          "s"(2.75573192239858906526e-06), "s"(2.48015873015873015873e-05), "s"(1.98412698412698412698e-04),
          "s"(1.38888888888888888889e-03), "s"(8.33333333333333333333e-03), "s"(4.16666666666666666667e-02),
          "s"(1.66666666666666666667e-01));
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)k);
}
// log(u) for finite u >= 1:  u = m 2^e, m in [sqrt(1/2), sqrt 2), log m = 2 atanh((m-1)/(m+1))
__device__ __forceinline__ double log_ge1(double u) {
    int e = __builtin_amdgcn_frexp_exp(u);
    double m = __builtin_amdgcn_frexp_mant(u);
    const bool lo = m < 0.70710678118654752440;
    m = lo ? m + m : m;
    e = lo ? e - 1 : e;
    const double f = (m - 1.0) * rcp_nr(m + 1.0);
    const double f2 = f * f;
    double p = 9.52380952380952380952e-02;                  // 2/21
    p = fma(p, f2, 1.05263157894736842105e-01);             // 2/19
    p = fma(p, f2, 1.17647058823529411765e-01);
    p = fma(p, f2, 1.33333333333333333333e-01);
    p = fma(p, f2, 1.53846153846153846154e-01);
    p = fma(p, f2, 1.81818181818181818182e-01);
    p = fma(p, f2, 2.22222222222222222222e-01);
    p = fma(p, f2, 2.85714285714285714286e-01);
    p = fma(p, f2, 4.0e-01);
    p = fma(p, f2, 6.66666666666666666667e-01);             // 2/3
    const double logm = fma(f * f2, p, f + f);
    const double ed = (double)e;
    return fma(ed, 6.93147180369123816490e-01, fma(ed, 1.90821492927058770002e-10, logm));
}
// log1p(z) for z >= 0, and 1/(1+z) as a by-product
__device__ __forceinline__ double log1p_pos(double z, double& inv1pz) {
    const double u = 1.0 + z;
    inv1pz = rcp_nr(u);
    return fma(z - (u - 1.0), inv1pz, log_ge1(u));
}

// sin and cos of 2 pi u, u in [0, 1): the argument is reduced EXACTLY (quarter turns), then Taylor to x^15 / x^16 on
// |x| <= pi / 4 (truncation < 5e-17); ~45 instructions where the device library's sincospi takes ~150
__device__ __forceinline__ void sincos_2pi(double u, double& sn, double& cs) {
    const double t = 4.0 * u;
    const double k = __builtin_rint(t);
    const double x = (t - k) * 1.5707963267948966192;       // (t - k is exact)
    const double x2 = x * x;
    double s = -7.6471637318198164759e-13;                   // -1 / 15!
    s = fma(s, x2, 1.6059043836821614599e-10);
    s = fma(s, x2, -2.5052108385441718775e-08);
    s = fma(s, x2, 2.7557319223985890653e-06);
    s = fma(s, x2, -1.9841269841269841270e-04);
    s = fma(s, x2, 8.3333333333333333333e-03);
    s = fma(s, x2, -1.6666666666666666667e-01);
    s = fma(s * x2, x, x);
    double c = 4.7794773323873852974e-14;                    // 1 / 16!
    c = fma(c, x2, -1.1470745597729724714e-11);
    c = fma(c, x2, 2.0876756987868098979e-09);
    c = fma(c, x2, -2.7557319223985890653e-07);
    c = fma(c, x2, 2.4801587301587301587e-05);
    c = fma(c, x2, -1.3888888888888888889e-03);
    c = fma(c, x2, 4.1666666666666666667e-02);
    c = fma(c, x2, -0.5);
    c = fma(c, x2, 1.0);
    const int ki = (int)k & 3;                               // the angle is x + ki pi / 2
    sn = (ki & 1) ? c : s;
    cs = (ki & 1) ? s : c;
    cs = (ki == 1 || ki == 2) ? -cs : cs;
    sn = (ki >= 2) ? -sn : sn;
}
// The Box-Muller pair of (u1, u2) as normals_kernel computes it -- sqrt(-2 log1p(-u1)) (cos, sin)(2 pi u2) -- from the lean
// functions above (-log(1 - u1) = log1p(u1 / (1 - u1)), accurate for small u1; 1 - u1 is exact): the same values to ~2 ulp at
// half the instructions (the momentum draw of config 5 is bound by them, not by its 268 MB of stores)
__device__ __forceinline__ void box_muller_lean(double u1, double u2, double& z0, double& z1) {
    double inv;
    const double l2 = 2.0 * log1p_pos(u1 * rcp_nr(1.0 - u1), inv);
    const double rad = l2 > 0.0 ? l2 * rsqrt_nr(l2) : 0.0;
    double sn, cs;
    sincos_2pi(u2, sn, cs);
    z0 = rad * cs;
    z1 = rad * sn;
}

// log pi_phi with the target adapter's failure convention (bridgestan.py:45-49)
__device__ __forceinline__ double combine_lp(double lpri, double llik, double phi) {
    const double lp = lpri + phi * llik;
    return finite_d(lp) ? lp : -kInf;
}

}  // namespace smcn
