#!/usr/bin/env python3
"""Generates misc_issue.hip: cycles per instruction of ONE wavefront per SIMD for the instruction classes of the NUTS
bookkeeping (selects, 64-bit moves, integer multiplies of Philox, AGPR moves, exec-mask branches, LDS round trips)."""
import sys
pats = {}
pats["cndmask_b32"] = [f"v_cndmask_b32_e32 v{10 + i}, v2, v4, vcc" for i in range(8)] * 3
pats["cndmask_e64"] = [f"v_cndmask_b32_e64 v{10 + i}, v2, v4, s[20:21]" for i in range(8)] * 3
pats["mov_b32"] = [f"v_mov_b32_e32 v{10 + i}, v2" for i in range(8)] * 3
pats["mov_b64"] = [f"v_mov_b64_e32 v[{10 + 2 * i}:{11 + 2 * i}], v[2:3]" for i in range(8)] * 3
pats["xor_b32"] = [f"v_xor_b32_e32 v{10 + i}, v2, v{10 + i}" for i in range(8)] * 3
pats["add_u32"] = [f"v_add_u32_e32 v{10 + i}, v2, v{10 + i}" for i in range(8)] * 3
pats["mul_lo_u32"] = [f"v_mul_lo_u32 v{10 + i}, v2, v{10 + i}" for i in range(8)] * 3
pats["mul_hi_u32"] = [f"v_mul_hi_u32 v{10 + i}, v2, v{10 + i}" for i in range(8)] * 3
pats["mad_u64_u32"] = [f"v_mad_u64_u32 v[{10 + 2 * i}:{11 + 2 * i}], s[22:23], v2, v4, v[{10 + 2 * i}:{11 + 2 * i}]" for i in range(8)] * 3
pats["mad_u64_u32_0"] = [f"v_mad_u64_u32 v[{10 + 2 * i}:{11 + 2 * i}], s[22:23], v2, v{10 + 2 * i}, 0" for i in range(8)] * 3
pats["accvgpr_write"] = [f"v_accvgpr_write_b32 a{i}, v2" for i in range(8)] * 3
pats["accvgpr_read"] = [f"v_accvgpr_read_b32 v{10 + i}, a{i}" for i in range(8)] * 3
pats["cmp_f64"] = [f"v_cmp_lt_f64_e32 vcc, v[2:3], v[{10 + 2 * i}:{11 + 2 * i}]" for i in range(8)] * 3
pats["cmp_u32"] = [f"v_cmp_lt_u32_e32 vcc, v2, v{10 + i}" for i in range(8)] * 3
pats["cvt_f64_u32"] = [f"v_cvt_f64_u32_e32 v[{10 + 2 * i}:{11 + 2 * i}], v2" for i in range(8)] * 3
pats["cndmask_e64_vcc"] = [f"v_cndmask_b32_e64 v{10 + i}, v2, v4, vcc" for i in range(8)] * 3
pats["cndmask_e32_mixmov"] = [x for i in range(12) for x in (f"v_cndmask_b32_e32 v{10 + i % 8}, v2, v4, vcc", f"v_mov_b32_e32 v{20 + i % 8}, v2")]
pats["cndmask_e32_samedst"] = ["v_cndmask_b32_e32 v10, v2, v4, vcc"] * 24
pats["cndmask_e32_chain"] = [f"v_cndmask_b32_e32 v{10 + i}, v2, v{10 + i}, vcc" for i in range(8)] * 3
pats["cmp_f64_e64_sgpr"] = [f"v_cmp_lt_f64_e64 s[24:25], v[2:3], v[{10 + 2 * i}:{11 + 2 * i}]" for i in range(8)] * 3
pats["cmp_f64_e64_vcc"] = [f"v_cmp_lt_f64_e64 vcc, v[2:3], v[{10 + 2 * i}:{11 + 2 * i}]" for i in range(8)] * 3
pats["cmp_u32_e64_sgpr"] = [f"v_cmp_lt_u32_e64 s[24:25], v2, v{10 + i}" for i in range(8)] * 3
pats["cmp_mixmov"] = [x for i in range(12) for x in (f"v_cmp_lt_u32_e32 vcc, v2, v{10 + i % 8}", f"v_mov_b32_e32 v{20 + i % 8}, v2")]
pats["cmp_cnd_vcc"] = [x for i in range(12) for x in (f"v_cmp_lt_u32_e32 vcc, v2, v{10 + i % 8}", f"v_cndmask_b32_e32 v{20 + i % 8}, v2, v4, vcc")]
pats["cmp_cnd_sgpr"] = [x for i in range(12) for x in (f"v_cmp_lt_u32_e64 s[24:25], v2, v{10 + i % 8}", f"v_cndmask_b32_e64 v{20 + i % 8}, v2, v4, s[24:25]")]
pats["cmp_2cnd_vcc"] = [x for i in range(8) for x in (f"v_cmp_lt_u32_e32 vcc, v2, v{10 + i % 8}", f"v_cndmask_b32_e32 v{20 + i % 8}, v2, v4, vcc", f"v_cndmask_b32_e32 v{28 + i % 8}, v2, v4, vcc")]
pats["cmp_2cnd_sgpr"] = [x for i in range(8) for x in (f"v_cmp_lt_u32_e64 s[24:25], v2, v{10 + i % 8}", f"v_cndmask_b32_e64 v{20 + i % 8}, v2, v4, s[24:25]", f"v_cndmask_b32_e64 v{28 + i % 8}, v2, v4, s[24:25]")]
pats["addc_vcc"] = [f"v_addc_co_u32_e32 v{10 + i}, vcc, v2, v{10 + i}, vcc" for i in range(8)] * 3
pats["add_co_vcc"] = [f"v_add_co_u32_e32 v{10 + i}, vcc, v2, v{10 + i}" for i in range(8)] * 3
pats["fma_then_cmp"] = [x for i in range(12) for x in (f"v_fma_f64 v[{10 + 2 * (i % 4)}:{11 + 2 * (i % 4)}], v[2:3], v[4:5], v[{10 + 2 * (i % 4)}:{11 + 2 * (i % 4)}]", f"v_cmp_lt_f64_e32 vcc, v[2:3], v[{10 + 2 * (i % 4)}:{11 + 2 * (i % 4)}]")]
pats["salu_and"] = [f"s_and_b64 s[24:25], s[20:21], s[22:23]" for i in range(24)]
pats["saveexec_pair"] = ["s_and_saveexec_b64 s[24:25], s[26:27]", "s_or_b64 exec, exec, s[24:25]"] * 12
pats["valu_salu_mix"] = ["v_mov_b32_e32 v10, v2", "s_and_b64 s[24:25], s[20:21], s[22:23]"] * 12
pats["readlane"] = [f"v_readlane_b32 s24, v{10 + i}, 3" for i in range(8)] * 3
pats["readfirstlane"] = [f"v_readfirstlane_b32 s24, v{10 + i}" for i in range(8)] * 3
# not-taken and taken branches (forward, over nothing)
pats["branch_not_taken"] = ["s_cmp_eq_u32 s20, 0x12345", "s_cbranch_scc1 1f", "v_mov_b32_e32 v10, v2", "1:"] * 8
pats["branch_taken"] = ["s_cmp_lg_u32 s20, 0x12345", "s_cbranch_scc1 1f", "v_mov_b32_e32 v10, v2", "1:"] * 8
pats["execz_not_taken"] = ["s_and_saveexec_b64 s[24:25], s[26:27]", "s_cbranch_execz 1f", "v_mov_b32_e32 v10, v2", "1:", "s_or_b64 exec, exec, s[24:25]"] * 6
# LDS: a dependent b128 round trip, and 5 back-to-back reads + one wait
pats["ds_read_b128_rt"] = ["ds_read_b128 v[10:13], v6", "s_waitcnt lgkmcnt(0)"] * 6
pats["ds_read_b128_x5"] = ["ds_read_b128 v[10:13], v6", "ds_read_b128 v[14:17], v6 offset:1024", "ds_read_b128 v[18:21], v6 offset:2048",
                           "ds_read_b128 v[22:25], v6 offset:3072", "ds_read_b128 v[26:29], v6 offset:4096", "s_waitcnt lgkmcnt(0)"] * 3
pats["ds_write_b128_x5"] = ["ds_write_b128 v6, v[10:13]", "ds_write_b128 v6, v[14:17] offset:1024", "ds_write_b128 v6, v[18:21] offset:2048",
                            "ds_write_b128 v6, v[22:25] offset:3072", "ds_write_b128 v6, v[26:29] offset:4096"] * 3
pats["ds_write_then_read"] = ["ds_write_b128 v6, v[10:13]", "ds_read_b128 v[14:17], v6", "s_waitcnt lgkmcnt(0)"] * 4
names = list(pats)
def count(p):   # labels are not instructions
    return sum(1 for x in p if not x.endswith(":"))
out = ['// GENERATED by gen_misc_issue.py -- cycles per instruction, one wavefront per SIMD, non-fp64 instruction classes',
       '#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <vector>', '#include <algorithm>']
clob = ", ".join(f'"v{i}"' for i in range(2, 40)) + ", " + ", ".join(f'"a{i}"' for i in range(8)) + ', "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "vcc", "scc", "memory"'
NL = "\\n\\t"
init = " ".join('"v_mov_b32 v%d, 0x%x' % (r, 0x9E3779B1 * (r + 1) & 0x7fffffff) + NL + '"' for r in range(2, 40))
for i, n in enumerate(names):
    body = NL.join(pats[n])
    out.append("""__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) k%d(unsigned long long* cyc, int iters, double a) {
    extern __shared__ double pad[];
    if (iters < 0) pad[threadIdx.x] = a;
    asm volatile("s_mov_b32 s20, 0x55555555\\n\\ts_mov_b32 s21, 0x33333333\\n\\ts_mov_b32 s22, 0x0f0f0f0f\\n\\ts_mov_b32 s23, 0x00ff00ff\\n\\ts_mov_b32 s26, 0xffff0000\\n\\ts_mov_b32 s27, 0x0000ffff\\n\\t" %s "v_lshlrev_b32 v6, 4, v0\\n\\ts_nop 0" ::: %s);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) asm volatile("%s" ::: %s);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}""" % (i, init, clob, body, clob))
out.append('int main() {\n    unsigned long long* dc; (void)hipMalloc(&dc, 1024 * 8);\n    const int iters = 20000; const size_t lds = 36 * 1024;\n    std::vector<unsigned long long> c(1024);')
for i, n in enumerate(names):
    out.append(f'''    (void)hipFuncSetAttribute((const void*)k{i}, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 3; ++rep) k{i}<<<1024, 64, lds>>>(dc, iters, 1.0);
    (void)hipDeviceSynchronize(); (void)hipMemcpy(c.data(), dc, 1024 * 8, hipMemcpyDeviceToHost); std::sort(c.begin(), c.end());
    printf("%-18s %2d instr/iter: %.2f cycles per instruction (median wave; max %.2f)\\n", "{n}", {count(pats[n])}, (double)c[512] / iters / {count(pats[n])}, (double)c[1023] / iters / {count(pats[n])});''')
out.append('    return 0;\n}')
open(sys.argv[1] if len(sys.argv) > 1 else "misc_issue.hip", "w").write("\n".join(out) + "\n")
