"""Copies the round-4 evidence from gpurun_out/ (scratch) into profiles/ (tracked): kernel-stats CSVs, the JSON lines of
the profiled runs, SQ counter summaries, section profiles, and profiles/r04_traffic.json (HBM bytes of the timed NUTS
launches, each entry stamped with the hash of the kernel sources it was measured on -- bench.py ignores stale ones)."""
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def cp(src, dst):
    s = os.path.join(G, src)
    if os.path.exists(s):
        shutil.copyfile(s, os.path.join(P, dst))
        print("  ", dst)
    else:
        print("   (missing)", src)


def line(path):
    return json.loads(open(os.path.join(G, path)).read().strip().splitlines()[-1])


entries = json.load(open(os.path.join(G, "r04_prof_arma", "traffic.json")))["entries"]
for eps, tag in ((0.25, "0.25"), (0.1, "0.1")):
    d = os.path.join("c5_r04_" + tag)
    try:
        b = line(os.path.join(d, "bench.json"))
        vals = {}
        for ln in open(os.path.join(G, d, "summary.txt")):
            f = ln.split()
            if len(f) == 2 and f[0] in ("FETCH_SIZE", "WRITE_SIZE"):
                vals[f[0]] = float(f[1])
        fb, wb = vals["FETCH_SIZE"] * 2048, vals["WRITE_SIZE"] * 1024
        entries.append(dict(config="c5", N=b["config"]["particles_per_gpu"], steps=6, warmup=2, fuse_max=1, step_size=eps,
                            kernel="nuts_kernel<GaussModel<64,4>,hbm_stack>", csrc_sha=b["roofline"]["csrc_sha"],
                            FETCH_SIZE_KB=vals["FETCH_SIZE"], WRITE_SIZE_KB=vals["WRITE_SIZE"], fetch_bytes_corrected_x2=fb,
                            write_bytes=wb, hbm_bytes_per_launch=fb + wb,
                            source=f"tools/pmc_c5.sh r04 {tag}: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- "
                                   f"python3 bench.py --config c5 --steps 6 --warmup 2 --step-size {tag} --repeats 1; mean over the timed "
                                   "launches; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md"))
    except Exception as e:       # noqa: BLE001
        print("c5", tag, "skipped:", e)
json.dump(dict(entries=entries), open(os.path.join(P, "r04_traffic.json"), "w"), indent=1)
print("profiles/r04_traffic.json:", [(e["config"], e["steps"], e.get("step_size"), e["csrc_sha"], round(e["hbm_bytes_per_launch"] / 1e6, 1)) for e in entries])
for k, w in ((20, 5), (50, 10)):
    cp(f"r04_prof_arma/stats_{k}_{w}_kernel_stats.csv", f"r04_a_bench_{k}_{w}_kernel_stats.csv")
    cp(f"r04_prof_arma/stats_{k}_{w}.json", f"r04_a_bench_{k}_{w}_under_rocprof.json")
cp("pmc_r04/summary.txt", "r04_a_pmc_sq_nuts3.txt")
cp("pmc_r04_n131072/summary.txt", "r04_a_pmc_sq_nuts3_queue_n131072.txt")
cp("r04_c4_ab.txt", "r04_c4_lanes_and_occupancy.txt")
cp("r04_sections.txt", "r04_a_nuts3_sections.txt")
cp("prof_r04_c4/stats_kernel_stats.csv", "r04_c4_kernel_stats.csv")
cp("prof_r04_c4/bench.json", "r04_c4_bench.json")
cp("prof_r04_c4/under_rocprof.json", "r04_c4_bench_under_rocprof.json")
cp("c4_r04/summary.txt", "r04_c4_pmc.txt")
for tag in ("0.25", "0.1"):
    cp(f"c5_r04_{tag}/summary.txt", f"r04_c5_eps{tag}_pmc.txt")
    cp(f"c5_r04_{tag}/stats_kernel_stats.csv", f"r04_c5_eps{tag}_kernel_stats.csv")
for f, d in (("bench_20_5.json", "r04_a_bench_20_5.json"), ("bench_50_10.json", "r04_a_bench_50_10.json"),
             ("bench_20_5_nowide.json", "r04_a_bench_20_5_one_lane_evaluation.json"), ("c5_025.json", "r04_c5_eps0.25_bench.json"),
             ("c5_01.json", "r04_c5_eps0.1_bench.json"), ("n_sweep.txt", "r04_a_n_sweep.txt"),
             ("experiments_arma.txt", "r04_experiments_arma.txt"), ("c4.json", "r04_c4_bench_final.json"),
             ("rehearsal_2ranks_gloo.json", "r04_rehearsal_2ranks_gloo.json"), ("ubench_mfma_f64.txt", "r04_ubench_mfma_f64.txt")):
    cp("r04_final/" + f, d)
