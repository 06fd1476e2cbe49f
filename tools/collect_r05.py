"""Copies the round-5 evidence from gpurun_out/ (scratch) into profiles/ (tracked): kernel-stats CSVs, the JSON lines of
the profiled runs, SQ counter summaries, and profiles/r05_traffic.json (HBM bytes of the timed NUTS launches, each entry
stamped with the hash of the kernel sources it was measured on -- bench.py ignores stale ones)."""
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


entries = json.load(open(os.path.join(G, "r05_prof_arma", "traffic.json")))["entries"]
for eps, tag in ((0.25, "0.25"), (0.1, "0.1")):
    d = os.path.join("c5_r05_" + tag)
    try:
        b = line(os.path.join(d, "bench.json"))
        vals = {}
        for ln in open(os.path.join(G, d, "summary.txt")):
            f = ln.split()
            if len(f) == 2 and f[0] in ("FETCH_SIZE", "WRITE_SIZE"):
                vals[f[0]] = float(f[1])
        fb, wb = vals["FETCH_SIZE"] * 2048, vals["WRITE_SIZE"] * 1024
        entries.append(dict(config="c5", N=b["config"]["particles_per_gpu"], steps=6, warmup=2, fuse_max=1, step_size=eps,
                            kernel="nuts_wave_kernel<GaussModel<64,4>,full,no_likelihood,3,3>", csrc_sha=b["roofline"]["csrc_sha"],
                            FETCH_SIZE_KB=vals["FETCH_SIZE"], WRITE_SIZE_KB=vals["WRITE_SIZE"], fetch_bytes_corrected_x2=fb,
                            write_bytes=wb, hbm_bytes_per_launch=fb + wb,
                            source=f"tools/pmc_c5.sh r05 {tag}: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- "
                                   f"python3 bench.py --config c5 --steps 6 --warmup 2 --step-size {tag} --repeats 1; mean over the timed "
                                   "launches; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md"))
    except Exception as e:       # noqa: BLE001
        print("c5", tag, "skipped:", e)
json.dump(dict(entries=entries), open(os.path.join(P, "r05_traffic.json"), "w"), indent=1)
print("profiles/r05_traffic.json:", [(e["config"], e["steps"], e.get("step_size"), e["csrc_sha"], round(e["hbm_bytes_per_launch"] / 1e6, 1)) for e in entries])
for k, w in ((20, 5), (50, 10)):
    cp(f"r05_prof_arma/stats_{k}_{w}_kernel_stats.csv", f"r05_a_bench_{k}_{w}_kernel_stats.csv")
    cp(f"r05_prof_arma/stats_{k}_{w}.json", f"r05_a_bench_{k}_{w}_under_rocprof.json")
cp("pmc_r05/summary.txt", "r05_a_pmc_sq_nuts3.txt")
cp("pmc_r05_n131072/summary.txt", "r05_a_pmc_sq_nuts3_queue_n131072.txt")
cp("prof_r05_c4/stats_kernel_stats.csv", "r05_c4_kernel_stats.csv")
cp("prof_r05_c4/bench.json", "r05_c4_bench.json")
cp("prof_r05_c4/under_rocprof.json", "r05_c4_bench_under_rocprof.json")
cp("c4_r05/summary.txt", "r05_c4_pmc.txt")
cp("c4_r05_g4/summary.txt", "r05_c4_pmc_four_lanes_per_particle.txt")
for tag in ("0.25", "0.1"):
    cp(f"c5_r05_{tag}/summary.txt", f"r05_c5_eps{tag}_pmc.txt")
    cp(f"c5_r05_{tag}/stats_kernel_stats.csv", f"r05_c5_eps{tag}_kernel_stats.csv")
for f, d in (("bench_20_5.json", "r05_a_bench_20_5.json"), ("bench_50_10.json", "r05_a_bench_50_10.json"),
             ("bench_20_5_nowide.json", "r05_a_bench_20_5_one_lane_evaluation.json"), ("c5_025.json", "r05_c5_eps0.25_bench.json"),
             ("c5_01.json", "r05_c5_eps0.1_bench.json"), ("n_sweep.txt", "r05_a_n_sweep.txt"),
             ("experiments_arma.txt", "r05_experiments_arma.txt"), ("c4.json", "r05_c4_bench_final.json"),
             ("rehearsal_2ranks_gloo.json", "r05_rehearsal_2ranks_gloo.json")):
    cp("r05_final/" + f, d)
