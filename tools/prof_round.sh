#!/bin/bash
# Round evidence for bench.py's numbers, run on the GPU box:  tools/prof_round.sh <round-tag, e.g. r02> [config]
#   1. rocprofv3 --kernel-trace --stats of the driver's command (kernel summary CSV + the JSON line of that run)
#   2. FETCH_SIZE / WRITE_SIZE of the timed NUTS launch in their own --pmc passes (kernel trace only)
# Outputs land in gpurun_out/<tag>_prof/; the summaries to commit are copied into profiles/ by the caller.
set -e
TAG=$1; CFG=${2:-arma}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${TAG}_prof_$CFG
mkdir -p $OUT
EXTRA="--no-end-to-end"; [ "$CFG" = "c5" ] && EXTRA="--config c5 ${C5ARGS:-}"
for cmd in "20 5" "50 10"; do set -- $cmd
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats_$1_$2 -- python3 bench.py --steps $1 --warmup $2 --no-cpu-baseline $EXTRA > $OUT/stats_$1_$2.json 2> $OUT/stats_$1_$2.err || { tail -5 $OUT/stats_$1_$2.err; exit 1; }
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT -o ${c}_$1_$2 -- python3 bench.py --steps $1 --warmup $2 --no-cpu-baseline $EXTRA > $OUT/${c}_$1_$2.json 2> $OUT/${c}_$1_$2.err || { tail -5 $OUT/${c}_$1_$2.err; exit 1; }
  done
done
python3 - $OUT $CFG <<'PY'
import csv, glob, json, sys, os
out, cfg = sys.argv[1], sys.argv[2]
entries = []
for steps, warm in ((20, 5), (50, 10)):
    vals = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        rows = list(csv.DictReader(open(f"{out}/{c}_{steps}_{warm}_counter_collection.csv")))
        nuts = [r for r in rows if "nuts" in r["Kernel_Name"] and "prep" not in r["Kernel_Name"] and "post" not in r["Kernel_Name"]
                and "host" not in r["Kernel_Name"]]
        # the timed launches: bench.py repeats the K iterations 5 times; take the last dispatch group
        last = max(int(r["Dispatch_Id"]) for r in nuts)
        v = sum(float(r["Counter_Value"]) for r in nuts if int(r["Dispatch_Id"]) == last)
        vals[c] = v
        kname = [r["Kernel_Name"] for r in nuts if int(r["Dispatch_Id"]) == last][0]
    line = json.loads(open(f"{out}/FETCH_SIZE_{steps}_{warm}.json").read().strip().splitlines()[-1])
    fetch_b = vals["FETCH_SIZE"] * 1024 * 2       # KB; gfx950 counts 128-B requests as 64 B (MI355X_MICROARCH.md, HBM)
    write_b = vals["WRITE_SIZE"] * 1024
    entries.append(dict(config=cfg, N=line["config"]["particles_per_gpu"], steps=steps, warmup=warm,
                        fuse_max=line["config"]["iterations_per_nuts_launch_max"], kernel=kname.split("(")[0],
                        csrc_sha=line["roofline"]["csrc_sha"],   # the kernel sources these bytes were measured on (bench.py drops stale entries)
                        FETCH_SIZE_KB=vals["FETCH_SIZE"], WRITE_SIZE_KB=vals["WRITE_SIZE"],
                        fetch_bytes_corrected_x2=fetch_b, write_bytes=write_b, hbm_bytes_per_launch=fetch_b + write_b,
                        launches_in_timed_region=line["roofline"]["launches"],
                        source=f"rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py "
                               f"--steps {steps} --warmup {warm} --no-cpu-baseline; last NUTS dispatch (the timed launch of the "
                               "last repeat); FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md"))
json.dump(dict(entries=entries), open(f"{out}/traffic.json", "w"), indent=1)
print(json.dumps(entries, indent=1))
PY
