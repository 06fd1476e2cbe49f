#!/bin/bash
# config 5 (iso-Gaussian D = 256, N = 131072 on one GPU): HBM traffic and SQ counters of the NUTS kernel
#   tools/pmc_c5.sh <tag> <step size>
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/c5_$1_$2
mkdir -p $OUT
ARGS="--config c5 --steps 6 --warmup 2 --step-size $2 --repeats 1"
python3 bench.py $ARGS > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
run() { n=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT -o $n -- python3 bench.py $ARGS > $OUT/$n.log 2>&1 || { tail -5 $OUT/$n.log; exit 1; }; }
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- python3 bench.py $ARGS > $OUT/stats.json 2> $OUT/stats.err
run fetch FETCH_SIZE
run write WRITE_SIZE
run sqa SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run sqb SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_FMA_F64
run sqc SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU
python3 - $OUT <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
tot = collections.OrderedDict()
for f in sorted(glob.glob(out + "/*counter_collection.csv")):
    rows = [r for r in csv.DictReader(open(f)) if "nuts_kernel" in r["Kernel_Name"] or "nuts_wave_kernel" in r["Kernel_Name"]]
    per = collections.defaultdict(dict)
    for r in rows:
        per[int(r["Dispatch_Id"])][r["Counter_Name"]] = per[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    ids = sorted(per)
    timed = ids[2:]                    # 2 warm-up launches, then the timed ones
    for cn in per[ids[0]]:
        tot[cn] = sum(per[i][cn] for i in timed) / len(timed)
line = json.loads(open(out + "/bench.json").read().strip().splitlines()[-1])
with open(out + "/summary.txt", "w") as fh:
    print(f"# nuts_wave_kernel<GaussModel> (nuts_kernel<GaussModel> before round 5), mean over the {len(timed)} timed launches; bench: {line['value']/1e9:.4f} G leapfrog/s, "
          f"{line['leapfrogs_per_particle_step']:.1f} leapfrogs per particle-step, kernel {line['roofline']['avg_launch_ms']:.3f} ms, "
          f"NUTS share of step {line['nuts_kernel_share_of_step']:.3f}", file=fh)
    for k, v in tot.items():
        print(f"{k:28s} {v:.4e}", file=fh)
    if "FETCH_SIZE" in tot:
        b = tot["FETCH_SIZE"] * 2048 + tot["WRITE_SIZE"] * 1024
        print(f"hbm_bytes_per_launch (FETCH x2 + WRITE) {b:.4e}  -> {b / line['roofline']['avg_launch_ms'] / 1e6:.1f} GB/s", file=fh)
print(open(out + "/summary.txt").read())
PY
