#!/bin/bash
# config 4 (PRMwCD, Gaussian L-kernel + tempering, N = 65536): SQ counters of the NUTS kernel; tools/pmc_c4.sh <tag>
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/c4_$1
mkdir -p $OUT
ARGS="--config c4 --steps 6 --warmup 12 --repeats 1 --no-peaks $PMC_EXTRA"
run() { n=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT -o $n -- python3 bench.py $ARGS > $OUT/$n.log 2>&1 || { tail -5 $OUT/$n.log; exit 1; }; }
run sqa SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run sqb SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_FMA_F64
run sqc SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU
python3 - $OUT <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
# one SMC step launches the 8-lane kernel and (two-phase launches, the default for the shipped data) the wave-per-tree
# finisher: counters per kernel, mean over the last 6 dispatches of each
tot = collections.OrderedDict()
for f in sorted(glob.glob(out + "/*counter_collection.csv")):
    rows = [r for r in csv.DictReader(open(f)) if "nuts_kernel" in r["Kernel_Name"] or "nuts_fin_kernel" in r["Kernel_Name"]]
    per = collections.defaultdict(lambda: collections.defaultdict(dict))
    for r in rows:
        kind = "finisher<64 lanes>" if "PrmwcdDistModel<64" in r["Kernel_Name"] else "main<8 lanes>"
        d = per[kind][int(r["Dispatch_Id"])]
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for kind, disp in per.items():
        ids = sorted(disp)[-6:]
        for cn in disp[ids[0]]:
            tot.setdefault(kind, collections.OrderedDict())[cn] = sum(disp[i][cn] for i in ids) / len(ids)
# durations of the two kernels from the kernel trace of the first pass (last 6 dispatches of each)
dur = collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob(out + "/sqa_kernel_trace.csv")[0])):
    if "nuts_kernel" in r["Kernel_Name"] or "nuts_fin_kernel" in r["Kernel_Name"]:
        kind = "finisher<64 lanes>" if "PrmwcdDistModel<64" in r["Kernel_Name"] else "main<8 lanes>"
        dur[kind].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for kind in dur:
    tot.setdefault(kind, collections.OrderedDict())["duration_ms_under_pmc"] = sum(dur[kind][-6:]) / len(dur[kind][-6:])
line = json.loads([l for l in open(out + "/sqa.log") if l.startswith("{")][-1])
with open(out + "/summary.txt", "w") as fh:
    print(f"# nuts_kernel<PrmwcdDistModel>, mean over the 6 timed launches of each kernel; {line['leapfrogs_per_particle_step']:.1f} leapfrogs per "
          f"particle-step, {line['roofline']['launches']} launches of {line['roofline']['avg_launch_ms']:.3f} ms on average, nuts_cap {line['config']['nuts_cap']}", file=fh)
    for kind, d in tot.items():
        print(f"## {kind}", file=fh)
        for k, v in d.items():
            print(f"{k:28s} {v:.4e}", file=fh)
print(open(out + "/summary.txt").read())
PY
