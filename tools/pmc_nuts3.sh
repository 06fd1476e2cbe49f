#!/bin/bash
# SQ counters of the NUTS kernel on the driver's command (each --pmc set in its own run, kernel trace only):
#   tools/pmc_nuts3.sh <tag> [steps] [warmup]   ->  gpurun_out/pmc_<tag>/*.csv + summary.txt
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$1
S=${2:-20}; W=${3:-5}
mkdir -p $OUT
run() { n=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT -o $n -- python3 bench.py --steps $S --warmup $W --no-cpu-baseline --no-end-to-end ${PMC_EXTRA:-} > $OUT/$n.log 2>&1 || { tail -5 $OUT/$n.log; exit 1; }; }
run a SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES
run b SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES
run c SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES
run d SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_CYCLES
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.OrderedDict()
for f in sorted(glob.glob(out + "/*counter_collection.csv")):
    rows = list(csv.DictReader(open(f)))
    # the timed launch = the nuts kernel dispatch with the largest counter sum
    nuts = [r for r in rows if "nuts" in r["Kernel_Name"] and "prep" not in r["Kernel_Name"] and "post" not in r["Kernel_Name"]]
    last = max(int(r["Dispatch_Id"]) for r in nuts)
    for r in nuts:
        if int(r["Dispatch_Id"]) == last:
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            name = r["Kernel_Name"]
with open(out + "/summary.txt", "w") as fh:
    print("# last NUTS dispatch:", name[:90], file=fh)
    for k, v in tot.items():
        print(f"{k:28s} {v:.4e}", file=fh)
print(open(out + "/summary.txt").read())
PY
