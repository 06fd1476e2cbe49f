# SQ counters of the wave kernel for two library variants (A/B): tools/dbg/pmc_wave_ab.sh <eps> <tag> [<tag> ...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
EPS=$1; shift
for v in "$@"; do
  if [ "$v" = base ]; then unset SMCN_LIB; else export SMCN_LIB=smcnuts_amd/variants/libsmcnuts_$v.so; fi
  OUT=gpurun_out/pmcab_$v; mkdir -p $OUT
  ARGS="--config c5 --steps 3 --warmup 1 --step-size $EPS --repeats 1 --no-peaks"
  run() { n=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT -o $n -- python3 bench.py $ARGS > $OUT/$n.log 2>&1 || { tail -5 $OUT/$n.log; exit 1; }; }
  run sqa SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES
  run sqb SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_FMA_F64
  run sqc SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU
  run mem FETCH_SIZE
  run mem2 WRITE_SIZE
  python3 - $OUT $v <<'PY'
import csv, glob, sys, collections
out, tag = sys.argv[1], sys.argv[2]
tot = collections.OrderedDict()
for f in sorted(glob.glob(out + "/*counter_collection.csv")):
    rows = [r for r in csv.DictReader(open(f)) if "nuts_wave_kernel" in r["Kernel_Name"] or "nuts_kernel" in r["Kernel_Name"]]
    per = collections.defaultdict(dict)
    for r in rows:
        per[int(r["Dispatch_Id"])][r["Counter_Name"]] = per[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    ids = sorted(per)
    last = ids[-1]
    for cn in per[last]:
        tot[cn] = per[last][cn]
print("##", tag, " ".join(f"{k}={v:.4g}" for k, v in tot.items()))
PY
done
