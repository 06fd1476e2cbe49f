set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_pair2
mkdir -p $OUT
run() { n=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT -o $n -- python3 bench.py --steps 20 --warmup 10 --no-cpu-baseline > $OUT/$n.log 2>&1 || { tail -5 $OUT/$n.log; exit 1; }; }
run a SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_REQ SQ_IFETCH SQ_INSTS_BRANCH SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU
run b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_FLOPS_FP64
