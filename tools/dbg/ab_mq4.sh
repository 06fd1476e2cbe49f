#!/bin/bash
# step alignment x half-way level on config 4 (variants build): tools/dbg/ab_mq4.sh <lib tag>
cd "$GRAFT_REPO_ROOT"
export SMCN_LIB=smcnuts_amd/variants/libsmcnuts_$1.so
for rep in 1 2; do
  for cfg in "16 0" "8 0" "4 0" "16 1" "8 1" "4 1" "2 1"; do
    set -- $cfg
    SMCN_STEP_ALIGN=$1 SMCN_NUTS_REQUEUE_HALF=$2 timeout -k 10 120 python3 bench.py --config c4 --steps 10 --warmup 12 --no-peaks 2>gpurun_out/ab_mq4.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('align $1 half $2:', round(d['value']/1e9, 4), 'G lf/s', round(d['ms_per_step'], 3), 'ms per step', 'launch avg', round(d['roofline']['avg_launch_ms'],3), 'ess', d['final_ess'])" || { echo "$cfg FAILED"; tail -3 gpurun_out/ab_mq4.err; exit 1; }
  done
done
