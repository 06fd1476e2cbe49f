#!/bin/bash
# cap x inner level of config 4's two-phase launch: tools/dbg/ab_mq2.sh <lib tag> "cap:requeue ..."
cd "$GRAFT_REPO_ROOT"
export SMCN_LIB=smcnuts_amd/variants/libsmcnuts_$1.so
for rep in 1 2; do
  for cr in ${2:-9:8 10:9 10:8 8:7 9:0}; do
    cap=${cr%%:*}; b=${cr##*:}
    SMCN_NUTS_REQUEUE=$b timeout -k 10 120 python3 bench.py --config c4 --steps 10 --warmup 12 --no-peaks --nuts-cap $cap 2>gpurun_out/ab_mq2.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cap $cap requeue $b:', round(d['value']/1e9, 4), 'G lf/s', round(d['ms_per_step'], 3), 'ms per step', 'launch avg', round(d['roofline']['avg_launch_ms'],3), 'ess', d['final_ess'])" || { echo "cap $cap requeue $b FAILED"; tail -3 gpurun_out/ab_mq2.err; exit 1; }
  done
done
