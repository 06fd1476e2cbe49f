#!/bin/bash
# A/B of the inner park level of config 4's first NUTS launch (SMCN_NUTS_REQUEUE = doublings, 0 = off): tools/dbg/ab_mq.sh <lib tag> [levels]
cd "$GRAFT_REPO_ROOT"
export SMCN_LIB=smcnuts_amd/variants/libsmcnuts_$1.so
for rep in 1 2; do
  for b in ${2:-0 8 7 6}; do
    SMCN_NUTS_REQUEUE=$b timeout -k 10 120 python3 bench.py --config c4 --steps 10 --warmup 12 --no-peaks 2>gpurun_out/ab_mq_$b.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('requeue $b:', round(d['value']/1e9, 4), 'G lf/s', round(d['ms_per_step'], 3), 'ms per step', 'launch avg', round(d['roofline']['avg_launch_ms'],3), 'lf', d['leapfrogs_per_particle_step'], 'ess', d['final_ess'])" || { echo "requeue $b FAILED"; tail -3 gpurun_out/ab_mq_$b.err; exit 1; }
  done
done
