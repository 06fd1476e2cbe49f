#!/bin/bash
# the half-way park level on config 4: tools/dbg/ab_mq3.sh <lib tag>
cd "$GRAFT_REPO_ROOT"
export SMCN_LIB=smcnuts_amd/variants/libsmcnuts_$1.so
SMCN_NUTS_REQUEUE_HALF=1 SMCN_MQ_DEBUG=1 timeout -k 10 100 python3 bench.py --config c4 --steps 2 --warmup 1 --no-peaks --repeats 1 > gpurun_out/mq_dbg.json 2> gpurun_out/mq_dbg.err; echo rc=$?; tail -3 gpurun_out/mq_dbg.err
for rep in 1 2; do
  for h in 0 1; do
    SMCN_NUTS_REQUEUE_HALF=$h timeout -k 10 120 python3 bench.py --config c4 --steps 10 --warmup 12 --no-peaks 2>gpurun_out/ab_mq3.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('half $h:', round(d['value']/1e9, 4), 'G lf/s', round(d['ms_per_step'], 3), 'ms per step', 'launch avg', round(d['roofline']['avg_launch_ms'],3), 'lf', d['leapfrogs_per_particle_step'], 'ess', d['final_ess'])" || { echo "half $h FAILED"; tail -3 gpurun_out/ab_mq3.err; exit 1; }
  done
done
