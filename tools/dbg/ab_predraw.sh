#!/bin/bash
# momenta of the next transition drawn on a side stream behind the NUTS launch (config 5): tools/dbg/ab_predraw.sh <lib tag>
cd "$GRAFT_REPO_ROOT"
export SMCN_LIB=smcnuts_amd/variants/libsmcnuts_$1.so
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "config5 or wide_particles or gauss or momenta or wave_kernel" 2>&1 | tail -3
for rep in 1 2; do
  for off in 1 0; do
    for eps in 0.25 0.1; do
      SMCN_NO_PREDRAW=$off timeout -k 10 120 python3 bench.py --config c5 --steps 6 --warmup 2 --step-size $eps --no-peaks 2>gpurun_out/ab_pd.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('predraw off=$off eps $eps:', round(d['value']/1e9, 4), 'G lf/s', round(d['ms_per_step'], 4), 'ms per step', 'launch', round(d['roofline']['avg_launch_ms'],3), 'ess', d['final_ess'])" || { echo FAILED; tail -3 gpurun_out/ab_pd.err; exit 1; }
    done
  done
done
