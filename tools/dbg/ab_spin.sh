#!/bin/bash
# A/B of the polled stream wait (SMCN_SPIN_US) on config 4, config 5 and the headline: tools/dbg/ab_spin.sh <lib tag>
cd "$GRAFT_REPO_ROOT"
export SMCN_LIB=smcnuts_amd/variants/libsmcnuts_$1.so
show() { python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
e = d.get('end_to_end') or {}
print('$1:', round(d['value']/1e9, 4), 'G lf/s', round(d['ms_per_step'], 4), 'ms per step', 'cold run_time', e.get('run_time_s'), 'first', (e.get('first_in_fresh_process') or {}).get('run_time_s'))"; }
for rep in 1 2; do
  for us in 0 120; do
    export SMCN_SPIN_US=$us
    python3 bench.py --config c4 --steps 10 --warmup 12 --no-peaks 2>/dev/null | show "c4 spin=$us"
    python3 bench.py --config c5 --steps 6 --warmup 2 --step-size 0.25 --no-peaks 2>/dev/null | show "c5 0.25 spin=$us"
    python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --no-peaks 2>/dev/null | show "arma spin=$us"
  done
done
