#!/bin/bash
# config 4: one launch against two-phase launches (smcn_set_nuts_cap) on the same box: tools/c4_two_phase.sh
for args in "--nuts-cap 0" "--nuts-cap 9" "--nuts-cap 10" "--nuts-cap 9 --no-widen" "--nuts-cap 8"; do
  python bench.py --config c4 --steps 10 --warmup 12 $args 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$args:', round(d['value']/1e9, 3), 'G lf/s', round(d['ms_per_step'], 2), 'ms per step; NUTS launches', d['roofline']['launches'], 'x', round(d['roofline']['avg_launch_ms'], 2), 'ms')"
done
