#!/bin/bash
# A/B of library variants on config 4 (GPU box): tools/ab_c4.sh <tag> [<tag> ..]   (tag "base" = the product library)
for t in "$@"; do
  if [ "$t" = base ]; then unset SMCN_LIB; else export SMCN_LIB=smcnuts_amd/variants/libsmcnuts_$t.so; fi
  python bench.py --config c4 --steps 10 --warmup 12 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$t:', round(d['value']/1e9, 3), 'G lf/s', round(d['roofline']['avg_launch_ms'], 2), 'ms per launch', round(d['ms_per_step'], 2), 'ms per step', d['leapfrogs_per_particle_step'])"
done
