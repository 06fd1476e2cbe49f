#!/bin/bash
# A/B of variant libraries on the driver's command: tools/dbg/ab_libs.sh <tag> <tag> ...  (each twice, interleaved)
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ab_libs
for rep in 1 2; do
  for t in "$@"; do
    SMCN_LIB=smcnuts_amd/variants/libsmcnuts_$t.so timeout -k 10 120 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --no-extra-configs --no-peaks --repeats 3 --settle-ms 150 ${AB_EXTRA:-} > gpurun_out/ab_libs/${t}_$rep.json 2> gpurun_out/ab_libs/${t}_$rep.err || { tail -5 gpurun_out/ab_libs/${t}_$rep.err; exit 1; }
    python3 - gpurun_out/ab_libs/${t}_$rep.json $t <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(f"{sys.argv[2]:>16}: {d['value']/1e9:.3f} G  launch {d['roofline']['avg_launch_ms']:.4f} ms  step {d['ms_per_step']:.4f} ms  lf/particle-step {d['leapfrogs_per_particle_step']:.4f}")
PY
  done
done
