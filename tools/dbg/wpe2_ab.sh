#!/bin/bash
# A/B of the lane-queue kernel at two wavefronts per SIMD (variants build, SMCN_LANE_WPE=2): tools/dbg/wpe2_ab.sh <lib tag> [N list]
set -e
cd "$GRAFT_REPO_ROOT"
LIB=smcnuts_amd/variants/libsmcnuts_$1.so
NS=${2:-131072,262144}
mkdir -p gpurun_out/wpe2
for mode in 1 2 1 2; do
  for n in ${NS//,/ }; do
    SMCN_LIB=$LIB SMCN_LANE_WPE=$mode timeout -k 10 120 python3 bench.py --steps 20 --warmup 5 --particles $n --no-cpu-baseline --no-end-to-end --no-extra-configs --repeats 3 --settle-ms 150 > gpurun_out/wpe2/$1_${mode}_$n.json 2> gpurun_out/wpe2/$1_${mode}_$n.err || { tail -5 gpurun_out/wpe2/$1_${mode}_$n.err; exit 1; }
    python3 - gpurun_out/wpe2/$1_${mode}_$n.json $mode $n <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(f"wpe {sys.argv[2]} N {sys.argv[3]}: {d['value']/1e9:.3f} G  launch {d['roofline']['avg_launch_ms']:.3f} ms  lf/particle-step {d['leapfrogs_per_particle_step']:.4f}  ess {d.get('final_ess')}")
PY
  done
done
