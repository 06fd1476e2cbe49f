#!/bin/bash
# GPU box: the builder-run configurations (c4, c5 at both step sizes): tools/gpu_cfg.sh <tag>
T=${1:-cfg}
mkdir -p gpurun_out/$T
run() { name=$1; shift
  timeout -k 10 400 python bench.py "$@" > gpurun_out/$T/$name.json 2> gpurun_out/$T/$name.err || { echo "$name FAILED"; tail -5 gpurun_out/$T/$name.err; return 1; }
  python - gpurun_out/$T/$name.json $name <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(f"{sys.argv[2]}: {d['value']/1e9:.3f} G lf/s  ms/step {d['ms_per_step']:.3f}  kernel {d['roofline']['avg_launch_ms']:.3f} ms  valu_f64_frac {d['roofline']['valu_f64_frac']:.3f}  lf/particle/step {d['leapfrogs_per_particle_step']:.1f}")
PY
}
run c4 --config c4 --steps 10 --warmup 12 && run c5_025 --config c5 --steps 6 --warmup 2 --step-size 0.25 --repeats 3 && run c5_01 --config c5 --steps 6 --warmup 2 --step-size 0.1 --repeats 3
