#!/bin/bash
# GPU box: parity tests, then the driver's bench command (and the 50-step one): tools/gpu_check.sh <tag>
T=${1:-check}
mkdir -p gpurun_out/$T
python -m pytest tests -m gpu -x -q > gpurun_out/$T/tests.log 2>&1; echo tests_rc=$? >> gpurun_out/$T/tests.log; tail -2 gpurun_out/$T/tests.log
for cfg in "20 5" "50 10"; do set -- $cfg
  python bench.py --steps $1 --warmup $2 --no-cpu-baseline > gpurun_out/$T/bench_$1_$2.json 2> gpurun_out/$T/bench_$1_$2.err
  python - gpurun_out/$T/bench_$1_$2.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(f"steps {d['steps']}: {d['value']/1e9:.3f} G lf/s  (min {d['repeats']['value_min']/1e9:.3f} max {d['repeats']['value_max']/1e9:.3f})  "
      f"kernel {d['roofline']['avg_launch_ms']:.3f} ms  valu_f64_frac {d['roofline']['valu_f64_frac']:.3f}")
PY
done
