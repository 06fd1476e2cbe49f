#!/bin/bash
# GPU box: parity tests, then the driver's bench command (and the 50-step one): tools/gpu_check.sh <tag> [pytest args]
T=${1:-check}; shift
mkdir -p gpurun_out/$T
timeout -k 10 900 python -m pytest tests -m gpu -q "$@" > gpurun_out/$T/tests.log 2>&1; echo tests_rc=$? >> gpurun_out/$T/tests.log; tail -15 gpurun_out/$T/tests.log
cp gpurun_out/tolerances_observed.json gpurun_out/$T/ 2>/dev/null
for cfg in "20 5" "50 10"; do set -- $cfg
  timeout -k 10 300 python bench.py --steps $1 --warmup $2 --no-cpu-baseline > gpurun_out/$T/bench_$1_$2.json 2> gpurun_out/$T/bench_$1_$2.err
  python - gpurun_out/$T/bench_$1_$2.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(f"steps {d['steps']}: {d['value']/1e9:.3f} G lf/s  (min {d['repeats']['value_min']/1e9:.3f} max {d['repeats']['value_max']/1e9:.3f})  "
      f"kernel {d['roofline']['avg_launch_ms']:.3f} ms  valu_f64_frac {d['roofline']['valu_f64_frac']:.3f}")
e = d.get("end_to_end")
if e: print(f"  end_to_end: construct {e['construct_s']*1e3:.1f} ms run_time {e['run_time_s']*1e3:.1f} ms  {e['value_over_run_time']/1e9:.2f} G")
PY
done
