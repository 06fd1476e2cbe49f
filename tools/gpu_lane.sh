#!/bin/bash
# GPU box: lane-queue tests, bench at the driver's command, population sweep, c4 lane-count A/B: tools/gpu_lane.sh <tag>
T=${1:-lane}
mkdir -p gpurun_out/$T
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k "lane_queue or full_loop or device_resident or fused or philox or ragged or series_lengths" > gpurun_out/$T/tests.log 2>&1; echo rc=$? >> gpurun_out/$T/tests.log; tail -6 gpurun_out/$T/tests.log
grep -q "rc=0" gpurun_out/$T/tests.log || exit 1
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/$T/bench_20_5.json 2> gpurun_out/$T/bench_20_5.err || { tail -5 gpurun_out/$T/bench_20_5.err; exit 1; }
python - gpurun_out/$T/bench_20_5.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(f"steps {d['steps']}: {d['value']/1e9:.3f} G lf/s  (min {d['repeats']['value_min']/1e9:.3f} max {d['repeats']['value_max']/1e9:.3f})  kernel {d['roofline']['avg_launch_ms']:.3f} ms")
PY
timeout -k 10 600 python tools/n_sweep.py 20 5 > gpurun_out/$T/n_sweep.txt 2>&1; tail -8 gpurun_out/$T/n_sweep.txt
