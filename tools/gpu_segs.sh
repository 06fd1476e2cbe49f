#!/bin/bash
# GPU box: lane-queue tests, then the population sweep with whole blocks / 2 / 4 segments: tools/gpu_segs.sh <tag>
T=${1:-segs}
mkdir -p gpurun_out/$T
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k "lane_queue or full_loop or device_resident or fused" > gpurun_out/$T/tests.log 2>&1; echo rc=$? >> gpurun_out/$T/tests.log; tail -4 gpurun_out/$T/tests.log
grep -q "rc=0" gpurun_out/$T/tests.log || exit 1
for s in ${SEGS:-1 2 4}; do
  SWEEP_SEGS=$s SWEEP_N=${SWEEP_N:-65600,73728,98304,131072,196608,262144} timeout -k 10 600 python tools/n_sweep.py 20 5 > gpurun_out/$T/n_sweep_s$s.txt 2>&1 || exit 1
  echo "segments $s"; tail -7 gpurun_out/$T/n_sweep_s$s.txt
done
