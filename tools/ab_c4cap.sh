#!/bin/bash
# config 4: doublings of the first launch (tools/ab_c4cap.sh 8 9 10)
for cap in "$@"; do
  python bench.py --config c4 --steps 10 --warmup 12 --repeats 3 --no-peaks --nuts-cap $cap 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cap $cap:', round(d['value']/1e9, 3), 'G lf/s', round(d['roofline']['avg_launch_ms'], 2), 'ms per launch', round(d['ms_per_step'], 2), 'ms per step')"
done
