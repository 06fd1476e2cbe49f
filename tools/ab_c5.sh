#!/bin/bash
# config 5 A/B on one box: tools/ab_c5.sh <label> [env assignments...]  ->  one line per step size
cd "$GRAFT_REPO_ROOT"
L=$1; shift
for e in 0.25 0.1; do
  env "$@" python3 bench.py --config c5 --steps 6 --warmup 2 --step-size $e --repeats 3 --no-peaks 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L eps $e:', round(d['value']/1e9,4),'G lf/s; ms/step', round(d['ms_per_step'],3),'launch ms', round(d['roofline']['avg_launch_ms'],3))"
done
