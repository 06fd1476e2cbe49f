#!/bin/bash
# kernel-stats profile + JSON line of a secondary configuration: tools/prof_cfg.sh <tag> <bench args...>
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
python3 bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- python3 bench.py "$@" > $OUT/under_rocprof.json 2> $OUT/stats.err || { tail -5 $OUT/stats.err; exit 1; }
head -8 $OUT/stats_kernel_stats.csv | cut -c1-170
python3 -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print(round(d['value']/1e9,4),'G lf/s; kernel ms', round(d['roofline']['avg_launch_ms'],3), 'frac', round(d['roofline']['frac'],3), 'valu', round(d['roofline']['valu_f64_frac'],3), 'share', round(d['nuts_kernel_share_of_step'],3))"
