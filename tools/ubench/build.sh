#!/bin/bash
# Compile the micro-benchmarks for gfx950 next to their sources (binaries are git-ignored;
# they travel to the GPU box with the gpurun snapshot).
set -e
cd "$(dirname "$0")"
for f in *.hip; do
    [ "$f" = lane_eval.hip ] && { /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -DSMCN_VARIANTS -o lane_eval lane_eval.hip; continue; }
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o "${f%.hip}" "$f"
done
