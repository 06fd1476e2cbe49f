#!/bin/bash
# Compile the micro-benchmarks for gfx950 next to their sources (binaries are git-ignored;
# they travel to the GPU box with the gpurun snapshot).
set -e
cd "$(dirname "$0")"
for f in *.hip; do
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o "${f%.hip}" "$f"
done
