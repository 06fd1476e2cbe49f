#!/bin/bash
# SQ counters of the lane-queue kernel at one / two wavefronts per SIMD: tools/dbg/wpe2_pmc.sh <lib tag> <N>
set -e
cd "$GRAFT_REPO_ROOT"
export SMCN_LIB=smcnuts_amd/variants/libsmcnuts_$1.so
export PMC_EXTRA="--particles $2 --no-extra-configs --no-peaks"
SMCN_LANE_WPE=1 tools/pmc_nuts3.sh $1_w1_$2 > /dev/null
SMCN_LANE_WPE=2 tools/pmc_nuts3.sh $1_w2_$2 > /dev/null
paste gpurun_out/pmc_$1_w1_$2/summary.txt gpurun_out/pmc_$1_w2_$2/summary.txt | awk '{print $1, $2, $4}'
