#!/bin/bash
# Round-4 evidence, run on the GPU box in three calls (each under gpurun's 20-minute limit):
#   tools/evidence_r04.sh a   arma: kernel stats + HBM traffic of both bench commands, SQ counters (N = 65 536, and the
#                             queue kernel at N = 131 072), in-kernel sections
#   tools/evidence_r04.sh b   config 4 (stats, SQ counters, lane-count / occupancy A/B) and config 5 (stats, traffic, SQ
#                             counters at both step sizes)
#   tools/evidence_r04.sh c   final bench lines, population-size sweep, the 2-rank rehearsal line, the reference's
#                             Monte-Carlo protocol, the fp64 MFMA micro-benchmark
# Everything lands under gpurun_out/r04_*; the summaries to keep are copied into profiles/ by tools/collect_r04.py.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
case "$1" in
a)
  bash tools/prof_round.sh r04 arma > gpurun_out/r04_prof_round.log 2>&1 || { tail -20 gpurun_out/r04_prof_round.log; exit 1; }
  bash tools/pmc_nuts3.sh r04 20 5 > gpurun_out/r04_pmc_nuts3.log 2>&1 || { tail -20 gpurun_out/r04_pmc_nuts3.log; exit 1; }
  PMC_EXTRA="--particles 131072" bash tools/pmc_nuts3.sh r04_n131072 20 5 > gpurun_out/r04_pmc_nuts3_q.log 2>&1 || { tail -20 gpurun_out/r04_pmc_nuts3_q.log; exit 1; }
  SMCN_LIB=smcnuts_amd/variants/libsmcnuts_prof.so python3 tools/prof_sections.py 20 5 > gpurun_out/r04_sections.txt 2>&1
  tail -3 gpurun_out/r04_sections.txt
  ;;
b)
  bash tools/prof_cfg.sh r04_c4 --config c4 --steps 10 --warmup 12 > gpurun_out/r04_c4.log 2>&1 || { tail -20 gpurun_out/r04_c4.log; exit 1; }
  bash tools/pmc_c4.sh r04 > gpurun_out/r04_c4_pmc.log 2>&1 || { tail -20 gpurun_out/r04_c4_pmc.log; exit 1; }
  { echo "# config 4 (PRMwCD, N = 65536, Gaussian L-kernel + tempering): what bounds the launch -- A/B of lanes per particle and of"
    echo "# wavefronts per SIMD on the same box (tools/ab_c4.sh; G = 16 / 32: -DSMCN_VARIANTS build, RED = 2, 4 LDS levels)"
    echo "## product: 8 lanes per particle, two wavefronts per SIMD"; bash tools/ab_c4.sh base
    echo "## one block (4 wavefronts) per CU = one wavefront per SIMD, same kernel (SMCN_NUTS_BLOCKS_PER_CU=1)"; SMCN_NUTS_BLOCKS_PER_CU=1 bash tools/ab_c4.sh base
    echo "## 16 lanes per particle"; SMCN_PRMWCD_DIST=162 bash tools/ab_c4.sh c4g
    echo "## 32 lanes per particle"; SMCN_PRMWCD_DIST=322 bash tools/ab_c4.sh c4g
  } > gpurun_out/r04_c4_ab.txt 2>&1
  bash tools/pmc_c5.sh r04 0.25 > gpurun_out/r04_c5_025.log 2>&1 || { tail -20 gpurun_out/r04_c5_025.log; exit 1; }
  bash tools/pmc_c5.sh r04 0.1 > gpurun_out/r04_c5_01.log 2>&1 || { tail -20 gpurun_out/r04_c5_01.log; exit 1; }
  tail -3 gpurun_out/r04_c5_01.log
  ;;
c)
  mkdir -p gpurun_out/r04_final
  python3 bench.py --steps 20 --warmup 5 > gpurun_out/r04_final/bench_20_5.json 2> gpurun_out/r04_final/bench_20_5.err
  python3 bench.py --steps 50 --warmup 10 > gpurun_out/r04_final/bench_50_10.json 2> gpurun_out/r04_final/bench_50_10.err
  python3 bench.py --steps 20 --warmup 5 --no-wide --no-cpu-baseline --no-end-to-end > gpurun_out/r04_final/bench_20_5_nowide.json 2> gpurun_out/r04_final/bench_20_5_nowide.err
  python3 bench.py --config c5 --steps 6 --warmup 2 --step-size 0.25 --repeats 3 > gpurun_out/r04_final/c5_025.json 2> gpurun_out/r04_final/c5_025.err
  python3 bench.py --config c5 --steps 6 --warmup 2 --step-size 0.1 --repeats 3 > gpurun_out/r04_final/c5_01.json 2> gpurun_out/r04_final/c5_01.err
  python3 bench.py --config c4 --steps 10 --warmup 12 > gpurun_out/r04_final/c4.json 2> gpurun_out/r04_final/c4.err
  python3 tools/n_sweep.py 20 5 > gpurun_out/r04_final/n_sweep.txt 2>&1
  SMCN_BENCH_SAME_DEVICE=1 python3 bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04_final/rehearsal_2ranks_gloo.json 2> gpurun_out/r04_final/rehearsal_2ranks_gloo.err || true
  python3 experiments/run_experiments.py --runs 25 > gpurun_out/r04_final/experiments_arma.txt 2>&1 || true
  tools/ubench/mfma_f64 > gpurun_out/r04_final/ubench_mfma_f64.txt 2>&1 || true
  tail -8 gpurun_out/r04_final/n_sweep.txt
  ;;
esac
