#!/bin/bash
# Round-3 evidence, run on the GPU box in three calls (each under gpurun's 20-minute limit):
#   tools/evidence_r03.sh a   arma: kernel stats + HBM traffic of both bench commands, SQ counters, in-kernel sections
#   tools/evidence_r03.sh b   config 4 (stats, SQ counters) and config 5 (stats, traffic, SQ counters at both step sizes)
#   tools/evidence_r03.sh c   final bench lines, population-size sweep, the reference's Monte-Carlo protocol
# Everything lands under gpurun_out/r03_*; the summaries to keep are copied into profiles/ by tools/collect_r03.py.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
case "$1" in
a)
  bash tools/prof_round.sh r03 arma > gpurun_out/r03_prof_round.log 2>&1 || { tail -20 gpurun_out/r03_prof_round.log; exit 1; }
  bash tools/pmc_nuts3.sh r03 20 5 > gpurun_out/r03_pmc_nuts3.log 2>&1 || { tail -20 gpurun_out/r03_pmc_nuts3.log; exit 1; }
  SMCN_LIB=smcnuts_amd/variants/libsmcnuts_prof.so python3 tools/prof_sections.py 20 5 > gpurun_out/r03_sections.txt 2>&1
  tail -3 gpurun_out/r03_sections.txt
  ;;
b)
  bash tools/prof_cfg.sh r03_c4 --config c4 --steps 10 --warmup 12 > gpurun_out/r03_c4.log 2>&1 || { tail -20 gpurun_out/r03_c4.log; exit 1; }
  bash tools/pmc_c4.sh r03 > gpurun_out/r03_c4_pmc.log 2>&1 || { tail -20 gpurun_out/r03_c4_pmc.log; exit 1; }
  bash tools/pmc_c5.sh r03 0.25 > gpurun_out/r03_c5_025.log 2>&1 || { tail -20 gpurun_out/r03_c5_025.log; exit 1; }
  bash tools/pmc_c5.sh r03 0.1 > gpurun_out/r03_c5_01.log 2>&1 || { tail -20 gpurun_out/r03_c5_01.log; exit 1; }
  tail -3 gpurun_out/r03_c5_01.log
  ;;
c)
  mkdir -p gpurun_out/r03_final
  python3 bench.py --steps 20 --warmup 5 > gpurun_out/r03_final/bench_20_5.json 2> gpurun_out/r03_final/bench_20_5.err
  python3 bench.py --steps 50 --warmup 10 > gpurun_out/r03_final/bench_50_10.json 2> gpurun_out/r03_final/bench_50_10.err
  python3 bench.py --steps 20 --warmup 5 --no-wide --no-cpu-baseline --no-end-to-end > gpurun_out/r03_final/bench_20_5_nowide.json 2> gpurun_out/r03_final/bench_20_5_nowide.err
  python3 bench.py --config c5 --steps 6 --warmup 2 --step-size 0.25 --repeats 3 > gpurun_out/r03_final/c5_025.json 2> gpurun_out/r03_final/c5_025.err
  python3 bench.py --config c5 --steps 6 --warmup 2 --step-size 0.1 --repeats 3 > gpurun_out/r03_final/c5_01.json 2> gpurun_out/r03_final/c5_01.err
  python3 tools/n_sweep.py 20 5 > gpurun_out/r03_final/n_sweep.txt 2>&1
  python3 experiments/run_experiments.py --runs 25 > gpurun_out/r03_final/experiments_arma.txt 2>&1 || true
  tail -5 gpurun_out/r03_final/n_sweep.txt
  ;;
esac
