#!/bin/bash
# Round-5 evidence, run on the GPU box (each call under gpurun's 20-minute limit):
#   tools/evidence_r05.sh a   arma: kernel stats + HBM traffic of both bench commands, SQ counters, in-kernel sections
#   tools/evidence_r05.sh b   config 4 (kernel stats, SQ counters of both kernels) and config 5 (stats, traffic, SQ counters,
#                             both step sizes)
#   tools/evidence_r05.sh c   final bench lines (the driver's command with `configs`, 50/10), population-size sweep, the
#                             2-rank rehearsal line, the reference's Monte-Carlo protocol
# Everything lands under gpurun_out/r05_*; the summaries to keep are copied into profiles/ by tools/collect_r05.py.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
case "$1" in
a)
  bash tools/prof_round.sh r05 arma > gpurun_out/r05_prof_round.log 2>&1 || { tail -20 gpurun_out/r05_prof_round.log; exit 1; }
  bash tools/pmc_nuts3.sh r05 20 5 > gpurun_out/r05_pmc_nuts3.log 2>&1 || { tail -20 gpurun_out/r05_pmc_nuts3.log; exit 1; }
  PMC_EXTRA="--particles 131072" bash tools/pmc_nuts3.sh r05_n131072 20 5 > gpurun_out/r05_pmc_nuts3_q.log 2>&1 || { tail -20 gpurun_out/r05_pmc_nuts3_q.log; exit 1; }
  tail -3 gpurun_out/r05_pmc_nuts3_q.log
  ;;
b)
  bash tools/prof_cfg.sh r05_c4 --config c4 --steps 10 --warmup 12 --repeats 3 > gpurun_out/r05_c4.log 2>&1 || { tail -20 gpurun_out/r05_c4.log; exit 1; }
  bash tools/pmc_c4.sh r05 > gpurun_out/r05_c4_pmc.log 2>&1 || { tail -20 gpurun_out/r05_c4_pmc.log; exit 1; }
  bash tools/pmc_c5.sh r05 0.25 > gpurun_out/r05_c5_025.log 2>&1 || { tail -20 gpurun_out/r05_c5_025.log; exit 1; }
  bash tools/pmc_c5.sh r05 0.1 > gpurun_out/r05_c5_01.log 2>&1 || { tail -20 gpurun_out/r05_c5_01.log; exit 1; }
  tail -3 gpurun_out/r05_c5_01.log
  ;;
c)
  mkdir -p gpurun_out/r05_final
  python3 bench.py --steps 20 --warmup 5 > gpurun_out/r05_final/bench_20_5.json 2> gpurun_out/r05_final/bench_20_5.err
  python3 bench.py --steps 50 --warmup 10 --no-extra-configs > gpurun_out/r05_final/bench_50_10.json 2> gpurun_out/r05_final/bench_50_10.err
  python3 bench.py --steps 20 --warmup 5 --no-wide --no-cpu-baseline --no-end-to-end > gpurun_out/r05_final/bench_20_5_nowide.json 2> gpurun_out/r05_final/bench_20_5_nowide.err
  python3 bench.py --config c5 --steps 6 --warmup 2 --step-size 0.25 --repeats 3 > gpurun_out/r05_final/c5_025.json 2> gpurun_out/r05_final/c5_025.err
  python3 bench.py --config c5 --steps 6 --warmup 2 --step-size 0.1 --repeats 3 > gpurun_out/r05_final/c5_01.json 2> gpurun_out/r05_final/c5_01.err
  python3 bench.py --config c4 --steps 10 --warmup 12 --repeats 3 > gpurun_out/r05_final/c4.json 2> gpurun_out/r05_final/c4.err
  python3 tools/n_sweep.py 20 5 > gpurun_out/r05_final/n_sweep.txt 2>&1
  SMCN_BENCH_SAME_DEVICE=1 python3 bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05_final/rehearsal_2ranks_gloo.json 2> gpurun_out/r05_final/rehearsal_2ranks_gloo.err || true
  python3 experiments/run_experiments.py --runs 25 > gpurun_out/r05_final/experiments_arma.txt 2>&1 || true
  tail -8 gpurun_out/r05_final/n_sweep.txt
  ;;
esac
