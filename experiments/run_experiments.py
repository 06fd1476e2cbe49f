#!/usr/bin/env python3
"""Monte-Carlo harness on the MI355X path.

Same protocol and on-disk format as the reference's experiments/run_experiments.py
(:38-47,102-215): R independent runs x the three L-kernel strategies, each writing
mean_estimate_<i>.csv, var_estimate_<i>.csv, ess_<i>.csv, phi_<i>.csv,
acceptance_rate_<i>.csv (np.savetxt, comma-delimited, [K+1, Dc] / [K+1]) under
<out>/<model>/<strategy>/ -- so the reference's plot_experiments.py reads them as is.
Run i is seeded 10*(i+1) like the reference's RandomState(10*(i+1)); the draws themselves
come from Philox (DESIGN.md "RNG"), so individual runs differ from the reference's while
the Monte-Carlo averages agree.  Also prints the MSE of the final mean estimate against
the ground truth in <model>.params (plot_experiments.py:61-79).

    python experiments/run_experiments.py --model arma --runs 25 --N 100 --K 15
"""
import argparse
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

STRATEGIES = (("forward_lkernel", "forwardsLKernel", False),
              ("gaussian_lkernel", "GaussianApproxLKernel", False),
              ("asymptotic_lkernel", "asymptoticLKernel", True))


def ground_truth(model):
    path = ROOT / "smcnuts_amd" / "model" / "data" / f"{model}.params"
    rows = [line.split() for line in open(path) if line.strip()]
    return [r[0] for r in rows], np.array([float(r[1]) for r in rows])


def save_output(smc, out_dir, i):
    out_dir.mkdir(parents=True, exist_ok=True)
    for name, arr in (("mean_estimate", smc.mean_estimate), ("var_estimate", smc.variance_estimate),
                      ("ess", smc.ess), ("phi", smc.phi), ("acceptance_rate", smc.acceptance_rate)):
        np.savetxt(out_dir / f"{name}_{i}.csv", arr, delimiter=",")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="arma", choices=["arma", "PRMwCD"])
    ap.add_argument("--runs", type=int, default=25)
    ap.add_argument("--N", type=int, default=100)
    ap.add_argument("--K", type=int, default=15)
    ap.add_argument("--step-size", type=float, default=0.01)      # stan_models/<m>/model_config.json
    ap.add_argument("--out", default=str(ROOT / "output"))
    ap.add_argument("--strategies", default="forward_lkernel,gaussian_lkernel,asymptotic_lkernel")
    args = ap.parse_args()

    from smcnuts_amd import SMCSampler, StanModel
    target = StanModel(args.model)
    names, truth = ground_truth(args.model)
    wanted = set(args.strategies.split(","))
    finals = {s[0]: [] for s in STRATEGIES if s[0] in wanted}
    for i in range(args.runs):
        for tag, lkernel, tempering in STRATEGIES:
            if tag not in wanted:
                continue
            smc = SMCSampler(K=args.K, N=args.N, target=target, step_size=args.step_size, lkernel=lkernel,
                             tempering=tempering, seed=10 * (i + 1))
            smc.sample(show_progress=False)
            save_output(smc, Path(args.out) / args.model / tag, i)
            finals[tag].append(smc.mean_estimate[-1])
            print(f"run {i + 1}/{args.runs} {tag:20s} {smc.run_time * 1e3:8.1f} ms  ess[K]={smc.ess[-1]:.1f}", flush=True)
    print(f"\nMSE of the final mean estimate against {args.model}.params ({', '.join(names)})")
    for tag, est in finals.items():
        est = np.array(est)
        print(f"  {tag:20s} mean={np.round(est.mean(axis=0), 5).tolist()}  mse={np.mean((est - truth) ** 2, axis=0).round(8).tolist()}")


if __name__ == "__main__":
    main()
