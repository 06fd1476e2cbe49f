"""The Monte-Carlo harness: same protocol and on-disk format as the reference's
experiments/run_experiments.py:102-215 -- per run and strategy five comma-delimited CSVs
(np.savetxt; mean_estimate/var_estimate [K+1, Dc], ess/phi/acceptance_rate [K+1])."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_harness_writes_the_reference_files(tmp_path):
    K, N, runs = 15, 100, 2
    p = subprocess.run([sys.executable, os.path.join(ROOT, "experiments", "run_experiments.py"), "--model", "arma", "--runs",
                        str(runs), "--N", str(N), "--K", str(K), "--out", str(tmp_path)], capture_output=True, text=True,
                       timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    truth = np.loadtxt(os.path.join(ROOT, "smcnuts_amd", "model", "data", "arma.params"), usecols=(1,))
    for strategy in ("forward_lkernel", "gaussian_lkernel", "asymptotic_lkernel"):      # run_experiments.py:136-186
        d = tmp_path / "arma" / strategy
        assert sorted(os.listdir(d)) == sorted(f"{n}_{i}.csv" for i in range(runs) for n in
                                                ("mean_estimate", "var_estimate", "ess", "phi", "acceptance_rate"))
        for i in range(runs):
            first = open(d / f"mean_estimate_{i}.csv").readline()
            assert first.count(",") == 3 and " " not in first.strip()                 # delimiter="," (:207-215)
            mean = np.loadtxt(d / f"mean_estimate_{i}.csv", delimiter=",")
            var = np.loadtxt(d / f"var_estimate_{i}.csv", delimiter=",")
            assert mean.shape == (K + 1, 4) and var.shape == (K + 1, 4)
            for name in ("ess", "phi", "acceptance_rate"):
                a = np.loadtxt(d / f"{name}_{i}.csv", delimiter=",")
                assert a.shape == (K + 1,) and np.all(np.isfinite(a))
            ess, phi = np.loadtxt(d / f"ess_{i}.csv", delimiter=","), np.loadtxt(d / f"phi_{i}.csv", delimiter=",")
            assert np.all((ess > 0) & (ess <= N + 1e-9)) and np.all(np.diff(phi) >= 0) and phi[-1] == 1.0
            # plot_experiments.py:61-79 looks at the squared error against <model>.params; N = 100 particles
            # after 15 iterations sit within a few posterior standard deviations of it
            assert np.all((mean[-1] - truth) ** 2 < np.array([0.02, 0.02, 0.1, 0.01])), (strategy, i, mean[-1])
    assert "MSE of the final mean estimate" in p.stdout
