"""The RCCL plumbing on one GPU: torch.distributed "nccl" with world_size 1.
Exercises the device path of the shard exchange (all-gather on raw device
pointers aliased as torch tensors, stream-ordered with the library's kernels)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_nccl_world1_device_exchange_matches_plain_run():
    """Runs in a fresh interpreter: torch has to initialise the GPU before the HIP library does
    (as under torchrun), which an earlier test of this session may already have prevented."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    p = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0 and "DIST-OK" in p.stdout, p.stdout[-3000:] + p.stderr[-3000:]


def _body():
    import torch
    import torch.distributed as dist
    from smcnuts_amd import ArmaModel, SMCSampler
    from smcnuts_amd.parallel import TorchDistComm
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        comm = TorchDistComm(torch.device("cuda", 0))
        assert comm.device_path, "aliased all-gather self test failed"
        comm.force_exchange = True          # run the exchange although there is only one shard
        a = SMCSampler(K=6, N=8192, target=ArmaModel(), step_size=0.01, seed=5, comm=comm, wide_eval=False)
        a.sample(show_progress=False)
        b = SMCSampler(K=6, N=8192, target=ArmaModel(), step_size=0.01, seed=5, wide_eval=False)
        b.sample(show_progress=False)
        np.testing.assert_array_equal(a.x_saved, b.x_saved)
        np.testing.assert_array_equal(a.ess, b.ess)
        np.testing.assert_array_equal(a.mean_estimate, b.mean_estimate)
        assert a.resampled == b.resampled and any(a.resampled)
        # the population all-gather of the global resampling ran through RCCL on device pointers
        assert a.samples.global_route == "device" and a.samples.global_resamplings == sum(a.resampled)
        # the step-by-step strategies take the same route
        kw = dict(K=4, N=2048, step_size=0.01, seed=7, lkernel="GaussianApproxLKernel", wide_eval=False)
        c = SMCSampler(target=ArmaModel(), comm=comm, **kw)
        c.sample(show_progress=False)
        d = SMCSampler(target=ArmaModel(), **kw)
        d.sample(show_progress=False)
        assert c.resampled == d.resampled and any(c.resampled)
        np.testing.assert_array_equal(c.x_saved, d.x_saved)
        # adaptive tempering with the shard protocol forced on: the ESS bisection takes the sharded device route
        # (smcn_temper_bisect_pass / _decide; the all-gather itself is skipped at world size 1)
        kw = dict(K=5, N=2048, step_size=0.01, seed=9, lkernel="forwardsLKernel", tempering=True, wide_eval=False)
        e = SMCSampler(target=ArmaModel(), comm=comm, **kw)
        e.sample(show_progress=False)
        f = SMCSampler(target=ArmaModel(), **kw)
        f.sample(show_progress=False)
        assert 0 < e.phi[0] < 1 and e.resampled == f.resampled
        np.testing.assert_allclose(e.phi, f.phi, rtol=0, atol=1e-11)
        np.testing.assert_allclose(e.ess, f.ess, rtol=1e-8)
        info = comm.info()
        assert info["world_seen"] == 1 and info["rank_seen"] == 0
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    _body()
    print("DIST-OK")


def test_in_library_rccl_world1_matches_plain_run():
    """The in-library communicator (smcn_comm_*: RCCL looked up at run time, no torch): world size 1 with the shard
    protocol forced on -- all-gather of the partials in the context's stream, and the routed global resampling
    (tile totals, all-to-all of keys and ancestor rows through ncclSend/ncclRecv) -- equals the plain run bit for bit."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = f"""
import sys, numpy as np
sys.path.insert(0, {root!r})
from smcnuts_amd import ArmaModel, SMCSampler
from smcnuts_amd.parallel import RcclComm
s = __import__("socket").socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
comm = RcclComm(rank=0, world_size=1, addr="127.0.0.1", port=port)
comm.force_exchange = True
a = SMCSampler(K=6, N=8192, target=ArmaModel(), step_size=0.01, seed=5, comm=comm, wide_eval=False)
a.sample(show_progress=False)
b = SMCSampler(K=6, N=8192, target=ArmaModel(), step_size=0.01, seed=5, wide_eval=False)
b.sample(show_progress=False)
np.testing.assert_array_equal(a.x_saved, b.x_saved)
np.testing.assert_array_equal(a.ess, b.ess)
np.testing.assert_array_equal(a.mean_estimate, b.mean_estimate)
assert a.resampled == b.resampled and any(a.resampled)
assert a.samples.global_route == "device" and a.samples.global_resamplings == sum(a.resampled)
comm2 = RcclComm(rank=0, world_size=1, addr="127.0.0.1", port=port)
comm2.force_exchange = True
kw = dict(K=4, N=2048, step_size=0.01, seed=7, lkernel="GaussianApproxLKernel", wide_eval=False)
c = SMCSampler(target=ArmaModel(), comm=comm2, **kw); c.sample(show_progress=False)
d = SMCSampler(target=ArmaModel(), **kw); d.sample(show_progress=False)
assert c.resampled == d.resampled and any(c.resampled)
np.testing.assert_array_equal(c.x_saved, d.x_saved)
info = comm2.info()                      # what the communicator itself reports (ncclCommCount / ncclCommUserRank)
assert info["world_seen"] == 1 and info["rank_seen"] == 0 and info["rccl_version"] > 0, info
comm3 = RcclComm(rank=0, world_size=1, addr="127.0.0.1", port=port)
comm3.force_exchange = True
kw = dict(K=5, N=2048, step_size=0.01, seed=9, lkernel="forwardsLKernel", tempering=True, wide_eval=False)
e = SMCSampler(target=ArmaModel(), comm=comm3, **kw); e.sample(show_progress=False)
f = SMCSampler(target=ArmaModel(), **kw); f.sample(show_progress=False)
assert 0 < e.phi[0] < 1 and e.resampled == f.resampled
np.testing.assert_allclose(e.phi, f.phi, rtol=0, atol=1e-11)
np.testing.assert_allclose(e.ess, f.ess, rtol=1e-8)
print("RCCL-OK")
"""
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "RCCL-OK" in p.stdout, p.stdout[-3000:] + p.stderr[-3000:]


def test_bench_contract_smoke():
    """bench.py prints ONE JSON line with the contract's keys (small shard, few steps)."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "6", "--warmup", "4", "--particles",
                        "8192", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["steps"] == 6 and d["warmup"] == 4 and d["n_gpus"] == 1 and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["metric"] == "leapfrog-steps/sec" and d["value"] > 0 and d["higher_is_better"] is True
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"])
    assert "workload" in d["config"] and "model" not in d["config"]


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no torchrun environment (how the driver calls it) starts its
    own two ranks before touching the GPU; here both share GPU 0 over gloo (one-GPU box).  One JSON
    line, n_gpus 2, aggregate over both shards."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR",
                                                              "MASTER_PORT")}
    env["SMCN_BENCH_SAME_DEVICE"] = "1"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps",
                        "6", "--warmup", "4", "--particles", "8192", "--no-cpu-baseline"], env=env,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["warmup"] == 4 and d["value"] > 0
    assert d["config"]["particles_total"] == 16384 and d["config"]["shard_exchange"] in ("host", "rccl-device")
    assert d["scaling"] == "weak"
    cm = d["config"]["comm"]                 # the line says how many ranks the COMMUNICATOR saw, and what each rank did
    assert cm["world_seen"] == 2 and cm["world_env"] == 2 and "gloo" in cm["backend"]
    assert [r["rank"] for r in cm["per_rank"]] == [0, 1] and all(r["leapfrogs"] > 0 and r["median_s"] > 0 for r in cm["per_rank"])
    assert sum(r["leapfrogs"] for r in cm["per_rank"]) * d["steps"] > 0


def test_bench_walks_down_the_backend_chain():
    """Two ranks forced onto ONE GPU with the default backend: the in-library RCCL communicator gets through its
    rendezvous (rank 0's id reaches rank 1 through the file keyed by the launch) and RCCL then refuses the duplicate
    device, torch's "nccl" group refuses it too, and the host exchange (gloo) carries the run -- the same decisions on
    both ranks, one JSON line, exit code 0."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR",
                                                              "MASTER_PORT")}
    env["SMCN_BENCH_SAME_DEVICE"] = "1"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "4",
                        "--particles", "8192", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["particles_total"] == 16384 and d["value"] > 0
    assert d["config"]["shard_exchange"] == "host"
    assert p.stderr.count("shard exchange over 'rccl' could not be set up") == 2
    assert "smcn_comm_init" in p.stderr          # the communicator got as far as RCCL's own device check
