#!/usr/bin/env python3
"""Headline benchmark: leapfrog-steps/s of the SMC-NUTS hot path on the arma
Stan model, N = 65 536 particles per GPU (BASELINE.json configs[1]; configs[2]
when launched on 8 GPUs: 524 288 particles, weak scaling).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one SMC iteration (normalise -> estimate -> ESS -> resample ->
NUTS proposal -> re-weight -> commit, smc_sampler.py:109-140) over the resident
particle shard.  W warm-up iterations advance the same chain untimed; then
exactly K iterations are timed between barrier + device synchronisation on both
sides; the maximum over ranks is taken and rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
FP64_VALU_PEAK_TFLOPS = 78.6   # MI355X vector fp64 (spec)
BYTES_PER_LEAPFROG = 48 * 4    # read+write x, r, grad in fp64, D = 4
FLOPS_PER_LEAPFROG = 12 * 4 + 4400


def csrc_hash():
    """sha256 over the kernel sources (smcnuts_amd/csrc/*, sorted by name): stamps measured-traffic entries, so that a
    profile taken on OTHER kernels is never quoted beside this build's timings."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "smcnuts_amd", "csrc", "*"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def measured_traffic(key):
    """HBM bytes of the timed NUTS launch from the committed PMC passes (profiles/r0*_traffic.json, newest first, written by
    tools/prof_round.sh) of this very command AND these very kernel sources; (None, reason) otherwise."""
    entries = []
    for name in ("r05_traffic.json", "r04_traffic.json", "r03_traffic.json"):
        try:
            entries += json.load(open(os.path.join(ROOT, "profiles", name)))["entries"]
        except Exception:
            pass
    if not entries:
        return None, "no profiles/r0*_traffic.json"
    here = csrc_hash()
    stale = False
    for ent in entries:
        k = (ent["config"], ent["N"], ent["steps"], ent["warmup"], ent["fuse_max"], ent.get("step_size"))
        if k == key or (k[:5] == key[:5] and k[5] is None):
            if ent.get("csrc_sha") == here:
                return ent["hbm_bytes_per_launch"], ent["source"]
            stale = True
    return None, ("entry measured on other kernel sources (csrc hash differs): re-run tools/prof_round.sh" if stale
                  else "no entry for this command")


def cpu_baseline(x_state, model_data, seed, budget_s=6.0):
    """CPU timings beside the GPU number, on the GPU run's own post-warm-up particles, in a
    child process that never touches the GPU (oracle/cpu_baseline.py): the C port of the NUTS proposal on one
    core (the reference is single-threaded: this is `cpu_baseline`), the same on all usable cores, and the
    reference-shaped serial Python loop (oracle/pynuts.py)."""
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        xs, ms = os.path.join(td, "x.npy"), os.path.join(td, "md.npy")
        np.save(xs, np.ascontiguousarray(x_state))
        np.save(ms, np.ascontiguousarray(model_data))
        p = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), xs, ms, str(seed),
                            str(budget_s)], capture_output=True, text=True)
    if p.returncode != 0:
        raise RuntimeError("cpu_baseline failed: " + p.stderr[-1000:])
    r = json.loads(p.stdout.strip().splitlines()[-1])
    host = f"{r['cpu_model']}, nproc {r['nproc']}, {r['usable_cores']} usable"
    c1 = r["c_one_core"]
    out = {"value": c1["value"], "unit": "leapfrog/s", "cores": 1, "kind": "port",
           "sample": f"oracle/smcnuts_oracle.c NUTS proposal (C port, one thread: the reference is single-threaded), "
                     f"{c1['sample']}, particles from the GPU run's post-warm-up state; host: {host}",
           "cpu_model": r["cpu_model"], "nproc": r["nproc"],
           "variants": {"c_all_cores": r["c_all_cores"], "python_serial_one_core": r["python_serial"]}}
    return out


EXTRA_CONFIGS = {
    # BASELINE configs[3], configs[4] (per-GPU share, both step sizes of DESIGN.md 4.2) and the arma kernel at two particles
    # per lane: each one is this file's own command line, run in a child process after the headline has been measured
    "c4": ["--config", "c4", "--steps", "10", "--warmup", "12", "--repeats", "3"],
    "c5_eps025": ["--config", "c5", "--steps", "6", "--warmup", "2", "--step-size", "0.25", "--repeats", "3"],
    "c5_eps01": ["--config", "c5", "--steps", "6", "--warmup", "2", "--step-size", "0.1", "--repeats", "3"],
    "arma_n131072": ["--particles", "131072", "--steps", "20", "--warmup", "5", "--repeats", "3"],
}


def extra_configs(timeout_s=150.0):
    """The other BASELINE configurations, timed by the SAME driver command as the headline: one child process each (this
    file with the arguments above; the parent has released the GPU by then and holds no context), its JSON line cut down to
    the figures and merged under `configs`.  A child that fails or overruns leaves {"error": ...}: the headline stands."""
    import subprocess
    res = {}
    for name, argv in EXTRA_CONFIGS.items():
        cmd = [sys.executable, os.path.abspath(__file__)] + argv + ["--no-cpu-baseline", "--no-end-to-end", "--no-extra-configs",
                                                                     "--no-peaks"]
        t0 = time.perf_counter()
        try:
            pr = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s)
            lines = [ln for ln in pr.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
            if pr.returncode != 0 or not lines:
                res[name] = {"error": f"rc {pr.returncode}: " + pr.stderr.strip()[-300:], "command": " ".join(argv)}
                continue
            d = json.loads(lines[-1])
        except subprocess.TimeoutExpired:
            res[name] = {"error": f"no result within {timeout_s:.0f} s", "command": " ".join(argv)}
            continue
        rf = d["roofline"]
        res[name] = {
            "command": "python bench.py " + " ".join(argv), "value": d["value"], "unit": d["unit"], "steps": d["steps"],
            "warmup": d["warmup"], "ms_per_step": d["ms_per_step"], "dtype": d["dtype"],
            "workload": d["config"]["workload"], "particles_per_gpu": d["config"]["particles_per_gpu"],
            "nuts_cap": d["config"]["nuts_cap"],
            "leapfrogs_per_particle_step": d["leapfrogs_per_particle_step"],
            "repeats": {k: d["repeats"][k] for k in ("n", "median_s", "min_s", "max_s", "value_min", "value_max")},
            "nuts_kernel_share_of_step": d["nuts_kernel_share_of_step"],
            "roofline": {k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "kernel",
                                                 "avg_launch_ms", "launches", "algorithmic_bytes_per_leapfrog",
                                                 "achieved_basis", "valu_f64_tflops", "valu_f64_frac")},
            "child_wall_s": time.perf_counter() - t0,
        }
        if "phi_first_last" in d:
            res[name]["phi_first_last"] = d["phi_first_last"]
    return res


def cold_first_sampler(n_particles, eps, keep_hist, wide, timeout_s=120.0):
    """What the FIRST sampler of a fresh process costs: a child process that has created no stream, no
    context and no buffer before builds SMCSampler(K=50, N, arma) and runs sample() once."""
    import subprocess
    code = (
        "import json, sys, time\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "t_imp = time.perf_counter()\n"
        "from smcnuts_amd import ArmaModel, SMCSampler\n"
        "t0 = time.perf_counter()\n"
        f"s = SMCSampler(K=50, N={int(n_particles)}, target=ArmaModel(), step_size={float(eps)!r}, seed=11, "
        f"save_history={bool(keep_hist)}, wide_eval={bool(wide)})\n"
        "t1 = time.perf_counter()\n"
        "s.sample(show_progress=False)\n"
        "t2 = time.perf_counter()\n"
        "print(json.dumps({'import_s': t0 - t_imp, 'construct_first_in_process_s': t1 - t0, 'sample_wall_s': t2 - t1, "
        "'run_time_s': float(s.run_time), 'leapfrogs': int(s.leapfrogs.sum())}))\n")
    try:
        pr = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout_s)
        if pr.returncode != 0:
            return {"error": f"rc {pr.returncode}: " + pr.stderr.strip()[-300:]}
        return json.loads(pr.stdout.strip().splitlines()[-1])
    except subprocess.TimeoutExpired:
        return {"error": f"no result within {timeout_s:.0f} s"}


def launch_ranks(n, timeout_s):
    """`python bench.py --gpus N` without a launcher: start the N ranks (torch.distributed.run, one per GPU) in their own
    process group, relay rank 0's JSON line and the exit code.  The whole run has a deadline: a rendezvous or collective
    that never returns ends as a non-zero exit with the ranks' output, not as a hang the caller has to kill."""
    import signal
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)   # (stderr: passed through)
    try:
        out, _ = p.communicate(timeout=timeout_s)
        rc = p.returncode
    except subprocess.TimeoutExpired:
        for sig in (signal.SIGTERM, signal.SIGKILL):       # the launcher AND its ranks: exactly the group started above
            try:
                os.killpg(p.pid, sig)
            except ProcessLookupError:
                break
            try:
                p.wait(timeout=10)
                break
            except subprocess.TimeoutExpired:
                continue
        out, _ = p.communicate()
        print(f"bench.py: the {n} ranks did not finish within {timeout_s:.0f} s (--launch-timeout); their output follows",
              file=sys.stderr)
        rc = 124
    lines = [ln for ln in out.splitlines() if ln.startswith("{") and '"metric"' in ln]
    if rc == 0 and lines:
        print(lines[-1])
    else:
        sys.stderr.write(out[-8000:] + ("\n" if out and not out.endswith("\n") else ""))
    sys.stdout.flush()
    if rc == 0 and not lines:
        print("bench.py: the ranks printed no result line", file=sys.stderr)
        return 1
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--particles", type=int, default=65536, help="particles per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--history", action="store_true",
                    help="also time the K iterations with x_saved / logw_saved DOWNLOADED inside the clock (pcie_inclusive_value)")
    ap.add_argument("--no-history", action="store_true",
                    help="save_history=False (generation ring instead of the device-side x_saved / logw_saved of every generation)")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the cold SMCSampler(K=50).sample() line (end_to_end)")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="arma, one GPU: do not time configs[3], configs[4] and the N = 131 072 point in child processes (`configs`); "
                         "profiling / diagnostic runs (--no-cpu-baseline) never do")
    ap.add_argument("--no-peaks", action="store_true", help="skip smcn_measure_peaks (roofline.peak_measured)")
    ap.add_argument("--settle-ms", type=float, default=250.0,
                    help="untimed repeats of the K-iteration block for this long before the timed repeats (clock ramp); 0 = none")
    ap.add_argument("--repeats", type=int, default=5, help="times the K timed iterations are repeated from the saved state")
    ap.add_argument("--config", default="arma", choices=["arma", "c4", "c5"],
                    help="arma: BASELINE configs[1]/[2] (default, the headline); c4: PRMwCD, Gaussian L-kernel + adaptive "
                         "tempering (BASELINE configs[3]); c5: iso-Gaussian D=256, 131072 particles per GPU "
                         "(BASELINE configs[4], the HBM-roofline configuration)")
    ap.add_argument("--step-size", type=float, default=None)
    ap.add_argument("--fuse-max", type=int, default=64,
                    help="max SMC iterations per NUTS launch (speculative, rolled back on resampling); 1 = off")
    ap.add_argument("--launch-timeout", type=float, default=900.0,
                    help="--gpus N without a launcher: seconds before the ranks started here are killed (exit code 124)")
    ap.add_argument("--no-wide", action="store_true",
                    help="arma: every evaluation by one lane (smcn_set_wide_eval 0; A/B of the lane-group evaluation of stragglers)")
    ap.add_argument("--nuts-cap", type=int, default=None,
                    help="c4: doublings of the first NUTS launch (longer trees are finished by a second one); 0 = one launch; "
                         "default: the sampler's own choice")
    ap.add_argument("--nuts-requeue", type=int, default=0,
                    help="with --nuts-cap: an inner park level of the first launch (smcn_set_nuts_requeue; 0 = none)")
    ap.add_argument("--lane-segments", type=int, default=None,
                    help="arma, more particles than lanes: segments a block is handed on in (default: the launcher's rule)")
    ap.add_argument("--no-widen", action="store_true", help="c4: the second launch uses the kernel of the first")
    ap.add_argument("--shard-resampling", default="global", choices=["global", "local"],
                    help="several GPUs: resample over the whole population (reference semantics) or per shard")
    ap.add_argument("--backend", default="rccl",
                    help="shard exchange: rccl = the library's own RCCL communicator (default, no torch.distributed); "
                         "nccl = torch.distributed over RCCL; gloo = torch.distributed on the host (rehearsing ranks on one GPU)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` as the driver calls it: this process only starts the N ranks
        # (one per GPU, torch.distributed.run) and relays rank 0's JSON line and the exit code.  It
        # never touches the GPU itself (no torch import, no HIP call) and re-execs nothing.
        return launch_ranks(args.gpus, args.launch_timeout)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("SMCN_BENCH_SAME_DEVICE") == "1":   # rehearsal: every rank on GPU 0 (needs --backend gloo)
        local_rank = 0
    torch = None             # the product is torch-free (ctypes + the library's own stream); only --backend nccl / gloo use it
    if world > 1 and args.backend != "gloo":
        # Several ranks: the backend chain may fall back from the in-library RCCL to torch.distributed, and torch ships an
        # RCCL of its own.  Loading it AFTER the library has dlopen'ed the system's gives the process two RCCLs (the ranks
        # then abort in static destructors at exit); imported first, the library's dlopen("librccl.so.1") resolves to the
        # copy that is already there.  (Module import only: no device is touched here.)
        import torch             # noqa: F811
    import __graft_entry__ as ge
    ge.build()              # a no-op when the library is current; several ranks serialise on a file lock
    comm = None
    dist = None

    def make_comm(backend):
        """rccl: the library's own communicator (RCCL behind the C ABI, no torch.distributed; bound to the sampler's
        context when that exists); nccl: torch.distributed over RCCL; gloo: torch.distributed on the host."""
        if backend == "rccl":
            from smcnuts_amd import _capi
            from smcnuts_amd.parallel import RcclComm
            _capi.lib()
            return RcclComm(), None
        nonlocal torch
        import torch
        import torch.distributed as td
        from smcnuts_amd.parallel import TorchDistComm
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            td.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            return TorchDistComm(torch.device("cuda", local_rank)), td
        td.init_process_group(backend)
        return TorchDistComm(torch.device("cpu")), td

    from smcnuts_amd import ArmaModel, IsoGaussian, PRMwCDModel, SMCSampler

    K, W, NP, R = args.steps, args.warmup, args.particles, max(1, args.repeats)
    if args.config == "c5":
        D, eps = 256, (args.step_size or 0.25)
        if args.particles == 65536:
            NP = 131072
        target = IsoGaussian(D)
        args.no_cpu_baseline = True
        args.history, args.no_history = False, True     # x_saved would be 268 MB per generation
        global BYTES_PER_LEAPFROG, FLOPS_PER_LEAPFROG
        BYTES_PER_LEAPFROG, FLOPS_PER_LEAPFROG = 48 * D, 15 * D
    elif args.config == "c4":
        D, eps = 13, (args.step_size or 0.01)
        target = PRMwCDModel()
        args.no_cpu_baseline = True
        args.history, args.no_history = False, True
        # (the step-by-step strategies synchronise with the host every iteration and have no checkpoint: every repeat is a
        #  fresh sampler with the same seed -- the same chain --, warmed up untimed)
        if world > 1:
            R = 1
        BYTES_PER_LEAPFROG, FLOPS_PER_LEAPFROG = 48 * D, 12 * D + 7000
    else:
        D, eps = 4, (args.step_size or 0.01)
        target = ArmaModel()
    seed = 10
    stepwise = args.config == "c4"
    # the reference keeps x_saved / logw_saved of every generation (smc_sampler.py:139-140): so does the headline run, on
    # the device; their download is outside the clock (pcie_inclusive_value has it inside)
    keep_hist = not args.no_history
    def sampler(cm):
        return SMCSampler(K=W + K, N=NP * world, target=target, step_size=eps,
                          lkernel="GaussianApproxLKernel" if stepwise else "forwardsLKernel",
                          tempering=stepwise, seed=seed, comm=cm, device=local_rank, save_history=keep_hist,
                          shard_resampling=args.shard_resampling, wide_eval=not args.no_wide,
                          nuts_cap="auto" if args.nuts_cap is None else (args.nuts_cap, not args.no_widen, args.nuts_requeue))

    if world == 1:
        smc = sampler(None)
        if args.lane_segments is not None:
            smc.samples.ctx.call("smcn_set_lane_segments", args.lane_segments)
    else:
        # a backend that cannot be set up fails the same way on every rank (no RCCL library, RCCL refusing the
        # devices, ...), so every rank walks down the same chain: in-library RCCL, torch "nccl", host exchange
        chain = {"rccl": ["rccl", "nccl", "gloo"], "nccl": ["nccl", "gloo"]}.get(args.backend, [args.backend])
        smc = None
        for be in chain:
            try:
                comm, dist = make_comm(be)
                smc = sampler(comm)
                args.backend = be
                break
            except Exception as e:                      # noqa: BLE001
                print(f"bench.py: shard exchange over '{be}' could not be set up ({e})", file=sys.stderr)
                import torch.distributed as td
                if td.is_available() and td.is_initialized():
                    td.destroy_process_group()
                comm = dist = None
        if smc is None:
            raise SystemExit("bench.py: no shard exchange backend could be set up")
    ctx = smc.samples.ctx
    fusable = ctx.fused_transitions and args.fuse_max > 1 and not stepwise

    def advance(upto):
        if stepwise:
            while smc.k < upto:
                smc.step()
        elif fusable:
            smc.run_fused(upto=upto, fuse_max=args.fuse_max)   # several iterations per NUTS launch (bit-identical results)
        else:
            while smc.k < upto:
                smc.step_async()

    def fence():
        if world > 1:
            comm.barrier()
        ctx.call("smcn_synchronize")
        if torch is not None and dist is not None and args.backend == "nccl":
            torch.cuda.synchronize()

    # warm-up: W iterations of the same chain, untimed; the state after it is saved, and the SAME K
    # iterations are then timed R times from that state (identical work every time: same Philox keys)
    advance(W)
    ck = None if stepwise else smc.checkpoint()
    # clock settling: the card ramps its clocks over the first ~100 ms of sustained work (the first block after the
    # short warm-up runs 8-10 % slower than the sixth); the same K iterations are run untimed from the saved
    # state until --settle-ms of wall time has passed, the first of them is reported as `cold_block_s`
    settle = []
    t_settle = time.perf_counter()
    while ck is not None and args.settle_ms > 0:
        smc.restore(ck)
        fence()
        t0 = time.perf_counter()
        advance(W + K)
        smc.finalise_async(download_history=False)
        fence()
        settle.append(time.perf_counter() - t0)
        spent = time.perf_counter() - t_settle
        if world > 1:
            spent = float(comm.allgather(np.array([spent])).max())
        if spent * 1e3 >= args.settle_ms or len(settle) >= 64:
            break
    runs = []
    for rep in range(R):
        if ck is not None:
            smc.restore(ck)
        elif rep > 0:
            ctx.close()
            smc = sampler(None)
            ctx = smc.samples.ctx
            advance(W)
        ctx.timers(reset=True)
        fence()
        t0 = time.perf_counter()
        advance(W + K)
        if stepwise:
            smc.finalise()
        else:
            smc.finalise_async(download_history=False)   # closing normalise/estimate/ESS + ONE sync + scalar history
        fence()
        dt = time.perf_counter() - t0
        tm = ctx.timers()
        runs.append(dict(dt=dt, nuts_ms=float(tm[0]), launches=max(int(tm[1]), 1),
                         leaps=int(smc.leapfrogs[W:].sum())))
    assert len({r_["leaps"] for r_ in runs}) == 1, "the repeats did not do the same work"
    dt_pcie = None
    if args.history and keep_hist and world == 1:      # the same K iterations once more, x_saved / logw_saved downloaded inside the clock
        smc.restore(ck)
        fence()
        t0 = time.perf_counter()
        advance(W + K)
        smc.finalise_async(download_history=True)
        fence()
        dt_pcie = time.perf_counter() - t0
    leaps_local = runs[0]["leaps"]
    dts = np.array([r_["dt"] for r_ in runs])
    per_rank = None
    if world > 1:
        nm = np.array([r_["nuts_ms"] for r_ in runs])
        both = comm.allgather(np.concatenate([dts, [float(leaps_local)], nm]))
        # every rank's own clock and work, so that a straggler rank shows in the line (the headline takes the slowest)
        per_rank = [{"rank": i, "leapfrogs": int(both[i, R]), "median_s": float(np.median(both[i, :R])),
                     "nuts_kernel_ms_median": float(np.median(both[i, R + 1:]))} for i in range(world)]
        dts = both[:, :R].max(axis=0)             # slowest rank, per repeat
        leaps_total = int(both[:, R].sum())
    else:
        leaps_total = leaps_local
    order = np.argsort(dts)
    med = int(order[len(order) // 2])             # the median repeat: its launch timings go into `roofline`
    dt = float(dts[med])

    if rank == 0:
        traffic, traffic_src = (None, "several ranks") if world != 1 else measured_traffic(
            (args.config, NP, K, W, args.fuse_max if fusable else 1, eps))
        nuts_ms, launches = runs[med]["nuts_ms"], runs[med]["launches"]
        avg_kernel_s = nuts_ms / launches / 1e3
        leaps_per_launch = leaps_local / launches
        achieved = leaps_per_launch * BYTES_PER_LEAPFROG / avg_kernel_s / 1e9
        model_gbs = None
        if args.config == "c5":
            # The 48 D bytes-per-leapfrog model assumes (x, r, grad) round-trip HBM once per step; this kernel keeps
            # them in registers and only the tree-stack levels >= 3 travel, so the model can exceed the HBM peak.  The
            # fraction reported here is therefore the MEASURED traffic of this very command (committed PMC passes)
            # over the launch time; the model rate is kept beside it.
            model_gbs = achieved
            achieved = (traffic / avg_kernel_s / 1e9) if traffic else None   # no measured bytes for this command: no fraction
        two_phase = bool(smc.samples.nuts_cap)
        kname = {"arma": "nuts3_kernel<ArmaLaneModel,false,3,3," + ("true" if NP > 65536 else "false") + ">",
                 "c4": ("nuts_kernel<PrmwcdDistModel<8,100,11,2,4,true>,true,true> (trees up to 9 doublings) + "
                        "nuts_kernel<PrmwcdDistModel<64,100,11,2,5,true>,true,true> (the parked longer trees): avg_launch_ms averages both"
                        if two_phase else "nuts_kernel<PrmwcdDistModel<8,100,11,2,4,true>,true,false>"),
                 "c5": "nuts_wave_kernel<GaussModel<64,4>,full,no_likelihood>"}[args.config]
        out = {
            "metric": "leapfrog-steps/sec", "value": leaps_total / dt, "unit": "leapfrog/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": ("synthetic (model data shipped with the reference; x0 ~ N(0,I), Philox seed 10)" if args.config != "c5"
                     else "synthetic (x0 ~ N(0,I), Philox seed 10)"),
            "config": {"workload": (f"arma Stan model, N={NP} particles per GPU, fp64, forwardsLKernel, no tempering, "
                                    f"step_size=0.01, save_history={keep_hist} (x_saved / logw_saved of every generation kept on the device, "
                                    "downloaded outside the clock) (BASELINE configs[1]; configs[2] at 8 GPUs); "
                                    "timed = the whole sample()-equivalent of K iterations incl. the closing "
                                    "normalise/estimate/ESS and the download of the scalar history")
                                   if args.config == "arma" else
                                   (f"PRMwCD Stan model (D=13), N={NP} particles, fp64, GaussianApproxLKernel + adaptive (ESS) "
                                    f"tempering, step_size={eps}, save_history=False (BASELINE configs[3]); step-by-step loop: "
                                    "L-kernel algebra and tempering bisection on the device; NUTS in two launches when nuts_cap is set")
                                   if args.config == "c4" else
                                   (f"iso-Gaussian D=256 (device-native), N={NP} particles per GPU, fp64, forwardsLKernel, "
                                    f"step_size={eps}, save_history=False (BASELINE configs[4])"),
                       "particles_per_gpu": NP, "particles_total": NP * world, "K": K,
                       "save_history": bool(keep_hist), "wide_eval": not args.no_wide,
                       "nuts_cap": smc.samples.nuts_cap,
                       "iterations_per_nuts_launch_max": args.fuse_max if fusable else 1,
                       "parallelism": f"particle-shard x{world}",
                       "shard_resampling": "n/a" if world == 1 else args.shard_resampling,
                       "resamplings_in_timed_steps": int(sum(smc.resampled[W:W + K])),
                       "comm": ({"backend": "none", "world_seen": 1} if world == 1 else
                                dict(comm.info(), world_env=world, per_rank=per_rank)),
                       "shard_exchange": ("none" if world == 1 else (("rccl-in-library" if dist is None else "rccl-device (torch.distributed)")
                                                                      if getattr(comm, "device_path", False) else "host"))},
            "repeats": {"n": R, "median_s": dt, "min_s": float(dts.min()), "max_s": float(dts.max()),
                        "all_s": [float(v) for v in dts], "nuts_kernel_ms": [r_["nuts_ms"] for r_ in runs],
                        "value_min": leaps_total / float(dts.max()), "value_max": leaps_total / float(dts.min()),
                        "untimed_settle_blocks": len(settle), "cold_block_s": settle[0] if settle else None,
                        "note": "K timed iterations repeated from one saved post-warm-up state; value = median; "
                                "before them the same block runs untimed for --settle-ms so that the clocks have ramped"},
            "ess_per_sec": float(smc.ess[-1]) / dt,
            "mean_ess_times_steps_per_sec": float(np.mean(smc.ess[W + 1:])) * K / dt,   # SURVEY 8(d), second definition
            "final_ess": float(smc.ess[-1]),
            "leapfrogs_per_particle_step": leaps_total / (K * NP * world),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved is not None else None,
                         "traffic": traffic, "traffic_source": traffic_src, "csrc_sha": csrc_hash(),
                         "kernel": kname, "avg_launch_ms": avg_kernel_s * 1e3,
                         "launches": launches, "algorithmic_bytes_per_leapfrog": BYTES_PER_LEAPFROG,
                         "algorithmic_model_gbs": model_gbs,
                         "achieved_basis": ("measured HBM bytes of the launch (profiles/r0*_traffic.json)" if model_gbs and traffic
                                            else ("none: the 48 D bytes-per-leapfrog model exceeds the HBM peak for this register-resident "
                                                  "kernel and no measured traffic matches this command" if model_gbs
                                                  else "algorithmic bytes per leapfrog x leapfrogs of the launch")),
                         "valu_f64_tflops": leaps_per_launch * FLOPS_PER_LEAPFROG / avg_kernel_s / 1e12,
                         "valu_f64_frac": leaps_per_launch * FLOPS_PER_LEAPFROG / avg_kernel_s / 1e12
                                          / FP64_VALU_PEAK_TFLOPS},
            "nuts_kernel_share_of_step": nuts_ms / 1e3 / dt,
            "pcie_inclusive_value": (leaps_total / dt_pcie) if dt_pcie else None,
        }
        if world == 1 and args.config in ("arma", "c5"):
            # second roofline entry: the resampling kernels (no generation of the timed steps resamples in steady
            # state), timed on the final weights: 28 + 8 log2(N) + 16 D algorithmic bytes per particle
            import ctypes as C
            reps = 50 if args.config == "arma" else 10
            ms = C.c_double(0.0)
            ctx.call("smcn_bench_resample", reps, 1000, C.byref(ms))
            bpp = 28 + 8 * np.log2(NP) + 16 * D
            ach = bpp * NP * reps / (ms.value / 1e3) / 1e9
            out["roofline_resample"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": ach / HBM_PEAK_GBS, "traffic": None,
                                        "kernel": "scan_tile_kernel + scan_offsets_kernel + search_gather_kernel"
                                                  + (" + gather_rows_kernel" if D >= 16 else ""),
                                        "avg_resample_us": ms.value / reps * 1e3,
                                        "algorithmic_bytes_per_particle": float(bpp), "repetitions": reps}
        if world == 1 and args.config == "arma" and not args.no_end_to_end:
            # what a user gets: ONE cold SMCSampler(K=50).sample() from construction (smc_sampler.py:101-155) -- constructor,
            # the degenerate first generations with their resamplings, rolled-back speculative launches, x_saved downloaded
            def cold_run(sd):
                import cProfile, gc, pstats
                pr = cProfile.Profile() if os.environ.get("BENCH_PROFILE_CONSTRUCT") else None
                t0 = time.perf_counter()
                if pr: pr.enable()
                cold = SMCSampler(K=50, N=NP, target=ArmaModel(), step_size=eps, seed=sd, save_history=keep_hist,
                                  wide_eval=not args.no_wide)
                if pr:
                    pr.disable()
                    pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(12)
                t1 = time.perf_counter()
                cold.sample(show_progress=False)
                t2 = time.perf_counter()
                lf = int(cold.leapfrogs.sum())
                ctm = cold.samples.ctx.timers()
                res = {"construct_s": t1 - t0, "run_time_s": float(cold.run_time), "sample_wall_s": t2 - t1,
                       "nuts_kernels_s": float(ctm[0]) / 1e3, "nuts_launches": int(ctm[1]),
                       "leapfrogs": lf, "value_over_run_time": lf / float(cold.run_time),
                       "value_over_construct_plus_sample": lf / (t2 - t0),
                       "resamplings": int(sum(cold.resampled)), "discarded_launches": int(cold.discarded_launches),
                       "final_ess": float(cold.ess[-1]),
                       "history_rows_downloaded_beside_the_loop": int(cold._dl_upto) if keep_hist else 0}
                cold.samples.ctx.close()
                del cold
                gc.collect()          # (the 134 MB of its history are unmapped here, not inside the next run's clock)
                return res
            first = cold_run(seed + 1)
            out["end_to_end"] = dict({"what": f"SMCSampler(K=50, N={NP}, arma, save_history={keep_hist}).sample() from construction: every "
                                              "iteration from x0 ~ N(0, I), x_saved / logw_saved downloaded (rows of validated blocks beside the "
                                              "loop, the last block's behind it); the constructor allocates every device buffer of the loop.  "
                                              "construct_s is that of a WARM process (this process has a context already: streams come from "
                                              "the library's pool); a process's first sampler pays the stream creation: "
                                              "first_in_fresh_process.construct_first_in_process_s, measured in a child process"},
                                     **first)
            out["end_to_end"]["first_in_fresh_process"] = cold_first_sampler(NP, eps, keep_hist, not args.no_wide)
            # the same once more (streams and device buffers of the first one are reused: smcn_api.hip's pools)
            out["end_to_end_second"] = cold_run(seed + 2)
        if world == 1 and not args.no_peaks:
            # the denominators measured on THIS box in THIS run: streaming copy, fp64 FMA issue at the NUTS
            # kernel's occupancy (one wavefront per SIMD for the lane kernel) and at four wavefronts per SIMD
            import ctypes as C
            pk = (C.c_double * 3)()
            ctx.call("smcn_measure_peaks", pk)
            rf = out["roofline"]
            occ1 = args.config == "arma"
            rf["peak_measured"] = {"copy_GBs": pk[0], "fp64_fma_tflops_1_wave_per_simd": pk[1],
                                   "fp64_fma_tflops_4_waves_per_simd": pk[2],
                                   "kernel_occupancy_waves_per_simd": 1 if occ1 else 2,
                                   "frac_of_measured_copy": (rf["achieved"] / pk[0]) if rf["achieved"] is not None else None,
                                   "valu_f64_frac_of_measured": rf["valu_f64_tflops"] / (pk[1] if occ1 else pk[2]),
                                   "how": "smcn_measure_peaks in this process: 2 x 1 GiB copy, 16 independent FMA chains per lane"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(ck["x"], target.model_data, seed)
        if stepwise:
            out["phi_first_last"] = [float(smc.phi[0]), float(smc.phi[-1])]
        if world == 1 and args.config == "arma" and NP == 65536 and not args.no_extra_configs and not args.no_cpu_baseline:
            ctx.close()                       # the children get the whole card
            out["configs"] = extra_configs()
        print(json.dumps(out))
    if world > 1:
        comm.barrier()
        if dist is not None:
            dist.destroy_process_group()
        else:
            comm.close()


if __name__ == "__main__":
    sys.exit(main() or 0)
