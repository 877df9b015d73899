#!/usr/bin/env python3
"""Headline benchmark: leapfrog steps/s (and ESS/s) of adaptive NUTS on the poll model.

Default workload = BASELINE.json configs[1]: scripts/model/final_2016.R's posterior (51 states x 254 days, 1619 polls,
D = 15 098), 8 chains per MI355X, 1000 warm-up + 1000 sampling iterations, seed 1843, NUTS diag_e, delta 0.8,
max depth 10.

A STEP is `--chunk` (100) NUTS transitions of every chain on the GPU -- the granularity at which the reference's own
driver reports progress (`refresh`, final_2016.R:11,540).  `--steps K` runs the first K // 2 steps as warm-up and the
rest as sampling from a fresh initialisation, so the default K = 20 IS the configuration above (8 chains x (1000 +
1000)); `--warmup W` first runs W untimed steps on a throw-away sampler (clocks, code objects, allocator).
How the K steps reach the GPU is `--launch-steps L`: L steps per potus_run call = per kernel launch.  The default (0) is
what a caller without a progress display does -- potus_run(handle, num_warmup) and potus_run(handle, num_samples): ONE
launch per phase -- because a launch lasts as long as its slowest chain, and the chains of a short launch wait for each
other twenty times instead of twice (same draws either way; 650 k against 625 k leapfrogs/s, profiles/r06_launch_size.txt).
`--launch-steps 1` is rounds 1-6's one launch per step.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config 1|2|3|4]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset) starts its own N ranks -- it re-executes itself
under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` on 127.0.0.1 with a free port, one rank per GPU over the "nccl"
backend (= RCCL), rank 0 prints the ONE JSON line and the parent returns the ranks' exit status.  Fewer than N visible GPUs is an error
(exit status 2), never an n_gpus = 1 line.  HSA_ENABLE_IPC_MODE_LEGACY=0 is set (if unset) before the first HIP call: the host driver of
these boxes only supports dmabuf IPC and RCCL's intra-node transport fails without it.

--config 1/2  2016 backtest, C = 8 chains per GPU (configs[1]; with N GPUs configs[2]: 8 N chains, one RCCL all-gather
              of the draws-of-interest for pooled R-hat / ESS, device buffers end to end)
--config 0    the reference's own sampler calls as scripted (6 chains x (500 + 500), seed 1843: final_2016.R, final_2012.R, final_2008.R;
              BASELINE's 4 x 500), each run to completion on the GPU and on the CPU port, side by side (one GPU)
--config 3    2008 + 2012 + 2016 backtests concurrently, 4 chains of each per GPU (32 each over 8 GPUs), three
              posteriors advancing together under potus_run_many
--config 4    synthetic stress posterior (51 states x 600 days x 10 000 polls, D = 41 610), dense metric

Each chain runs on a cluster of K workgroups (potus_cluster.hpp); --cus-per-chain 1 selects the one-workgroup-per-chain
kernel.  N > 1: one process per GPU; rank r owns chains [r C, (r + 1) C) (RNG streams keyed by global chain id), no
communication while sampling.  Weak scaling: C chains per GPU whatever N is.

--single-process  (with --gpus N, configs[1] / [2]) the path the reference-side binding takes -- R is one process: potus_sample(gpus =
              0:7) --: ONE process, N handles on N devices advancing together under potus_run_many, R-hat / ESS of the pooled chains
              through potus_diagnostics (the other GPUs' blocks come over with peer copies); no torch.distributed, no RCCL.  The line says
              "launcher": "single_process".  Under the launcher (N > 1) rank 0 adds this run as side.single_process once the ranks have
              released their GPUs, so that a scaling run times both paths.

Rank 0 prints ONE JSON line.  `value` = leapfrogs of all chains on all GPUs / max-over-ranks wall time of
(init + K chunks [+ all-gather]), inputs already resident in HBM.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def algorithmic_bytes_per_leapfrog(d, variant="full", dense=False):
    """48 D + 36 N_state + 32 N_nat + 24 S^2 + 16 S (diag metric).  Dense metric: the survey adds
    8 D^2 (M^-1 read as a full matrix); M^-1 is symmetric and is stored and read as its upper triangle: + 4 D^2."""
    from us_potus_model_amd import _abi
    D = _abi.num_params(d, variant)
    S = int(d["S"])
    per_state, per_nat = (36, 32) if variant == "full" else (28, 24)
    b = 48 * D + per_state * int(d["N_state_polls"]) + per_nat * int(d["N_national_polls"]) + 24 * S * S + 16 * S
    return b + (4 * D * D if dense else 0)


def measured_traffic(kernel, sides=1):
    """HBM bytes per leapfrog from the committed rocprofv3 PMC passes of this command (profiles/*pmc_traffic.json,
    scripts/profile_round.sh): FETCH_SIZE and WRITE_SIZE cannot be collected from inside the benchmark, so the bench
    line quotes the per-leapfrog figure of the latest committed pass for the same kernel -- a committed constant x this
    run's rate, labelled as such."""
    best = None
    for f in sorted((ROOT / "profiles").glob("*pmc_traffic.json")):
        try:
            d = json.loads(f.read_text())
        except (OSError, ValueError):
            continue
        if d.get("kernel") == kernel and (kernel != "k_cl_run" or ("two clusters" in d.get("command", "")) == (sides == 2)):
            best = (f.name, d)
    return best


# ------------------------------------------------------------------------------------------------ N ranks
def visible_gpus():
    """HIP devices this process would see, counted in a child process so that the parent never initialises the runtime before its ranks
    do."""
    import subprocess
    try:
        out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count() if torch.cuda.is_available() else 0)"],
                             capture_output=True, text=True, timeout=300)
        return int(out.stdout.strip().splitlines()[-1])
    except Exception:
        return 0


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: N ranks of this very command under torch.distributed.run (see the module
    docstring)."""
    import socket
    import subprocess
    have = visible_gpus()
    if os.environ.get("POTUS_DIST_BACKEND") != "gloo" and have < n:
        print(f"bench.py: --gpus {n} but {have} HIP device(s) visible on this box: refusing to run (no line is printed for a run that "
                f"did not happen)", file=sys.stderr)
        return 2
    if have < 1:
        print("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port),
           str(Path(__file__).resolve())] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


# ------------------------------------------------------------------------------------------------ the other configurations, beside the
# default line
def launch_groups(steps, warm_steps, launch_steps):
    """The K steps as potus_run calls: how many steps each launch covers.  A launch never straddles the end of the warm-up (the line
    reports the sampling phase's time on its own); launch_steps = 0: one launch per phase."""
    out = []
    for n in (warm_steps, steps - warm_steps):
        per = launch_steps if launch_steps > 0 else n
        while n > 0:
            out.append(min(per, n))
            n -= out[-1]
    return out


def seed_summary(entry):
    """What `side.seeds` keeps of one configs[1] run: rate, time, ESS / s and how deep each chain's trees go in the sampling phase."""
    post = next(iter(entry["config"]["posteriors"].values()))
    return {"seed": entry["seed"], "leapfrogs_per_sec": entry["value"], "seconds": entry["seconds"], "leapfrogs": entry["leapfrogs"],
            "ess_per_sec": entry.get("ess_per_sec"), "ess_bulk_min": entry.get("ess_bulk_min"), "rhat_max": entry.get("rhat_max"),
            "treedepth_max_per_chain": post.get("treedepth_max_per_chain"), "stepsize_per_chain": post.get("stepsize_per_chain")}


def side_measurements(seed, budget_s=200.0, headline=None):
    """BASELINE's other single-GPU configurations under the same clock as the default line: each is this very script
    with --config X in a child process (GPU only: no CPU baseline, no saturated point, no side measurements of its own), after the timed
    region and after the default run's handles are gone; the child's JSON line is kept in a compact form.  A failure or a timeout costs
    its own entry only. Round 6: `seeds` -- the headline configuration itself (configs[1]) run again under seeds + 1
    and + 2, with the default line's own run (`headline`, its seed_summary) beside them and min / median / max over the three: a launch
    lasts as long as its slowest chain, and whether a chain of this posterior ends its warm-up in trees of eight or nine doublings is a
    property of (kernel, seed, summation order)."""
    import subprocess
    out, t_all = {}, time.perf_counter()
    plan = [("configs[0]", 0, [], 120), ("configs[3]", 3, [], 120)]
    plan += [(f"seed_{seed + k}", 1, ["--seed", str(seed + k), "--warmup", "1"], 90) for k in (1, 2)]
    plan += [("configs[4]_preset", 4, [], 240), ("configs[4]_pooled", 4, ["--pooled-metric"], 150),
             ("configs[4]_pooled_f32", 4, ["--pooled-metric", "--metric-storage", "f32"], 150)]
    seed_runs = [headline] if headline else []
    for key, cfg, extra, tmo in plan:
        left = budget_s - (time.perf_counter() - t_all)
        if left < 20:
            out[key] = {"skipped": "time budget of the side measurements spent"}
            continue
        t0 = time.perf_counter()
        try:
            r = subprocess.run([sys.executable, str(Path(__file__).resolve()), "--config",
                    str(cfg)] + ([] if "--seed" in extra else ["--seed", str(seed)]) +
                               ["--no-cpu-baseline", "--no-saturated", "--no-side"] + extra,
                               capture_output=True, text=True, timeout=min(tmo, left), env={k: v for k,
                                       v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or len(lines) != 1:
                out[key] = {"error": (r.stderr or r.stdout)[-300:], "rc": r.returncode}
                continue
            d = json.loads(lines[0])
            if key.startswith("seed_"):
                d["seed"] = int(extra[extra.index("--seed") + 1])
                seed_runs.append(seed_summary(d))
                continue
            e = {"baseline_config_index": cfg, "command": f"bench.py --config {cfg}", "value": d["value"], "unit": d["unit"],
                    "metric": d["metric"], "steps": d["steps"],
                 "ms_per_step": d["ms_per_step"], "workload": d["config"]["workload"],
                         "wall_seconds_with_process_start": time.perf_counter() - t0}
            for k in ("seconds", "leapfrogs", "ess_bulk_min", "ess_per_sec", "rhat_max", "us_per_leapfrog_per_chain"):
                if k in d:
                    e[k] = d[k]
            if "roofline" in d:
                e["roofline"] = {k: d["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel",
                        "launch_ms_total", "leapfrogs_in_launches",
                                                                "matrix_passes", "avg_pass_ms", "matrix_bytes_streamed", "metric_storage",
                                                                        "pooled_metric", "mfma") if k in d["roofline"]}
            if "runs" in d:     # configs[0]: the reference's scripted calls, one by one
                gpu_keys = ("seconds", "leapfrogs", "leapfrogs_per_sec", "ess_bulk_min", "ess_per_sec", "divergent_transitions",
                            "cus_per_chain", "clusters_per_chain")
                e["runs"] = [{"call": r_["call"], "posterior": r_["posterior"], "variant": r_["variant"], "chains": r_["chains"],
                              "iter_warmup": r_["iter_warmup"], "iter_sampling": r_["iter_sampling"], **{k: r_["gpu"][k] for k in gpu_keys}}
                             for r_ in d["runs"]]
            if "dense" in d:
                e["dense"] = {k: d["dense"][k] for k in ("window_ends", "window_end_seconds", "cholesky_seconds", "cholesky_tflops",
                        "adapted_phase") if k in d["dense"]}
                e["max_depth"] = d["config"].get("max_depth")
            if "posteriors" in d["config"]:
                e["posteriors"] = {n: {k: v[k] for k in ("chains_per_gpu", "D", "cus_per_chain", "clusters_per_chain",
                        "divergent_transitions", "ess_bulk_min", "rhat_max") if k in v}
                                   for n, v in d["config"]["posteriors"].items()}
            out[key] = e
        except subprocess.TimeoutExpired:
            out[key] = {"error": f"timed out after {time.perf_counter() - t0:.0f} s"}
        except Exception as ex:                        # never let a side measurement spoil the bench line
            out[key] = {"error": str(ex)[:200]}
    if seed_runs:
        def mmm(k):
            v = sorted(r[k] for r in seed_runs if r.get(k) is not None)
            return {"min": v[0], "median": v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2]),
                    "max": v[-1]} if v else None
        deepest = [max(r["treedepth_max_per_chain"]) for r in seed_runs if r.get("treedepth_max_per_chain")]
        out["seeds"] = {"runs": seed_runs, "leapfrogs_per_sec": mmm("leapfrogs_per_sec"), "seconds": mmm("seconds"),
                "ess_per_sec": mmm("ess_per_sec"),
                        "deepest_tree_by_seed": deepest,
                        "note": "configs[1] (8 chains x (1000 + 1000), the default line's command) under the default seed and the next "
                                "two; the first run is the "
                                "default line itself.  A launch lasts as long as its slowest chain: a chain that adapts into "
                                        "trees one doubling deeper "
                                "than the others sets the time of every sampling launch"}
    # the default line's figures first, so that a reader of `side` sees the spread before anything else
    out = {**({"seeds": out.pop("seeds")} if "seeds" in out else {}), **out}
    out["note"] = ("GPU-only runs of `bench.py --config X` in child processes after the default line's timed region: same box, same clock; "
                   "configs[0] = the reference's scripted sampler calls (final_2016.R:533-541 and its 2012 / 2008 siblings), "
                           "configs[3] = the three backtests "
                   "concurrently on this one GPU (4 chains each), configs[4]_preset = the dense-metric stress shape in its "
                           "driver-runnable preset, "
                   "configs[4]_pooled = the same preset with potus_opts.pooled_metric (one inverse metric per GPU: a declared "
                           "deviation from Stan), configs[4]_pooled_f32 = that one matrix kept rounded to fp32 as well "
                                   "(metric_storage = f32: "
                           "half the bytes per pass, fp64 arithmetic)")
    out["seconds"] = time.perf_counter() - t_all
    return out


# ------------------------------------------------------------------------------------------------ the R-facing multi-GPU path
def single_process_side(args, n):
    """`bench.py --gpus n --single-process` as a child process of rank 0 (the other ranks wait at a barrier with their samplers closed):
    compact form of its line."""
    import subprocess
    t0 = time.perf_counter()
    try:
        cmd = [sys.executable, str(Path(__file__).resolve()), "--gpus", str(n), "--single-process", "--seed", str(args.seed), "--steps",
               str(args.steps), "--warmup", "1", "--chunk", str(args.chunk), "--chains-per-gpu", str(args.chains_per_gpu),
               "--cus-per-chain", str(args.cus_per_chain), "--twin", str(args.twin), "--warm-steps", str(args.warm_steps), "--launch-steps",
               str(args.launch_steps)]
        cmd += ["--max-depth", str(args.max_depth)] if args.max_depth is not None else []
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=420,
                           env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR",
                                   "MASTER_PORT", "LOCAL_WORLD_SIZE", "GROUP_RANK")})
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or len(lines) != 1:
            return {"error": (r.stderr or r.stdout)[-300:], "rc": r.returncode}
        d = json.loads(lines[0])
        return {"launcher": d["launcher"], "command": f"bench.py --gpus {n} --single-process", "value": d["value"], "unit": d["unit"],
                "n_gpus": d["n_gpus"], "seconds": d["seconds"],
                "leapfrogs": d["leapfrogs"], "ess_bulk_min": d["ess_bulk_min"], "ess_per_sec": d["ess_per_sec"], "rhat_max": d["rhat_max"],
                "diagnostics_seconds": d["diagnostics_seconds"], "wall_seconds_with_process_start": time.perf_counter() - t0,
                "note": "ONE process, one handle per GPU under potus_run_many, pooled R-hat / ESS through potus_diagnostics (peer "
                        "copies): what the R shim's "
                        "potus_sample(gpus = ...) does; same chains, same draws as the multi-process line above"}
    except subprocess.TimeoutExpired:
        return {"error": f"timed out after {time.perf_counter() - t0:.0f} s"}
    except Exception as ex:                                # never let a side measurement spoil the line
        return {"error": str(ex)[:200]}



def single_process_line(args):
    """`bench.py --gpus N --single-process`: what potus_sample(gpus = 0:(N-1)) does from R's one process (R/potus_sampling.R;
    final_2016.R:536 is the reference's only parallelism, `parallel_chains`) -- N handles of 8 chains on N devices under potus_run_many,
    chain ids by block as in the multi-process run (same draws), the pooled chains' R-hat / bulk ESS of lp__, mu_b[:, T] and
    predicted_score[T, :] through potus_diagnostics on the first handle's GPU. POTUS_BENCH_DEVICES="0,0" (development / tests on a one-GPU
    box) names the device of every handle."""
    from us_potus_model_amd import Handle, device_diagnostics, run_many
    n = args.gpus
    devs = [int(x) for x in os.environ["POTUS_BENCH_DEVICES"].split(",")] if os.environ.get("POTUS_BENCH_DEVICES") else list(range(n))
    if len(devs) != n:
        raise SystemExit(f"--single-process: POTUS_BENCH_DEVICES names {len(devs)} devices, --gpus {n}")
    if not os.environ.get("POTUS_BENCH_DEVICES") and visible_gpus() < n:
        raise SystemExit(f"--gpus {n} --single-process: {visible_gpus()} HIP device(s) visible")
    name, data, variant, C, _ = load_workloads(1, args.chains_per_gpu)[0]
    steps = 20 if args.steps is None else args.steps
    warmup = 2 if args.warmup is None else args.warmup
    chunk = args.chunk or 100
    args.launch_steps = max(args.launch_steps, 0)
    warm_steps = steps // 2 if args.warm_steps < 0 else min(args.warm_steps, steps)
    nw, ns = warm_steps * chunk, (steps - warm_steps) * chunk
    md = 10 if args.max_depth is None else args.max_depth

    def make(seed, num_warmup, num_samples):
        return [Handle(data, variant, chains=C, chain_id_offset=r * C, num_warmup=num_warmup, num_samples=num_samples, seed=seed, device=d,
                       cus_per_chain=args.cus_per_chain, max_depth=md, twin=args.twin) for r, d in enumerate(devs)]

    if warmup > 0:
        hw = make(args.seed + 1, warmup * chunk, 0)
        for h in hw:
            h.init()
        for _ in range(warmup):
            run_many(hw, chunk)
        for h in hw:
            h.close()
    hs = make(args.seed, nw, ns)
    t0 = time.perf_counter()
    for h in hs:
        h.init()
    kernel_ms, t_warm_end, done = 0.0, None, 0
    for g in launch_groups(steps, warm_steps, args.launch_steps):
        run_many(hs, chunk * g)
        kernel_ms += max(h.last_run_timing()[0] for h in hs)
        done += g
        if done == warm_steps:
            t_warm_end = time.perf_counter()
    ess, rhat, diag_s = None, None, 0.0
    if ns >= 8:
        S, T = int(data["S"]), int(data["T"])
        a_mu, a_ps = hs[0].layout["mu_b"][0], hs[0].layout["predicted_score"][0]
        td0 = time.perf_counter()
        parts = [device_diagnostics(hs, 0, 1), device_diagnostics(hs, a_mu + S * (T - 1), a_mu + S * T)]
        rp, ep = device_diagnostics(hs, a_ps + (T - 1),
                a_ps + (T - 1) + T * (S - 1) + 1)             # predicted_score[T, s]: every T-th column
        parts.append((rp[::T], ep[::T]))
        diag_s = time.perf_counter() - td0
        rhat, ess = float(np.nanmax(np.concatenate([p_[0] for p_ in parts]))), float(np.nanmin(np.concatenate([p_[1] for p_ in parts])))
    t1 = time.perf_counter()
    elapsed, samp = t1 - t0, t1 - (t_warm_end or t0)
    lf = [h.total_leapfrogs() for h in hs]
    bpl = algorithmic_bytes_per_leapfrog(data, variant)
    K, sides = hs[0].cus_per_chain, hs[0].clusters_per_chain
    achieved = bpl * sum(lf) / n / (kernel_ms * 1e-3) / 1e9        # per GPU: the devices run side by side
    line = {"metric": "leapfrog_steps_per_sec", "value": sum(lf) / elapsed, "unit": "leapfrogs/s", "n_gpus": n, "steps": steps,
            "warmup": warmup,
            "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "poll data of the reference (fixtures built from its CSVs, tests/golden/data_*.npz); random inits",
                    "launcher": "single_process",
            "config": {"workload": f"configs[{1 if n == 1 else 2}]: 2016 backtest, adaptive NUTS diag_e, {C} chains per MI355X x {n}, "
                                   f"{nw} warmup + {ns} sampling, seed {args.seed}; "
                                   f"ONE process, {n} handles under potus_run_many (the R-facing path: potus_sample(gpus = "
                                           f"...)); a step = {chunk} transitions, one potus_run_many call per phase unless "
                                                   f"--launch-steps says otherwise",
                       "baseline_config_index": 1 if n == 1 else 2, "chains_per_gpu": C, "total_chains": C * n, "devices": devs,
                               "iter_warmup": nw, "iter_sampling": ns,
                       "parallelism": f"one host process, {n} handles of {C} chains, one per device; pooled R-hat / ESS through "
                                      f"potus_diagnostics (peer copies), no RCCL",
                       "cus_per_chain": K, "clusters_per_chain": sides},
            "leapfrogs": int(sum(lf)), "seconds": elapsed, "sampling_seconds": samp, "ess_bulk_min": ess,
                    "ess_per_sec": (ess / samp) if ess else None, "rhat_max": rhat,
            "diagnostics_seconds": diag_s,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": None,
                         "kernel": "k_cl_run" if K > 1 else "k_run", "algorithmic_bytes_per_leapfrog": bpl,
                                 "leapfrogs_in_launches": int(sum(lf)), "launch_ms_total": kernel_ms,
                         "note": "per GPU: leapfrogs of all handles / n_gpus x the algorithmic bytes / the launches' time (the handles "
                                 "of a step run side by side)"}}
    for h in hs:
        h.close()
    return line


# ------------------------------------------------------------------------------------------------ CPU baseline
def _cpu_nuts_worker(args):
    data, variant, chain, nw, ns, seed, budget = args
    from oracle_lib import OracleModel
    m = OracleModel(data, variant)
    o = m.default_opts(num_warmup=nw, num_samples=ns, seed=seed, fast_grad=1, pooled=1)
    draws, _, timing = m.sample_chain_timed(chain, o, budget_s=budget)
    if budget > 0:
        return chain, timing, None
    # the columns of the ESS definition: lp__, mu_b[:, T], predicted_score[T, :]
    from us_potus_model_amd import _abi
    lay, _ = _abi.column_layout(data, variant)
    S, T = int(data["S"]), int(data["T"])
    a = lay["mu_b"][0] - _abi.N_SAMPLER_COLS + S * (T - 1)
    mu = np.stack([m.write_array(q)[a:a + S] for q in draws[:, _abi.N_SAMPLER_COLS:]])
    return chain, timing, np.concatenate([draws[:, :1], mu, 1.0 / (1.0 + np.exp(-mu))], axis=1)


def _cpu_loop_worker(args):
    data, variant, fast, budget = args
    from oracle_lib import OracleModel
    m = OracleModel(data, variant)
    n, t, batch = 0, 0.0, 200
    while t < budget:
        t += m.time_leapfrogs(batch, eps=0.01, fast=fast, seed=1 + n)
        n += batch
    return n, t


def ess_min(cols):
    """min bulk-ESS over the columns of [chain, draw, column] (lp__, mu_b[:, T], predicted_score[T, :])."""
    from us_potus_model_amd import diagnostics as dg
    return float(min(dg.ess_bulk(cols[:, :, j]) for j in range(cols.shape[2])))


def cpu_baseline(data, variant, chains, seed, nw, ns, short, budget=12.0, loop_budget=3.0):
    """the CPU port (oracle/potus_oracle.c, kind "port": rstan / CmdStan cannot run here) on the box's own host
    cores, one chain per core as the reference runs its chains (final_2016.R:536), in its fastest form -- the scan/sparse
    gradient (the GPU's algebra) under the pooled-buffer NUTS (oracle_opts.pooled: same draws as the literal recursion, bit for
    bit).  Three measurements:
      value           the first `budget` seconds of the SAME run as the GPU's (8 x (nw + ns), same seed and chain ids): leapfrogs / s;
      short_config    a COMPLETE short configuration (8 x (150 + 100), same seed) run to the end: leapfrogs / s and the MEASURED
                      ESS / s of its sampling phase -- main() runs the same configuration on the GPU and puts the two side by side;
      leapfrog_loop*  plain leapfrog loops without any tree bookkeeping: the rate of the gradient arithmetic alone, an upper bound
                      for any sampler on these cores."""
    procs = max(1, min(chains, os.cpu_count() or 1))
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(procs) as pool:
        res = pool.map(_cpu_nuts_worker, [(data, variant, c + 1, nw, ns, seed, budget) for c in range(procs)])
        loop = {}
        for key, fast in (("leapfrog_loop_value", 0), ("leapfrog_loop_scan_sparse_value", 1)):
            r = pool.map(_cpu_loop_worker, [(data, variant, fast, loop_budget)] * procs)
            loop[key] = float(sum(n / t for n, t in r))
        snw, sns = short
        ts0 = time.perf_counter()
        sres = sorted(pool.map(_cpu_nuts_worker, [(data, variant, c + 1, snw, sns, seed, 0.0) for c in range(procs)]), key=lambda r: r[0])
        short_wall = time.perf_counter() - ts0
    wall = time.perf_counter() - t0
    timing = np.stack([r[1] for r in sorted(res, key=lambda r: r[0])])      # [chain][warm s, samp s, warm lf, samp lf, iterations]
    rate = float(sum((tm[2] + tm[3]) / (tm[0] + tm[1]) for tm in timing))    # every chain on its own core, at its own rate
    st = np.stack([r[1] for r in sres])
    cols = np.stack([r[2] for r in sres])                                    # [chain, draw, 1 + 2 S]
    s_ess = ess_min(cols)
    s_secs, s_samp = float((st[:, 0] + st[:, 1]).max()), float(st[:, 1].max())   # the chains run side by side: the slowest sets the time
    out = dict(value=rate, unit="leapfrogs/s", cores=procs, host_cores_total=os.cpu_count(), kind="port",
            leapfrogs_per_sec_per_core=rate / procs,
               value_is=f"the rate of the FIRST {int(timing[:, 4].min())}-{int(timing[:, 4].max())} iterations (early warm-up: the "
                        f"longest trees) of the configured "
                        f"{nw} + {ns}, {budget:.0f} s per chain -- a bounded sample, not a complete run; the like-for-like pair, "
                                f"both sides run to the end, is short_config",
               iterations_sampled=[int(timing[:, 4].min()), int(timing[:, 4].max())], iterations_configured=nw + ns,
               sample=f"the first {int(timing[:, 4].min())}-{int(timing[:, 4].max())} iterations of the same run ({procs} chains, ids "
                       f"1..{procs}, seed {seed}, "
                      f"{nw} warm-up + {ns} sampling configured) on {procs} host processes, cut after {budget:.0f} s each: "
                      f"{int(timing[:, 2:4].sum())} leapfrogs; scan/sparse gradient, pooled-buffer tree (the fastest form of the port)",
               cores_note=f"one process per chain, as the reference runs its chains (final_2016.R:536): {procs} of the box's "
                          f"{os.cpu_count()} host cores are used",
               seconds=wall, **loop,
               leapfrog_loop_note="plain leapfrog loops (unit metric, eps 0.01, no tree, no U-turn bookkeeping) of the literal stan:86 "
                                  "recursion and of the "
                                  f"scan/sparse gradient, {procs} processes x {loop_budget:.0f} s each: the rate of the arithmetic alone",
               short_config=dict(iter_warmup=snw, iter_sampling=sns, chains=procs, seed=seed, leapfrogs=int(st[:, 2:4].sum()),
                       seconds=s_secs,
                                 sampling_seconds=s_samp, leapfrogs_per_sec=float(st[:, 2:4].sum()) / s_secs, ess_bulk_min=s_ess,
                                 ess_per_sec=s_ess / s_samp, wall_seconds=short_wall,
                                 note="a complete run of the port, measured: min bulk-ESS over lp__, mu_b[:, T], predicted_score[T, "
                                         ":] of its own "
                                      "draws / the sampling time of its slowest chain"))
    full = ROOT / "tests" / "golden" / "posterior_2016.npz"
    if full.exists() and int(data["T"]) == 254:
        g = np.load(full)
        lf, sec = float(g["leapfrogs"].sum()), float(g["seconds"].max())
        ess = float(min(g["lp__ess_bulk"].min(), g["mu_b_T__ess_bulk"].min(), g["predicted_score_T__ess_bulk"].min()))
        out["full_run_in_build_container"] = dict(leapfrogs_per_sec=lf / sec, ess_bulk_min=ess, ess_per_sec_total_time=ess / sec,
                seconds=sec,
                                                  note="the whole 8 x (1000 + 1000) run of the oracle (scan/sparse gradient, "
                                                          "recursive tree) behind "
                                                       "tests/golden/posterior_2016.npz, 8 processes on the build container's 8 "
                                                               "vCPUs -- not this box")
    return out


# ------------------------------------------------------------------------------------------------ configs[0]
def reference_sampler_calls(local, seed, cus_per_chain, twin, max_depth, cpu=True):
    """BASELINE configs[0]: the reference's own sampler calls as they are scripted, each run to completion on the GPU and on the CPU
    port, side by side -- scripts/model/final_2016.R:6-11,533-541 (6 chains x (500 + 500), seed 1843, refresh 50), the same call of
    final_2012.R:558-569 and final_2008.R:562-573 (no_mode_adjustment model), and BASELINE's own wording of configs[0] (4 chains x 500
    iterations "via rstan": rstan's iter includes the warm-up, final_2016.R:525-529 -> 250 + 250).  The GPU advances in chunks of
    `refresh` transitions, as the R loop over potus_run would (R/potus_sampling.R)."""
    import torch
    from us_potus_model_amd import Handle, dataprep
    gold = ROOT / "tests" / "golden"
    calls = [("final_2016.R:533-541 as scripted", "2016", "full", 6, 500, 500),
            ("BASELINE configs[0]: 4 chains x 500 iterations (rstan: 250 + 250)", "2016", "full", 4, 250, 250),
             ("final_2012.R:558-569 as scripted", "2012", "no_mode_adjustment", 6, 500, 500), ("final_2008.R:562-573 as scripted", "2008",
                     "no_mode_adjustment", 6, 500, 500)]
    out = []
    for label, year, variant, chains, nw, ns in calls:
        data = dataprep.load_npz(gold / f"data_{year}.npz")["data"]
        S, T = int(data["S"]), int(data["T"])
        refresh = max(ns // 10, 1)
        hw = Handle(data, variant, chains=chains, num_warmup=refresh, num_samples=0, seed=seed + 1, device=local,
                cus_per_chain=cus_per_chain, twin=twin, max_depth=max_depth)
        hw.init(); hw.run(refresh); hw.close()                       # untimed: code objects, clocks
        h = Handle(data, variant, chains=chains, num_warmup=nw, num_samples=ns, seed=seed, device=local, cus_per_chain=cus_per_chain,
                twin=twin, max_depth=max_depth)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        h.init()
        for _ in range(nw // refresh):
            h.run(refresh)
        t1 = time.perf_counter()
        for _ in range(ns // refresh):
            h.run(refresh)
        t2 = time.perf_counter()
        a_mu = h.layout["mu_b"][0]
        mu = np.transpose(h.write_array(a_mu + S * (T - 1), a_mu + S * T, ns), (1, 0, 2))
        lpc = np.transpose(h.write_array(0, 1, ns), (1, 0, 2))
        cols = np.concatenate([lpc, mu, 1.0 / (1.0 + np.exp(-mu))], axis=2)
        g_ess, g_lf = ess_min(cols), h.total_leapfrogs()
        st, dv = h.chain_status()
        run = {"call": label, "posterior": year, "variant": variant, "chains": chains, "iter_warmup": nw, "iter_sampling": ns,
                "seed": seed, "D": h.D,
               "gpu": {"seconds": t2 - t0, "sampling_seconds": t2 - t1, "leapfrogs": g_lf, "leapfrogs_per_sec": g_lf / (t2 - t0),
                       "ess_bulk_min": g_ess,
                       "ess_per_sec": g_ess / (t2 - t1), "divergent_transitions": int(sum(dv)), "cus_per_chain": h.cus_per_chain,
                       "clusters_per_chain": h.clusters_per_chain, "mean_predicted_score_T": cols[:, :, 1 + S:].mean(axis=(0, 1)).tolist()}}
        h.close()
        if cpu:
            procs = max(1, min(chains, os.cpu_count() or 1))
            tc0 = time.perf_counter()
            with mp.get_context("spawn").Pool(procs) as pool:
                res = sorted(pool.map(_cpu_nuts_worker, [(data, variant, c + 1, nw, ns, seed, 0.0) for c in range(chains)]),
                        key=lambda r: r[0])
            wall = time.perf_counter() - tc0
            tm = np.stack([r[1] for r in res])
            ccols = np.stack([r[2] for r in res])
            c_ess = ess_min(ccols)
            c_secs, c_samp = float((tm[:, 0] + tm[:, 1]).max()), float(tm[:, 1].max())
            run["cpu_port"] = {"seconds": c_secs, "sampling_seconds": c_samp, "wall_seconds_with_process_start": wall,
                    "leapfrogs": int(tm[:, 2:4].sum()),
                               "leapfrogs_per_sec": float(tm[:, 2:4].sum()) / c_secs, "ess_bulk_min": c_ess, "ess_per_sec": c_ess / c_samp,
                                       "cores": procs,
                               "kind": "port", "mean_predicted_score_T": ccols[:, :, 1 + S:].mean(axis=(0, 1)).tolist(),
                               "note": "oracle/potus_oracle.c, scan/sparse gradient, pooled-buffer tree, one chain per host process: the "
                                       "same run (seed, chain ids) to completion"}
            score_gpu, score_cpu = np.array(run["gpu"]["mean_predicted_score_T"]), np.array(run["cpu_port"]["mean_predicted_score_T"])
            run["gpu_over_cpu"] = {"wall": c_secs / (t2 - t0),
                                   "leapfrogs_per_sec": run["gpu"]["leapfrogs_per_sec"] / run["cpu_port"]["leapfrogs_per_sec"],
                                   "ess_per_sec": run["gpu"]["ess_per_sec"] / run["cpu_port"]["ess_per_sec"],
                                   "max_abs_diff_of_mean_predicted_score_T": float(np.abs(score_gpu - score_cpu).max())}
        out.append(run)
    return out


# ------------------------------------------------------------------------------------------------ workloads
def load_workloads(cfg, chains_per_gpu, twin_posteriors="", storage=0, pooled=0):
    """[(name, data, variant, chains per GPU, options)] of the posteriors one GPU runs."""
    from us_potus_model_amd import dataprep, synthetic
    gold = ROOT / "tests" / "golden"
    if cfg in (1, 2):
        return [("2016", dataprep.load_npz(gold / "data_2016.npz")["data"], "full", chains_per_gpu or 8, {})]
    if cfg == 3:
        c = chains_per_gpu or 4
        # three fits side by side: 3 x 4 x 16 = 192 compute units; the 64 left over give ONE of them a second cluster per chain
        tw = set(twin_posteriors.split(",")) if twin_posteriors else set()
        return [(y, dataprep.load_npz(gold / f"data_{y}.npz")["data"], v, c, {"twin": 1 if y in tw else 0})
                for y, v in (("2008", "no_mode_adjustment"), ("2012", "no_mode_adjustment"), ("2016", "full"))]
    if cfg == 4:
        from us_potus_model_amd import _abi
        return [("stress", synthetic.stress(), "full", chains_per_gpu or 16, {"metric": _abi.METRIC_DENSE, "metric_storage": storage,
                "pooled_metric": pooled})]
    raise SystemExit(f"unknown --config {cfg}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
            help="launch chunks of --chunk transitions: K // 2 warm-up, the rest sampling (default 20; --config 4: 5)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed chunks on a throw-away sampler (default 2; --config 4: 0)")
    ap.add_argument("--config", type=int, default=1, help="BASELINE.json configs[] index: 1 (default), 0, 2, 3, 4")
    ap.add_argument("--chains-per-gpu", type=int, default=0, help="per posterior (0 = the configuration's: 8, 8, 4, 4)")
    ap.add_argument("--cus-per-chain", type=int, default=0,
            help="workgroups per chain (0 = the library's choice: 16, 10-14, 8, 4 or 1 by what fits)")
    ap.add_argument("--twin", type=int, default=-1,
            help="two clusters per chain, one per end of the trajectory: 1, 0, or -1 = the library's choice")
    ap.add_argument("--twin-posteriors", default="2016",
            help="--config 3: the posteriors (comma-separated years) that get two clusters per chain")
    ap.add_argument("--chunk", type=int, default=0, help="transitions per step (0 = the configuration's: 100; 1 for --config 4)")
    ap.add_argument("--warm-steps", type=int, default=-1, help="how many of the --steps are warm-up (-1 = half of them)")
    ap.add_argument("--launch-steps", type=int, default=-1,
                    help="steps per potus_run call = per kernel launch: 0 = one launch per phase (default; --config 4: 1, the run is "
                            "reported step by step)")
    ap.add_argument("--metric-storage", default="f64", choices=["f64", "f32"], help="--config 4: storage of the dense inverse metric")
    ap.add_argument("--pooled-metric", action="store_true",
            help="--config 4: ONE dense inverse metric per GPU, adapted from the window draws of all its chains "
                                                                 "(potus_opts.pooled_metric; a declared deviation from Stan)")
    ap.add_argument("--max-depth", type=int, default=None, help="default 10 (CmdStan's); --config 4: 7, stated in the line")
    ap.add_argument("--adapt-windows", default="",
            help="init_buffer,window,term_buffer of the warm-up (default: CmdStan's 75,25,50, rescaled by windowed_adaptation for "
                    "short warm-ups); "
                                                        "e.g. 6,8,6 puts two window ends into a 40-iteration warm-up of --config 4")
    ap.add_argument("--gather", default="full", choices=["full", "T"],
            help="what the all-gather pools: lp__ + all of mu_b (SURVEY 8e) or lp__ + mu_b[:, T] only")
    ap.add_argument("--seed", type=int, default=1843)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-saturated", action="store_true", help="skip the 256-chain reference point")
    ap.add_argument("--single-process", action="store_true",
            help="ONE process, --gpus N handles on N devices under potus_run_many: the R-facing multi-GPU path")
    ap.add_argument("--no-side", action="store_true",
            help="skip the side measurements of configs[0], configs[3] and the configs[4] preset (default line only)")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # before the first HIP call (module docstring)
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if args.single_process:
        if args.config not in (1, 2):
            raise SystemExit("--single-process runs configs[1] / configs[2]")
        print(json.dumps(single_process_line(args)), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))

    import torch
    from us_potus_model_amd import Handle, diagnostics as dg, parallel, run_many

    # POTUS_DIST_BACKEND=gloo: development aid -- every rank on GPU 0 and CPU tensors in the collectives, to exercise
    # the N > 1 flow on a one-GPU box (the measured configuration is one rank per GPU over RCCL)
    dev_backend = os.environ.get("POTUS_DIST_BACKEND")
    rank, world, local = parallel.init_process_group(dev_backend)
    # (the single-process side run needs the GPUs to itself: not in the development mode, where the ranks share GPU 0 -- unless a test asks
    # for it)
    sp_side = dev_backend != "gloo" or bool(os.environ.get("POTUS_BENCH_FORCE_SP_SIDE"))
    wait_for_rank0 = None
    # the job's key-value store (TCP, CPU side): how the ranks wait for rank 0's side measurement without a GPU kernel
    if world > 1:
        import torch.distributed as dist
        wait_for_rank0 = dist.distributed_c10d._get_default_store()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    if dev_backend == "gloo":
        local = 0
    elif torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {args.gpus}: {torch.cuda.device_count()} HIP device(s) visible, one rank per GPU needs {world}")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    coll_dev = None if dev_backend == "gloo" else dev       # collectives on CPU tensors in the development mode
    twin = 0 if dev_backend == "gloo" else args.twin        # (the ranks of the development mode share one GPU's compute units)

    cfg = args.config
    # --config 4 without further flags is a preset the driver can run in a few minutes: 16 chains of the stress shape on the GPU, dense
    # metric, 20 warm-up iterations (init buffer 3, one window of 15 draws whose end -- covariance, 16 Cholesky factorisations of 13.85 GB
    # matrices, init_stepsize -- falls into step 4 of 5) + 5 sampling iterations, trees cut at depth 7 (127 leapfrogs; CmdStan's 10
    # would make the early warm-up iterations ten times longer), chunks of 5 transitions
    preset4 = cfg == 4 and args.steps is None and args.chunk == 0
    if args.steps is None:
        args.steps = 5 if cfg == 4 else 20
    if args.warmup is None:
        args.warmup = 0 if cfg == 4 else 2
    if args.max_depth is None:
        args.max_depth = 7 if cfg == 4 else 10
    if preset4:
        args.chunk, args.warm_steps = 5, 4
    if args.launch_steps < 0:
        args.launch_steps = 1 if cfg == 4 else 0
    if cfg == 0:
        if world != 1:
            raise SystemExit("--config 0 (the reference's own sampler calls, GPU and CPU port side by side) runs on one GPU")
        runs = reference_sampler_calls(local, args.seed, args.cus_per_chain, args.twin, args.max_depth, cpu=not args.no_cpu_baseline)
        r0 = runs[0]
        print(json.dumps({"metric": "leapfrog_steps_per_sec", "value": r0["gpu"]["leapfrogs_per_sec"], "unit": "leapfrogs/s", "n_gpus": 1,
                "steps": 20, "warmup": 1,
                          "ms_per_step": 1e3 * r0["gpu"]["seconds"] / 20, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                                  "dtype": "f64",
                          "data": "poll data of the reference (fixtures built from its CSVs, tests/golden/data_*.npz); random inits",
                          "config": {"workload": "configs[0]: the reference's sampler calls as scripted (final_2016.R:533-541: 6 chains "
                                                 "x (500 + 500), seed 1843; final_2012.R, "
                                                 "final_2008.R likewise; BASELINE's 4 x 500), each run to completion on the "
                                                         "MI355X and on the CPU port; value = the 2016 call on the GPU; "
                                                 "a step = `refresh` = 50 transitions"},
                          "runs": runs}), flush=True)
        parallel.barrier()
        return
    chunk = args.chunk or (1 if cfg == 4 else 100)
    if args.pooled_metric and cfg != 4:
        raise SystemExit("--pooled-metric applies to --config 4")
    # (N > 1: every window end pooled over the ranks as well -- pooled_metric = 2, sampler.run_pooled: two small all-reduces and one of D x
    # D doubles per window end)
    pooled_mode = 0 if not args.pooled_metric else (2 if world > 1 else 1)
    work = load_workloads(cfg, args.chains_per_gpu, "" if dev_backend == "gloo" else args.twin_posteriors,
            1 if args.metric_storage == "f32" else 0, pooled_mode)
    warm_steps = args.steps // 2 if args.warm_steps < 0 else min(args.warm_steps, args.steps)
    nw, ns = warm_steps * chunk, (args.steps - warm_steps) * chunk

    def make(seed, num_warmup, num_samples):
        hs = []
        for i, (name, data, variant, C, extra) in enumerate(work):
            if args.adapt_windows:
                ib_, bw_, tb_ = (int(x) for x in args.adapt_windows.split(","))
                extra = {**extra, "init_buffer": ib_, "window": bw_, "term_buffer": tb_}
            while True:
                try:
                    hs.append(Handle(data, variant, chains=C, chain_id_offset=rank * C, num_warmup=num_warmup, num_samples=num_samples,
                                     seed=seed, device=local, cus_per_chain=args.cus_per_chain, max_depth=args.max_depth, **{"twin": twin,
                                             **extra}))
                    break
                except Exception as e:      # the dense metric keeps two D x D matrices per chain: as many chains as the HBM holds
                    if cfg != 4 or args.chains_per_gpu or C <= 1 or "GB free" not in str(e):
                        raise
                    C -= 1
            work[i] = (name, data, variant, C, extra)
        return hs

    if args.warmup > 0:  # untimed: throw-away samplers
        hw = make(args.seed + 1, args.warmup * chunk, 0)
        for h in hw:
            h.init()
        for _ in range(args.warmup):
            run_many(hw, chunk)
        for h in hw:
            h.close()

    hs = make(args.seed, nw, ns)
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for h in hs:
        h.init()
    kernel_ms, t_warm_end, lf_warm = 0.0, None, 0
    per_step = []                                                   # --config 4: the run chunk by chunk (where the window ends fall)
    groups, done = launch_groups(args.steps, warm_steps, args.launch_steps), 0
    for g in groups:
        ts0, lf0 = time.perf_counter(), sum(h.total_leapfrogs() for h in hs)
        dt0 = hs[0].dense_timing() if cfg == 4 else None
        if pooled_mode == 2:
            from us_potus_model_amd import sampler as _sampler
            _sampler.run_pooled(hs, chunk * g, coll_dev)
        else:
            run_many(hs, chunk * g)
        kernel_ms += max(h.last_run_timing()[0] for h in hs)        # the handles of a launch run concurrently
        if cfg == 4:
            dt1, at = hs[0].dense_timing(), hs[0].dense_adapt_timing()
            per_step.append({"iterations": [done * chunk, (done + g) * chunk], "seconds": time.perf_counter() - ts0,
                             "leapfrogs": sum(h.total_leapfrogs() for h in hs) - lf0, "matrix_pass_ms": dt1[0] - dt0[0],
                             "matrix_bytes": dt1[2] - dt0[2], "window_ends_so_far": at["window_ends"]})
        done += g
        if done == warm_steps:
            torch.cuda.synchronize()
            t_warm_end = time.perf_counter()
            lf_warm = sum(h.total_leapfrogs() for h in hs)
    # The one exchange of the path: pool the draws-of-interest of every chain of every GPU -- lp__ and the whole of
    # mu_b (S x T per draw: predicted_score is its inverse logit, column for column, and is not sent a second time), 12 955
    # doubles per draw for 2016 = 0.83 GB per rank with 8 chains x 1000 draws.  They are produced on the device
    # (potus_write_array_device), gathered on the device (RCCL all-gather over xGMI) and stay there; only the 1 + S columns
    # R-hat / ESS are taken on (lp__, mu_b[:, T]) come to the host, on rank 0's behalf.  --gather T sends those columns only.
    pooled, gathered_bytes, dev_diag = [], 0, []
    for (name, data, variant, C, _), h in zip(work, hs):
        S, T = int(data["S"]), int(data["T"])
        a_mu = h.layout["mu_b"][0]
        if ns == 0:
            pooled.append(None)
            dev_diag.append(None)
            continue
        ncol = S * T if args.gather == "full" else S
        loc = torch.empty((ns, C, 1 + ncol), dtype=torch.float64, device=dev)
        tmp = torch.empty((ns, C, 1), dtype=torch.float64, device=dev)
        h.write_array_device(0, 1, tmp)
        loc[:, :, :1] = tmp
        tmp = torch.empty((ns, C, ncol), dtype=torch.float64, device=dev)
        h.write_array_device(a_mu + (0 if args.gather == "full" else S * (T - 1)), a_mu + S * T, tmp)
        loc[:, :, 1:] = tmp
        del tmp
        full = parallel.all_gather_chains(loc, coll_dev)                # [ns, world * C, 1 + ncol] on every rank
        gathered_bytes += loc.numel() * 8
        # what the gather is for: rank-normalised split R-hat and bulk ESS of EVERY gathered column over the pooled chains, on the
        # device (potus_diagnostics_device, csrc/potus_diag.hpp), inside the timed region
        td0 = time.perf_counter()
        if ns >= 8 and full.is_cuda:
            from us_potus_model_amd import device_diagnostics_of_block
            # + predicted_score[T, :] = inv_logit(mu_b[:, T]) as S columns of their own (stan:137): a monotone map keeps the ranks, hence
            # the bulk ESS, but not the FOLDED split R-hat (|x - median| is not invariant under it), and the metric's column set names them
            # (two calls -- the gathered block as it is, then the S sigmoid columns -- instead of one concatenated copy of the whole block:
            # ADVICE r05)
            sig = torch.sigmoid(full[:, :, 1 + ncol - S:]).contiguous()
            nf = int(full.shape[2])
            ncols_all = nf + S

            def diag_cols(ca, cb):
                parts = ([full[:, :, ca:min(cb, nf)]] if ca < nf else []) + ([sig[:, :, max(ca, nf) - nf:cb - nf]] if cb > nf else [])
                res = [device_diagnostics_of_block(p_.contiguous()) for p_ in parts if p_.shape[2] > 0]
                return (np.concatenate([r_[0] for r_ in res]), np.concatenate([r_[1] for r_ in res])) if res else (np.zeros(0), np.zeros(0))

            if world == 1:
                rh, es = diag_cols(0, ncols_all)
            else:
                # every rank holds the pooled chains; each sorts its 1/world share of the columns (per-GPU work stays what it is at
                # N = 1: world x the draws, 1/world of the columns) and the per-column results are gathered
                ca, cb = parallel.column_block(ncols_all, rank, world)
                rh_l, es_l = diag_cols(ca, cb)
                rh, es = parallel.all_gather_columns(rh_l, ncols_all, coll_dev), parallel.all_gather_columns(es_l, ncols_all, coll_dev)
            dev_diag.append({"rhat": rh, "ess_bulk": es, "seconds": time.perf_counter() - td0, "columns": ncols_all, "S": S})
            del sig
        else:
            dev_diag.append(None)
        sel = torch.cat([full[:, :, :1], full[:, :, 1 + ncol - S:]], dim=2)   # lp__ and mu_b[:, T]
        pooled.append(sel.contiguous())
        del full, loc
    torch.cuda.synchronize()
    parallel.barrier()
    t1 = time.perf_counter()

    elapsed = parallel.max_over_ranks(t1 - t0, coll_dev)
    lf_local = [h.total_leapfrogs() for h in hs]
    dense_t = hs[0].dense_timing() if cfg == 4 else None
    leapfrogs = parallel.sum_over_ranks(float(sum(lf_local)), coll_dev)
    kernel_ms_max = parallel.max_over_ranks(kernel_ms, coll_dev)
    samp_time = parallel.max_over_ranks(t1 - (t_warm_end or t0), coll_dev)
    status = [h.chain_status() for h in hs]

    if rank == 0:
        per_post = {}
        ess_all, rhat_all = [], []
        for (name, data, variant, C, extra), h, pl, (st, dv), dd in zip(work, hs, pooled, status, dev_diag):
            info = {"chains_per_gpu": C, "D": h.D, "S": int(data["S"]), "T": int(data["T"]),
                    "polls": int(data["N_state_polls"]) + int(data["N_national_polls"]), "cus_per_chain": h.cus_per_chain,
                    "clusters_per_chain": h.clusters_per_chain,
                    "divergent_transitions": int(sum(dv)), "chain_status": st}
            if ns > 0 and cfg != 4:
                sp = h.write_array(2, 4,
                        ns)                                     # stepsize__, treedepth__ of the saved (sampling) draws: [draw, chain, 2]
                info["treedepth_max_per_chain"] = [int(v) for v in sp[:, :, 1].max(axis=0)]
                info["treedepth_mean_per_chain"] = [round(float(v), 3) for v in sp[:, :, 1].mean(axis=0)]
                info["stepsize_per_chain"] = [float(v) for v in sp[-1, :, 0]]
            if h.clusters_per_chain == 2:
                cnt, rb, rf = h.twin_stats()
                info["twin"] = {"leapfrogs_counted": cnt, "leaves_run_backward_side": rb, "leaves_run_forward_side": rf,
                                "leaves_run_per_counted": (rb + rf) / max(cnt, 1),
                                "note": "each side integrates the doublings of its end, those of speculative subtrees that are dropped "
                                        "included"}
            if dd is not None:
                # The metric's ESS: min bulk-ESS (and max R-hat) over lp__, mu_b[:, T], predicted_score[T, :]: column 0 and the
                # last 2 S columns of the block the diagnostics ran on (mu_b's last day, then its inverse logit).
                Sd = dd["S"]
                sel = np.r_[0, np.arange(dd["columns"] - 2 * Sd, dd["columns"])]
                info["ess_bulk_min"] = float(np.nanmin(dd["ess_bulk"][sel]))
                info["rhat_max"] = float(np.nanmax(dd["rhat"][sel]))
                info["pooled_draws"] = int(ns * C * world)
                info["device_diagnostics"] = {"columns": dd["columns"], "seconds": dd["seconds"],
                        "ess_bulk_min_all_columns": float(np.nanmin(dd["ess_bulk"])),
                                              "ess_bulk_median_all_columns": float(np.nanmedian(dd["ess_bulk"])),
                                                      "rhat_max_all_columns": float(np.nanmax(dd["rhat"])),
                                              "note": "potus_diagnostics_device over every gathered column (lp__ + mu_b) + "
                                                      "predicted_score[T, :], pooled chains of all ranks, inside the timed region"}
                ess_all.append(info["ess_bulk_min"]); rhat_all.append(info["rhat_max"])
            # development mode (collectives on CPU tensors): the numpy restatement
            elif pl is not None and ns >= 8:
                x = np.transpose(pl.cpu().numpy(), (1, 0, 2))                    # [chain, draw, 1 + S]
                cols = np.concatenate([x, 1.0 / (1.0 + np.exp(-x[:, :, 1:]))], axis=2)   # + predicted_score[T, :]
                info["ess_bulk_min"] = float(min(dg.ess_bulk(cols[:, :, j]) for j in range(cols.shape[2])))
                info["rhat_max"] = float(max(dg.rhat(cols[:, :, j]) for j in range(cols.shape[2])))
                info["pooled_draws"] = int(cols.shape[0] * cols.shape[1])
                ess_all.append(info["ess_bulk_min"]); rhat_all.append(info["rhat_max"])
            per_post[name] = info
        dense = cfg == 4
        bpl = [algorithmic_bytes_per_leapfrog(d, v, dense) for _, d, v, _, _ in work]
        alg_bytes = float(sum(b * n for b, n in zip(bpl, lf_local)))
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9            # this rank's launches, events on the samplers' own streams
        if dense:   # the dominant kernel is the matrix pass: bytes of matrix it streamed / its own time (HIP events around every launch)
            achieved = dense_t[2] / (dense_t[0] * 1e-3) / 1e9
        K, sides = hs[0].cus_per_chain, hs[0].clusters_per_chain
        kernel = "k_dn_symv" if dense else ("k_cl_run" if K > 1 else "k_run")
        tr = measured_traffic(kernel, sides)
        traffic = tr[1]["hbm_bytes_per_leapfrog"] * sum(lf_local) / (kernel_ms * 1e-3) / 1e9 if tr else None
        traffic_note = (f"NOT measured in this run: {tr[1]['hbm_bytes_per_leapfrog']:.0f} HBM bytes per leapfrog from the committed "
                        f"rocprofv3 --pmc passes of this command (2 x FETCH_SIZE + WRITE_SIZE, profiles/{tr[0]}; FETCH_SIZE "
                                f"calibrated at 0.500 counted "
                        f"bytes per streamed byte for the sampler's 8- and 16-byte plain and sc1 loads, "
                                f"profiles/r04_fetch_size_calibration.txt; WRITE_SIZE "
                        f"at 1.000, profiles/r03_write_size_calibration.txt) x this run's leapfrogs / launch time" if tr
                        else "no PMC pass committed for this kernel")
        if dense and args.pooled_metric:
            traffic, traffic_note = None, "no counter pass committed for k_dn_pool_mm"
            for f in sorted((ROOT / "profiles").glob("*dense_pooled_pmc_fetch.json")):
                try:
                    ratio = json.loads(f.read_text())["ratio"]
                except (OSError, ValueError, KeyError):
                    continue
                traffic = achieved * ratio
                traffic_note = (f"NOT measured in this run: HBM reads of k_dn_pool_mm = {ratio:.3f} x the bytes of the matrix (rocprofv3 "
                                f"--pmc FETCH_SIZE pass of the pooled "
                                f"sampler, doubled per the guide; profiles/{f.name}) x this run's rate")
        # the matrix pass: HBM reads measured / bytes loaded by construction, from the committed counter pass of the dense sampler
        elif dense:
            for f in sorted((ROOT / "profiles").glob("*dense_pmc_fetch.json")):
                try:
                    ratio = json.loads(f.read_text())["ratio_with_finish"]
                except (OSError, ValueError, KeyError):
                    continue
                traffic = achieved * ratio
                traffic_note = (f"NOT measured in this run: HBM reads of k_dn_symv + k_dn_symv_finish = {ratio:.3f} x the bytes the "
                                f"passes load by "
                                f"construction (rocprofv3 --pmc FETCH_SIZE pass of the dense sampler, doubled per the guide; "
                                        f"profiles/{f.name}) x this run's rate")
        C_tot = sum(w[3] for w in work)
        names = {1: "configs[1]: 2016 backtest", 2: "configs[2]: 2016 backtest, chains sharded over the GPUs",
                 3: "configs[3]: 2008 + 2012 + 2016 backtests concurrently",
                 4: "configs[4]: synthetic stress posterior, dense metric" + (" POOLED over the GPU's chains (a declared deviation from "
                                                                              "Stan)" if args.pooled_metric else "")}
        line = {
            "metric": "leapfrog_steps_per_sec", "value": leapfrogs / elapsed, "unit": "leapfrogs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": ("synthetic polls (us_potus_model_amd.synthetic.stress, seed 20201103)" if cfg == 4 else
                     "poll data of the reference (fixtures built from its CSVs, tests/golden/data_*.npz)") + "; random inits",
            "config": {"workload": f"{names[2 if (cfg == 1 and world > 1) else cfg]}, adaptive NUTS {'dense_e' if dense else 'diag_e'}, "
                                   f"{C_tot} chains per MI355X, "
                                   f"{nw} warmup + {ns} sampling, seed {args.seed}; a step = {chunk} transitions of every chain, "
                                   f"the {args.steps} steps issued as {len(groups)} potus_run call(s) = kernel launch(es) per handle",
                       "step": f"{chunk} NUTS transitions of every chain", "launch_steps": groups, "iter_warmup": nw, "iter_sampling": ns,
                               "max_depth": args.max_depth,
                       **({"adapt_windows_init_window_term": args.adapt_windows} if args.adapt_windows else {}),
                       "baseline_config_index": 2 if (cfg == 1 and world > 1) else cfg,
                       "chains_per_gpu": C_tot, "total_chains": C_tot * world, "posteriors": per_post,
                       "all_gather_bytes_per_rank": gathered_bytes,
                       "parallelism": (f"chains sharded {C_tot}/GPU x {world}, no data-path collective; one RCCL all-gather of the "
                                       f"draws-of-interest (device buffers); " if world > 1 else f"{C_tot} chains; ") +
                                      (f"each chain on two clusters of {K} workgroups, one per end of the NUTS trajectory "
                                              f"({C_tot * K * 2} of 256 CUs busy)"
                                       if K > 1 and sides == 2 else
                                       f"each chain on a cluster of {K} workgroups ({C_tot * K} of 256 CUs busy)" if K > 1
                                       else "one workgroup per chain")},
            "leapfrogs": int(leapfrogs), "seconds": elapsed,
            "us_per_leapfrog_per_chain": 1e6 * kernel_ms_max * 1e-3 * C_tot / max(sum(lf_local), 1),
            "ess_bulk_min": min(ess_all) if ess_all else None,
            "ess_per_sec": (min(ess_all) / samp_time) if ess_all else None,
            "rhat_max": max(rhat_all) if rhat_all else None, "sampling_seconds": samp_time,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_note": traffic_note,
                         "kernel": kernel, "algorithmic_bytes_per_leapfrog": bpl if len(bpl) > 1 else bpl[0],
                         "leapfrogs_in_launches": int(sum(lf_local)), "launch_ms_total": kernel_ms, "launches": len(groups),
                         "avg_launch_ms": kernel_ms / max(len(groups), 1),
                         **({"matrix_passes": dense_t[1], "matrix_pass_ms_total": dense_t[0], "matrix_bytes_streamed": dense_t[2],
                             "avg_pass_ms": dense_t[0] / max(dense_t[1], 1), "leaf_rounds": dense_t[3],
                             "metric_storage": args.metric_storage} if dense else {}),
                         **({"pooled_metric": True, "kernel": "k_dn_pool_mm",
                             # the pooled pass is a D x D times D x R product: its second bound is the fp64 matrix peak
                             # (MI355X_MICROARCH.md: 78.6 TFLOP/s dense)
                             "mfma": {"bound": "mfma", "achieved": 2.0 * hs[0].D ** 2 * 2.0 * sum(lf_local) / max(dense_t[0] * 1e-3,
                                     1e-9) / 1e12, "peak": 78.6, "unit": "TFLOP/s",
                                      "frac": 2.0 * hs[0].D ** 2 * 2.0 * sum(lf_local) / max(dense_t[0] * 1e-3, 1e-9) / 1e12 / 78.6,
                                      "note": "2 D^2 flops per right-hand side, two right-hand sides per counted leapfrog (the first "
                                              "pass of a transition carries three: not counted), "
                                              "over the time of the passes; the pass streams the FULL symmetric matrix (8 D^2 "
                                                      "bytes; 4 D^2 with --metric-storage f32) once per round for all chains"}}
                            if dense and args.pooled_metric else {}),
                         "note": (f"latency-bound at {C_tot} chains ({C_tot * K * sides} of 256 CUs busy): the state of a chain stays in "
                                  f"L2, a leapfrog "
                                  "is a chain of dependent phases and exchanges between the CUs of a cluster; see DESIGN.md")
                                 if not dense else
                                 ("pooled dense metric (potus_opts.pooled_metric, a declared deviation from Stan): every leaf "
                                         "ROUND streams the handle's one full "
                                  "symmetric D x D inverse metric (8 D^2 bytes; 4 D^2 when it is kept rounded to fp32) once for "
                                          "all chains and multiplies it with their "
                                          "right-hand sides on the fp64 matrix "
                                  "cores; achieved = bytes loaded by the passes / their time; roofline.mfma is the same passes "
                                          "against the matrix peak"
                                  if args.pooled_metric else
                                  "dense metric: every leapfrog streams the upper triangle of the chain's D x D inverse metric "
                                          "(4 D^2 bytes; the "
                                  "survey's 8 D^2 assumed the full matrix); achieved = bytes loaded by the matrix passes / their time")},
        }
        if dense:
            at = hs[0].dense_adapt_timing()
            first_adapted = next((q for q in per_step if q["window_ends_so_far"] > 0), None)
            adapted = [p for p in per_step if p["window_ends_so_far"] > 0 and p is not first_adapted]
            mv_ms, ad_s = sum(p["matrix_pass_ms"] for p in adapted), sum(p["seconds"] for p in adapted)
            line["dense"] = {
                "window_ends": at["window_ends"], "window_end_seconds": (at["cov_ms"] + at["chol_ms"] + at["init_stepsize_ms"]) * 1e-3,
                "covariance_seconds": at["cov_ms"] * 1e-3, "cholesky_seconds": at["chol_ms"] * 1e-3,
                        "init_stepsize_seconds": at["init_stepsize_ms"] * 1e-3,
                "cholesky_tflops": ((1 if args.pooled_metric else work[0][3]) * hs[0].D ** 3 / 3.0 * at["window_ends"])
                                   / max(at["chol_ms"] * 1e-3, 1e-9) / 1e12,
                "matrices_factored_per_window_end": 1 if args.pooled_metric else work[0][3],
                "adapted_phase": ({"steps": len(adapted), "leapfrogs": sum(p["leapfrogs"] for p in adapted),
                        "seconds": sum(p["seconds"] for p in adapted),
                                   "leapfrogs_per_sec": sum(p["leapfrogs"] for p in adapted) / max(sum(p["seconds"] for p in adapted),
                                           1e-9),
                                   "matrix_pass_TBps": sum(p["matrix_bytes"] for p in adapted) / max(mv_ms, 1e-9) / 1e9,
                                   "matrix_pass_share_of_wall": mv_ms * 1e-3 / max(ad_s, 1e-9),
                                   "note": "the steps after the one in which the first window ended: transitions under the adapted dense "
                                           "metric"}
                                  if adapted else None),
                "per_step": per_step}
        if world == 1 and cfg == 1 and not args.no_saturated:
            # The same posterior with the GPU full: 256 chains, one workgroup per chain (k_run).  Not the metric's
            # configuration -- a reference point for what the kernels deliver when parallelism is not the limit.
            try:
                _, data, variant, _, _ = work[0]

                def short_run(chains, twin_):
                    hsat = Handle(data, variant, chains=chains, num_warmup=60, num_samples=0, seed=args.seed + 7, device=local,
                            cus_per_chain=1, twin=twin_)
                    hsat.init()
                    ms_s, lf_s = 0.0, 0
                    for _ in range(3):
                        hsat.run(20)
                        ms1, lf1 = hsat.last_run_timing()
                        ms_s += ms1; lf_s += lf1
                    hsat.close()
                    return lf_s / (ms_s * 1e-3)

                rate = short_run(256, 0)
                r128 = [short_run(128, t) for t in (0, 1)]
                line["saturated"] = {"chains": 256, "cus_per_chain": 1, "kernel": "k_run", "iterations": 60, "value": rate,
                        "unit": "leapfrogs/s",
                                     "roofline_frac": rate * bpl[0] / 1e9 / HBM_PEAK_GBS,
                                     "chains_128": {"one_workgroup_per_chain": r128[0], "two_workgroups_per_chain": r128[1],
                                             "kernel": "k_run / k_run_twin",
                                                    "roofline_frac": max(r128) * bpl[0] / 1e9 / HBM_PEAK_GBS,
                                                    "note": "65-128 chains: the library's choice is two workgroups per chain, one per "
                                                            "end of the trajectory"},
                                     "note": "short warm-up run of 256 chains on the same posterior; kernel time of the launches"}
            except Exception as e:                     # never let the side measurement spoil the bench line
                line["saturated"] = {"error": str(e)[:200]}
        if not args.no_cpu_baseline and world == 1 and cfg == 4:
            # the CPU port under a dense metric: what it can afford of this shape -- a few leapfrogs of ONE chain, the D x D product
            # spread over the box's cores (oracle_time_leapfrogs_dense); every chain would take the same, one after the other
            from oracle_lib import OracleModel
            _, data, variant, C, _ = work[0]
            om = OracleModel(data, variant)
            nlf = 6
            secs, nthreads, mbytes = om.time_leapfrogs_dense(nlf)
            line["cpu_baseline"] = {"value": nlf / secs, "unit": "leapfrogs/s", "cores": nthreads, "kind": "port",
                                    "sample": f"{nlf} leapfrogs of one chain of the same posterior (D = {hs[0].D}) under a dense "
                                              f"{mbytes / 1e9:.2f} GB inverse metric, the rows of the "
                                              f"matrix-vector product over {nthreads} OpenMP threads (oracle/potus_oracle.c: "
                                                      f"dense_e_metric::dtau_dp + the scan/sparse gradient): {secs:.1f} s",
                                    "seconds": secs, "matrix_GBps": nlf * mbytes / secs / 1e9}
            line["value_over_cpu_sample"] = line["value"] / line["cpu_baseline"]["value"]
        if not args.no_cpu_baseline and world == 1 and cfg in (1, 2):   # the CPU port is timed beside the single-GPU run only
            _, data, variant, C, _ = work[0]
            short = (150, 100)
            # the same complete short configuration on the GPU (after the timed region), to put measured ESS / s side by side
            hg = Handle(data, variant, chains=C, num_warmup=short[0], num_samples=short[1], seed=args.seed, device=local,
                    cus_per_chain=args.cus_per_chain, twin=twin)
            tg0 = time.perf_counter()
            hg.init(); hg.run(short[0])
            tg1 = time.perf_counter()
            hg.run(short[1])
            tg2 = time.perf_counter()
            S_, T_ = int(data["S"]), int(data["T"])
            a_mu = hg.layout["mu_b"][0]
            mu = np.transpose(hg.write_array(a_mu + S_ * (T_ - 1), a_mu + S_ * T_, short[1]), (1, 0, 2))
            lpc = np.transpose(hg.write_array(0, 1, short[1]), (1, 0, 2))
            g_ess = ess_min(np.concatenate([lpc, mu, 1.0 / (1.0 + np.exp(-mu))], axis=2))
            g_lf = hg.total_leapfrogs()
            hg.close()
            cb = cpu_baseline(data, variant, C, args.seed, nw, ns, short)
            cb["short_config"]["gpu"] = dict(leapfrogs=g_lf, seconds=tg2 - tg0, sampling_seconds=tg2 - tg1,
                    leapfrogs_per_sec=g_lf / (tg2 - tg0),
                                             ess_bulk_min=g_ess, ess_per_sec=g_ess / (tg2 - tg1))
            like = dict(leapfrogs_per_sec=(g_lf / (tg2 - tg0)) / cb["short_config"]["leapfrogs_per_sec"],
                        ess_per_sec=(g_ess / (tg2 - tg1)) / cb["short_config"]["ess_per_sec"],
                        wall=cb["short_config"]["seconds"] / (tg2 - tg0),
                        of=f"the complete short configuration {C} chains x ({short[0]} + {short[1]}), seed {args.seed}, run to the "
                                f"end on both sides")
            cb["short_config"]["gpu_over_cpu"] = like
            # the like-for-like ratios first; the ratio against the bounded sample (`value`) after them, named for what it is
            line["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                    "sample": cb["sample"],
                                    "like_for_like_gpu_over_cpu": like, **{k: v for k, v in cb.items() if k not in ("value", "unit",
                                            "cores", "kind", "sample")}}
            line["gpu_over_cpu_like_for_like"] = like
            line["value_over_cpu_sample"] = line["value"] / cb["value"]
            line["value_over_cpu_leapfrog_loop"] = line["value"] / cb["leapfrog_loop_scan_sparse_value"]
            line["gpu_over_cpu_note"] = ("gpu_over_cpu_like_for_like: the same complete short configuration on both sides (leapfrogs / "
                                         "s, ESS / s of its own draws, wall time); "
                                         f"value_over_cpu_sample: this line's whole-run rate over the rate of the port's first "
                                                 f"{cb['iterations_sampled'][0]}-{cb['iterations_sampled'][1]} "
                                         f"iterations of the same run on {cb['cores']} host cores (a bounded sample of early "
                                                 f"warm-up, not the same iterations); "
                                         "value_over_cpu_leapfrog_loop: over the port's bare leapfrog loop, which no CPU sampler "
                                                 "can exceed.  None of them says anything "
                                         "about kernel quality: roofline.frac does")
        if world == 1 and cfg == 1 and not args.no_side:
            for h in hs:
                h.close()
            torch.cuda.empty_cache()
            line["side"] = side_measurements(args.seed, headline=seed_summary({**line, "seed": args.seed}))
        if world > 1 and cfg in (1, 2) and not args.no_side and sp_side:
            for h in hs:
                h.close()
            hs = []
            torch.cuda.empty_cache()
            parallel.barrier()                                   # every rank has released its sampler: the GPUs are free
            torch.cuda.synchronize()
            line["side"] = {"single_process": single_process_side(args, world)}
            wait_for_rank0.set("bench_side_done", "1")           # (see below)
        print(json.dumps(line), flush=True)
    elif world > 1 and cfg in (1, 2) and not args.no_side and sp_side:
        for h in hs:
            h.close()
        hs = []
        torch.cuda.empty_cache()
        parallel.barrier()
        torch.cuda.synchronize()
        # Rank 0's child process now needs every compute unit of every GPU resident for its cluster launches: the other ranks must NOT wait
        # for it inside an
        # RCCL collective (a barrier is a kernel that spins on the GPU and holds compute units) -- they block on the job's TCP store, on the
        # CPU
        wait_for_rank0.wait(["bench_side_done"], __import__("datetime").timedelta(seconds=900))
    for h in hs:
        h.close()
    parallel.barrier()
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
