#!/usr/bin/env python3
"""Headline benchmark: leapfrog steps/s (and ESS/s) of adaptive NUTS on the 2016 poll model.

Workload (BASELINE.json configs[1]): scripts/model/final_2016.R's posterior (51 states x 254
days, 1619 polls, D = 15 098), 8 chains per MI355X, 1000 warmup + 1000 sampling iterations,
seed 1843, NUTS diag_e, delta 0.8, max depth 10.  A "step" is one NUTS transition of every
chain on the GPU; `--steps K` runs K//2 warmup + K - K//2 sampling transitions from a fresh
initialisation (default K = 2000 = the configuration above).  `--warmup W` runs W untimed
transitions of a throw-away sampler first (clocks, code objects, allocator).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--chains-per-gpu C]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Each chain runs on a cluster of K workgroups (K compute units of one XCD; K = 16 for 8 chains, see
potus_cluster.hpp; --cus-per-chain 1 selects the one-workgroup-per-chain kernel instead).
N > 1: one process per GPU; rank r owns chains [r*C, (r+1)*C) (RNG streams keyed by global chain
id), no communication while sampling, one RCCL all-gather of the draws-of-interest for pooled
R-hat / ESS (inside the timed region).  Weak scaling: C chains per GPU whatever N is.

Rank 0 prints ONE JSON line.  `value` = leapfrogs of all chains on all GPUs / max-over-ranks
wall time of (init + K transitions [+ all-gather]), inputs already resident in HBM.
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


def algorithmic_bytes_per_leapfrog(d, variant="full"):
    """48*D + 36*N_state + 32*N_nat + 24*S^2 + 16*S (diag metric)."""
    from us_potus_model_amd import _abi
    D = _abi.num_params(d, variant)
    S = int(d["S"])
    per_state, per_nat = (36, 32) if variant == "full" else (28, 24)
    return 48 * D + per_state * int(d["N_state_polls"]) + per_nat * int(d["N_national_polls"]) + 24 * S * S + 16 * S


def measured_traffic(kernel):
    """HBM bytes per leapfrog from the committed rocprofv3 PMC passes of this command (profiles/*pmc_traffic.json,
    written by the recipe in scripts/profile_round.sh): FETCH_SIZE and WRITE_SIZE cannot be collected from inside
    the benchmark, so the bench line quotes the per-leapfrog figure of the latest committed pass for the same kernel."""
    best = None
    for f in sorted((ROOT / "profiles").glob("*pmc_traffic.json")):
        try:
            d = json.loads(f.read_text())
        except (OSError, ValueError):
            continue
        if d.get("kernel") == kernel:
            best = (f.name, d)
    return best


def _cpu_worker(args):
    data, variant, fast, budget = args
    from oracle_lib import OracleModel
    m = OracleModel(data, variant)
    n, t, batch = 0, 0.0, 200
    while t < budget:
        t += m.time_leapfrogs(batch, eps=0.01, fast=fast, seed=1 + n)
        n += batch
    return n, t


def cpu_baseline(data, variant, chains, budget=10.0):
    """The oracle (a port of the reference's CPU path: literal dense recursion of stan:86 with a
    hand-written reverse sweep -- no AD tape, so faster than Stan itself) timed on this box's
    host cores, one chain per core as the reference runs them (final_2016.R:536)."""
    procs = max(1, min(chains, os.cpu_count() or 1))
    out = {}
    with mp.get_context("spawn").Pool(procs) as pool:
        for key, fast in (("value", 0), ("scan_sparse_value", 1)):
            res = pool.map(_cpu_worker, [(data, variant, fast, budget if not fast else budget / 2)] * procs)
            out[key] = float(sum(n / t for n, t in res))
    return dict(value=out["value"], unit="leapfrogs/s", cores=procs, kind="port",
                sample=f"{procs} processes x ~{budget:.0f} s of leapfrogs (eps 0.01, unit metric) on the 2016 posterior, "
                       f"literal stan:86 recursion; scan_sparse_value = same with the reformulated gradient",
                scan_sparse_value=out["scan_sparse_value"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--chains-per-gpu", type=int, default=8)
    ap.add_argument("--cus-per-chain", type=int, default=0, help="workgroups per chain (0 = auto: 16, 8 or 1 by what fits)")
    ap.add_argument("--chunk", type=int, default=100, help="transitions per kernel launch")
    ap.add_argument("--seed", type=int, default=1843)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-saturated", action="store_true", help="skip the 256-chain reference point")
    ap.add_argument("--cpu-budget", type=float, default=10.0)
    args = ap.parse_args()

    import torch
    from us_potus_model_amd import Handle, dataprep, diagnostics as dg, parallel

    # POTUS_DIST_BACKEND=gloo: development aid -- every rank on GPU 0 and CPU tensors in the collectives, to exercise
    # the N > 1 flow on a one-GPU box (the measured configuration is one rank per GPU over RCCL)
    dev_backend = os.environ.get("POTUS_DIST_BACKEND")
    rank, world, local = parallel.init_process_group(dev_backend)
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    if dev_backend == "gloo":
        local = 0
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if dev_backend == "gloo":
        dev = None                                   # collectives on CPU tensors

    data = dataprep.load_npz(ROOT / "tests" / "golden" / "data_2016.npz")["data"]
    variant = "full"
    C = args.chains_per_gpu
    total_chains = C * world
    nw, ns = args.steps // 2, args.steps - args.steps // 2

    if args.warmup > 0:  # untimed: throw-away sampler
        hw = Handle(data, variant, chains=C, chain_id_offset=rank * C, num_warmup=args.warmup, num_samples=0,
                    seed=args.seed + 1, device=local, cus_per_chain=args.cus_per_chain)
        hw.init()
        hw.run(args.warmup)
        hw.close()

    h = Handle(data, variant, chains=C, chain_id_offset=rank * C, num_warmup=nw, num_samples=ns, seed=args.seed,
               device=local, cus_per_chain=args.cus_per_chain)
    K = h.cus_per_chain
    S, T = int(data["S"]), int(data["T"])
    a_mu = h.layout["mu_b"][0]

    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    h.init()
    done, kernel_ms, t_warm_end = 0, 0.0, None
    while done < args.steps:
        n = min(args.chunk, args.steps - done, (nw - done) if done < nw else args.steps)
        h.run(n)
        done += n
        kernel_ms += h.last_run_timing()[0]
        if done == nw:
            torch.cuda.synchronize()
            t_warm_end = time.perf_counter()
    pooled = None
    if world > 1:  # the one exchange of the path: pool draws-of-interest over xGMI
        # (lp__ and mu_b[:, T] through write_array: two small column ranges, not the 1 GB draws array)
        interest = np.concatenate([np.transpose(h.write_array(0, 1, ns), (1, 0, 2)),
                                   np.transpose(h.write_array(a_mu + S * (T - 1), a_mu + S * T, ns), (1, 0, 2))], axis=2)
        pooled = parallel.all_gather_draws(interest, total_chains, device=dev)
    torch.cuda.synchronize()
    parallel.barrier()
    t1 = time.perf_counter()

    elapsed = parallel.max_over_ranks(t1 - t0, dev)
    leapfrogs_local = h.total_leapfrogs()
    leapfrogs = parallel.sum_over_ranks(float(leapfrogs_local), dev)
    kernel_ms_max = parallel.max_over_ranks(kernel_ms, dev)
    samp_time = parallel.max_over_ranks(t1 - (t_warm_end or t0), dev)

    if pooled is None:
        pooled = np.concatenate([np.transpose(h.write_array(0, 1, ns), (1, 0, 2)),
                                 np.transpose(h.write_array(a_mu + S * (T - 1), a_mu + S * T, ns), (1, 0, 2))], axis=2)
    st, dv = h.chain_status()

    if rank == 0:
        ess = None
        if ns >= 8:
            ps = 1.0 / (1.0 + np.exp(-pooled[:, :, 1:]))
            cols = np.concatenate([pooled, ps], axis=2)
            ess = float(min(dg.ess_bulk(cols[:, :, j]) for j in range(cols.shape[2])))
            rh = float(max(dg.rhat(cols[:, :, j]) for j in range(cols.shape[2])))
        bpl = algorithmic_bytes_per_leapfrog(data, variant)
        achieved = leapfrogs_local * bpl / (kernel_ms * 1e-3) / 1e9  # this rank's kernel, its own stream's events
        kernel = "k_cl_run" if K > 1 else "k_run"
        tr = measured_traffic(kernel)
        traffic = tr[1]["hbm_bytes_per_leapfrog"] * leapfrogs_local / (kernel_ms * 1e-3) / 1e9 if tr else None
        line = {
            "metric": "leapfrog_steps_per_sec", "value": leapfrogs / elapsed, "unit": "leapfrogs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "2016 poll data (fixture built from the reference CSVs, tests/golden/data_2016.npz); random inits",
            "config": {"workload": "configs[1]: 2016 backtest, adaptive NUTS diag_e, 8 chains per MI355X, "
                                   f"{nw} warmup + {ns} sampling, seed {args.seed}",
                       "chains_per_gpu": C, "total_chains": total_chains, "D": h.D, "S": S, "T": T,
                       "polls": int(data["N_state_polls"]) + int(data["N_national_polls"]),
                       "cus_per_chain": K,
                       "parallelism": (f"chains sharded {C}/GPU x {world}, no data-path collective; one all-gather of draws-of-interest; "
                                       if world > 1 else f"{C} chains; ") +
                                      (f"each chain on a cluster of {K} workgroups ({C * K} of 256 CUs busy)" if K > 1
                                       else "one workgroup per chain")},
            "leapfrogs": int(leapfrogs), "seconds": elapsed,
            "us_per_leapfrog_per_chain": 1e6 * kernel_ms_max * 1e-3 * C / max(leapfrogs_local, 1),
            "ess_bulk_min": ess, "ess_per_sec": (ess / samp_time) if ess else None, "rhat_max": rh if ess else None,
            "divergent_transitions": int(sum(dv)), "chain_status": st,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_note": (f"GB/s over the same launches: {tr[1]['hbm_bytes_per_leapfrog']:.0f} HBM bytes per leapfrog "
                                          f"(2 x FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes, profiles/{tr[0]}) x leapfrogs / launch time"
                                          if tr else "no PMC pass committed for this kernel"),
                         "kernel": kernel, "algorithmic_bytes_per_leapfrog": bpl,
                         "leapfrogs_in_launches": int(leapfrogs_local), "launch_ms_total": kernel_ms,
                         "note": f"latency-bound at {C} chains ({C * K} of 256 CUs busy): the state of a chain stays in L2, "
                                 "the leapfrog is a chain of dependent exchanges between the CUs of a cluster; see DESIGN.md"},
        }
        if world == 1 and not args.no_saturated:
            # The same posterior with the GPU full: 256 chains, one workgroup per chain (k_run).  Not the metric's
            # configuration -- a reference point for what the kernels deliver when parallelism is not the limit.
            try:
                hs = Handle(data, variant, chains=256, num_warmup=60, num_samples=0, seed=args.seed + 7, device=local, cus_per_chain=1)
                hs.init()
                ms_s, lf_s = 0.0, 0
                for _ in range(3):
                    hs.run(20)
                    ms1, lf1 = hs.last_run_timing()
                    ms_s += ms1; lf_s += lf1
                hs.close()
                rate = lf_s / (ms_s * 1e-3)
                line["saturated"] = {"chains": 256, "cus_per_chain": 1, "kernel": "k_run", "iterations": 60, "value": rate, "unit": "leapfrogs/s",
                                     "roofline_frac": rate * bpl / 1e9 / HBM_PEAK_GBS,
                                     "note": "short warm-up run of 256 chains on the same posterior; kernel time of the launches"}
            except Exception as e:                     # never let the side measurement spoil the bench line
                line["saturated"] = {"error": str(e)[:200]}
        if not args.no_cpu_baseline and world == 1:   # the CPU port is timed beside the single-GPU run only
            line["cpu_baseline"] = cpu_baseline(data, variant, C, args.cpu_budget)
            line["speedup_vs_cpu_port"] = line["value"] / line["cpu_baseline"]["value"]
        print(json.dumps(line), flush=True)
    h.close()
    parallel.barrier()
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
