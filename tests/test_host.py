"""CPU tests of host logic: data prep fixture, diagnostics, chain partitioning, gloo all-gather."""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from us_potus_model_amd import _abi, diagnostics as dg, parallel, synthetic

ROOT = Path(__file__).resolve().parent.parent


def test_data_2016_fixture(data_2016):
    d = data_2016
    # sizes re-derived from the reference CSVs (T = 254 at final_2016.R:556)
    assert (d["N_state_polls"], d["N_national_polls"], d["T"], d["S"], d["P"], d["M"], d["Pop"]) == (1258, 361, 254, 51, 161, 3, 3)
    assert d["state"].min() >= 1 and d["state"].max() <= 51
    assert d["day_state"].max() <= 254 and d["day_national"].max() <= 254
    assert abs(d["state_weights"].sum() - 1) < 1e-12
    cov = d["state_covariance_0"]
    assert np.allclose(cov, cov.T) and np.linalg.eigvalsh(cov).min() > 0
    # cov_matrix(51, 0.07^2, 0.9) * corr; make.positive.definite lifts the diagonal slightly above 1
    assert (np.diag(cov) >= 0.07 ** 2 - 1e-12).all() and (np.diag(cov) < 0.07 ** 2 * 1.1).all()
    assert d["polling_bias_scale"] == pytest.approx(0.052) and d["mu_b_T_scale"] == pytest.approx(0.12)
    assert d["random_walk_scale"] == pytest.approx(0.05 / np.sqrt(300) * 4)
    assert (d["n_democrat_state"] <= d["n_two_share_state"]).all()
    # national prior close to the value the script prints (final_2016.R:412-414): Clinton ~ 0.51
    nat = (1 / (1 + np.exp(-d["mu_b_prior"]))) @ d["state_weights"]
    assert 0.49 < nat < 0.53


@pytest.mark.parametrize("year,sizes", [(2012, (966, 191, 251, 160)), (2008, (960, 251, 247, 90))])
def test_backtest_fixtures(cases, year, sizes):
    """2012 / 2008 backtests (final_2012.R:63-556, final_2008.R:63-560): sizes as re-derived in SURVEY.md section 8
    (T: the scripts index 250 / 247 at final_2012.R:582, final_2008.R:586); they run the no-mode model."""
    d, variant = cases[str(year)]
    assert variant == "no_mode_adjustment"
    assert (d["N_state_polls"], d["N_national_polls"], d["T"], d["P"]) == sizes and d["S"] == 51
    assert 1 <= d["state"].min() and d["state"].max() <= 51
    assert d["day_state"].max() <= d["T"] and d["day_national"].max() <= d["T"] and min(d["day_state"].min(), d["day_national"].min()) == 1
    assert d["poll_state"].max() <= d["P"] and d["poll_national"].max() <= d["P"]
    assert abs(d["state_weights"].sum() - 1) < 1e-12 and "sigma_a" in d          # the unused entry Stan ignores
    assert (d["n_democrat_state"] <= d["n_two_share_state"]).all() and (d["n_democrat_national"] <= d["n_two_share_national"]).all()
    assert set(np.unique(d["unadjusted_state"])) <= {0.0, 1.0}
    assert d["mu_b_T_scale"] == pytest.approx(0.12)                                # RUN_DATE = election day
    nat = (1 / (1 + np.exp(-d["mu_b_prior"]))) @ d["state_weights"]
    assert 0.48 < nat < 0.56                                                       # Obama two-party prior


@pytest.mark.skipif(not Path("/root/reference/data").exists(), reason="reference CSVs only exist in the build container")
@pytest.mark.parametrize("year", [2012, 2008])
def test_dataprep_reproduces_backtest_fixtures(cases, year):
    from us_potus_model_amd import dataprep
    d = dataprep.build_backtest("/root/reference/data", year)["data"]
    for k, v in cases[str(year)][0].items():
        assert np.allclose(np.asarray(d[k], dtype=float), np.asarray(v, dtype=float), rtol=1e-12, atol=1e-14), k


@pytest.mark.skipif(not Path("/root/reference/data").exists(), reason="reference CSVs only exist in the build container")
def test_dataprep_reproduces_fixture(data_2016):
    from us_potus_model_amd import dataprep
    d = dataprep.build_2016("/root/reference/data")["data"]
    for k, v in data_2016.items():
        assert np.allclose(np.asarray(d[k], dtype=float), np.asarray(v, dtype=float), rtol=1e-12, atol=1e-14), k


def test_make_positive_definite_and_cov_matrix():
    from us_potus_model_amd import dataprep
    m = dataprep.cov_matrix(4, 0.04, 0.9)
    assert np.allclose(np.diag(m), 0.04) and m[0, 1] == pytest.approx(0.036)
    a = np.array([[1.0, 0.99, 0.0], [0.99, 1.0, 0.99], [0.0, 0.99, 1.0]])   # indefinite
    p = dataprep.make_positive_definite(a)
    assert np.linalg.eigvalsh(p).min() > 0 and np.allclose(p, p.T)
    spd = np.eye(3) * 2
    assert np.allclose(dataprep.make_positive_definite(spd), spd)


def test_synthetic_variants():
    f, n = synthetic.small("full"), synthetic.small("no_mode_adjustment")
    assert "poll_mode_state" in f and "poll_mode_state" not in n and "sigma_a" in n
    assert _abi.num_params(f, "full") - _abi.num_params(n, "no_mode_adjustment") == 3 + 3 + 2 + 24
    st = synthetic.stress()
    assert _abi.num_params(st, "full") == 41610


def test_diagnostics_iid_and_ar1():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((4, 2000))
    assert 0.85 * 8000 < dg.ess_bulk(x) < 1.2 * 8000 and dg.rhat(x) < 1.01
    phi = 0.9
    y = np.zeros((4, 4000))
    e = rng.standard_normal((4, 4000))
    for t in range(1, 4000):
        y[:, t] = phi * y[:, t - 1] + e[:, t]
    expect = 16000 * (1 - phi) / (1 + phi)
    assert 0.6 * expect < dg.ess_mean(y) < 1.6 * expect
    z = x.copy()
    z[0] += 3.0
    assert dg.rhat(z) > 1.2


def test_chain_block_partition():
    for total in (1, 8, 64, 96, 7):
        for world in (1, 2, 4, 8):
            blocks = [parallel.chain_block(total, r, world) for r in range(world)]
            assert sum(n for _, n in blocks) == total
            off = 0
            for o, n in blocks:
                assert o == off
                off += n


_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch.distributed as dist
from us_potus_model_amd import parallel
rank, world, _ = parallel.init_process_group("gloo")
total = 5
off, n = parallel.chain_block(total, rank, world)
local = np.zeros((n, 3, 2))
for c in range(n):
    local[c] = 100 * (off + c + 1) + np.arange(6).reshape(3, 2)
pooled = parallel.all_gather_draws(local, total)
assert pooled.shape == (total, 3, 2), pooled.shape
for c in range(total):
    assert np.array_equal(pooled[c], 100 * (c + 1) + np.arange(6).reshape(3, 2))
# the device-side pooling used by bench.py (torch tensors [draw, chain, col]; CPU tensors under gloo)
import torch
loc = torch.zeros((4, 3, 2), dtype=torch.float64)
for c in range(3):
    loc[:, c, :] = 1000 * rank + 10 * c + torch.arange(8, dtype=torch.float64).reshape(4, 2)
pl = parallel.all_gather_chains(loc, None)
assert tuple(pl.shape) == (4, 3 * world, 2)
for r in range(world):
    for c in range(3):
        assert torch.equal(pl[:, 3 * r + c, :], 1000 * r + 10 * c + torch.arange(8, dtype=torch.float64).reshape(4, 2))
# pooled posterior summaries (what potus_posterior_summary_many does across GPUs, restated): every rank holds the predicted_score
# draws of its own chains; after the gather, the summaries of the pooled block are those of all chains together, on every rank
import sys
sys.path.insert(0, os.path.join(sys.argv[1], "oracle"))
from posterior_summary_ref import posterior_summary
rng = np.random.default_rng(99)
allps = 1.0 / (1.0 + np.exp(-rng.standard_normal((50, 2 * world, 3 * 4))))          # [draw, chain, T * S], the same on every rank
mine = torch.as_tensor(allps[:, 2 * rank:2 * rank + 2, :].copy())
got = parallel.all_gather_chains(mine, None).numpy()
assert np.array_equal(got, allps)
w8, ev = np.array([0.1, 0.2, 0.3, 0.4]), np.array([100.0, 150.0, 200.0, 88.0])
a = posterior_summary(got.reshape(-1, 3, 4), w8, ev)
b = posterior_summary(allps.reshape(-1, 3, 4), w8, ev)
assert all(np.array_equal(a[k], b[k]) for k in a)
# convergence diagnostics of the pooled block: the columns are dealt to the ranks, the per-column results gathered (bench.py, N > 1)
ncol = 11
ca, cb = parallel.column_block(ncol, rank, world)
assert (ca, cb) == ((0, 6) if rank == 0 else (6, 11))
res = parallel.all_gather_columns(np.arange(ca, cb, dtype=np.float64) * 1.5, ncol, None)
assert np.array_equal(res, np.arange(ncol) * 1.5)
one = parallel.all_gather_columns(np.full(1, 7.0) if rank == 0 else np.zeros(0), 1, None)        # (a rank without columns still takes part)
assert one.shape == (1,) and one[0] == 7.0
assert parallel.max_over_ranks(float(rank)) == world - 1
assert parallel.sum_over_ranks(1.0) == world
# the pooled dense metric across ranks (potus_opts.pooled_metric = 2; SURVEY 8e: the path's one all-reduce): every rank holds count, mean and
# M2 = sum of centred outer products of ITS chains' window draws -- different counts per rank --; after parallel.pool_window_moments every rank
# holds the count, the mean and the M2 of ALL draws, hence covar_adaptation's estimate of the pooled sample
Dp, LDp = 37, 40
rngp = np.random.default_rng(5)
alld = [rngp.standard_normal((11 + 6 * r, Dp)) * rngp.uniform(0.3, 2.0, Dp) + r for r in range(world)]   # the same on every rank
mine_d = alld[rank]
mean_l = torch.as_tensor(mine_d.mean(axis=0))
m2_l = torch.zeros((Dp, LDp), dtype=torch.float64)
m2_l[:, :Dp] = torch.as_tensor((mine_d - mine_d.mean(axis=0)).T @ (mine_d - mine_d.mean(axis=0)))
n_tot, gmean = parallel.pool_window_moments(float(len(mine_d)), mean_l, m2_l, None)
cat = np.concatenate(alld)
assert n_tot == len(cat) and np.allclose(gmean.numpy(), cat.mean(axis=0), rtol=1e-13, atol=1e-14)
want = (cat - cat.mean(axis=0)).T @ (cat - cat.mean(axis=0))
assert np.allclose(m2_l[:, :Dp].numpy(), want, rtol=1e-12, atol=1e-12 * np.abs(want).max()) and float(m2_l[:, Dp:].abs().max()) == 0.0
N = n_tot
est = (N / (N + 5.0)) * m2_l[:, :Dp].numpy() / (N - 1.0) + 1e-3 * (5.0 / (N + 5.0)) * np.eye(Dp)
assert np.allclose(est, (N / (N + 5.0)) * np.cov(cat.T) + 1e-3 * (5.0 / (N + 5.0)) * np.eye(Dp), rtol=1e-12, atol=1e-14)
parallel.pool_window_m2(torch.ones((5, 3), dtype=torch.float64), None, slice_bytes=24)      # (sliced: one row per all-reduce)
parallel.barrier()
dist.destroy_process_group()
print("ok", rank)
"""


def test_all_gather_world_size_2_gloo(tmp_path):
    """The N>1 path (chain partition + pooled draws) on CPU with gloo, world_size 2."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", str(script), str(ROOT)],
                         capture_output=True, text=True, env=env, timeout=240)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("ok") == 2


def test_bench_bookkeeping_on_the_committed_profiles():
    """bench.py's roofline bookkeeping without a GPU: the algorithmic bytes of the 2016 posterior (SURVEY section 8d:
    844 784 B per leapfrog), the committed counter passes it quotes for each mode of the cluster kernel, and the bench
    line committed for the round (one JSON line with the contract's keys, roofline and cpu_baseline)."""
    import importlib.util
    import json
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("bench_mod", ROOT / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from us_potus_model_amd import dataprep
    data = dataprep.load_npz(ROOT / "tests" / "golden" / "data_2016.npz")["data"]
    assert bench.algorithmic_bytes_per_leapfrog(data, "full", False) == 844784
    D = 15098
    assert bench.algorithmic_bytes_per_leapfrog(data, "full", True) == 844784 + 4 * D * D
    one, two = bench.measured_traffic("k_cl_run", 1), bench.measured_traffic("k_cl_run", 2)
    assert one and two and one[0] != two[0] and "two clusters" in two[1]["command"] and "two clusters" not in one[1]["command"]
    # (two clusters per chain: 1.04 x / 1.007 x the algorithmic bytes in rounds 3 / 4, 0.785 x since round 5 keeps a member's share of three vectors in LDS, 0.769 x with round 6's carry table)
    assert 0.5 < one[1]["hbm_bytes_per_leapfrog"] / 844784 < 0.8 and 0.6 < two[1]["hbm_bytes_per_leapfrog"] / 844784 < 1.3
    assert two[0] == "r06h_cl_twin_pmc_traffic.json"                    # the latest committed pass is the one the bench line quotes
    for committed in ("r03_bench_line.json", "r04b_bench_line.json", "r05_bench_line.json", "r05b_bench_line.json", "r06_bench_line.json",
                      "r06b_bench_line.json", "r06c_bench_line.json", "r06d_bench_line.json", "r06e_bench_line.json", "r06f_bench_line.json", "r06g_bench_line.json", "r06h_bench_line.json", "r06i_bench_line.json"):
        _check_committed_bench_line(json.loads([ln for ln in (ROOT / "profiles" / committed).read_text().splitlines() if ln.startswith("{")][0]),
                                    device_diagnostics=int(committed[1:3]) if int(committed[1:3]) >= 4 else 0)


def _check_committed_bench_line(line, device_diagnostics):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["metric"] == "leapfrog_steps_per_sec" and line["steps"] == 20 and line["config"]["iter_warmup"] == 1000 and line["config"]["iter_sampling"] == 1000
    r = line["roofline"]
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["peak"] == 8000.0
    assert abs(line["value"] - line["leapfrogs"] / line["seconds"]) < 1e-6 * line["value"]
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and line["rhat_max"] < 1.01
    # the CPU side is measured, not derived: a complete short configuration run to the end on both sides
    sc = cb["short_config"]
    if device_diagnostics >= 6:
        # round 6: the ESS / s of the short configuration lives under short_config only, the like-for-like ratios come first and
        # the ratio against the bounded sample is named for what it is
        assert "ess_per_sec_measured" not in cb and "speedup_vs_cpu_port" not in line
        assert sc["ess_per_sec"] > 0 and sc["gpu"]["ess_per_sec"] > sc["ess_per_sec"] and cb["like_for_like_gpu_over_cpu"] == sc["gpu_over_cpu"] == line["gpu_over_cpu_like_for_like"]
        assert list(cb)[:6] == ["value", "unit", "cores", "kind", "sample", "like_for_like_gpu_over_cpu"]
        assert cb["iterations_sampled"][1] < cb["iterations_configured"] == 2000 and "bounded sample" in cb["value_is"]
        assert abs(line["value_over_cpu_sample"] - line["value"] / cb["value"]) < 1e-9 and line["value_over_cpu_leapfrog_loop"] < line["value_over_cpu_sample"]
        assert abs(sc["gpu_over_cpu"]["ess_per_sec"] - sc["gpu"]["ess_per_sec"] / sc["ess_per_sec"]) < 1e-9
    else:
        assert sc["ess_per_sec_measured"] > 0 and sc["gpu"]["ess_per_sec_measured"] > sc["ess_per_sec_measured"] and cb["ess_per_sec_measured"] == sc["ess_per_sec_measured"]
        assert abs(line["speedup_vs_cpu_port"] - line["value"] / cb["value"]) < 1e-9 and line["speedup_vs_cpu_leapfrog_loop"] < line["speedup_vs_cpu_port"]
    assert line["config"]["all_gather_bytes_per_rank"] == 1000 * 8 * (1 + 51 * 254) * 8            # lp__ + all of mu_b
    if device_diagnostics:                                                                          # round 4: R-hat / ESS of every gathered column on the device
        dd = line["config"]["posteriors"]["2016"]["device_diagnostics"]
        extra = 51 if device_diagnostics >= 5 else 0                                                # round 5: + predicted_score[T, :] as columns of their own
        assert dd["columns"] == 1 + 51 * 254 + extra and dd["ess_bulk_min_all_columns"] > 100 and dd["rhat_max_all_columns"] < 1.05
    if device_diagnostics >= 5:
        # round 5: the other single-GPU configurations as side measurements under the same clock, the host's core count beside the cores used
        side = line["side"]
        assert {r["posterior"] for r in side["configs[0]"]["runs"]} == {"2016", "2012", "2008"} and side["configs[0]"]["baseline_config_index"] == 0
        assert side["configs[3]"]["baseline_config_index"] == 3 and side["configs[3]"]["value"] > 0 and side["configs[3]"]["roofline"]["kernel"] == "k_cl_run"
        assert side["configs[4]_preset"]["roofline"]["kernel"] == "k_dn_symv" and 0.5 < side["configs[4]_preset"]["roofline"]["frac"] < 0.9
        assert cb["host_cores_total"] >= cb["cores"]
    if device_diagnostics >= 6:
        # round 6: the configs[4] preset with the dense metric pooled over the GPU's chains (potus_opts.pooled_metric): the pass against BOTH its bounds
        pl = line["side"]["configs[4]_pooled"]
        assert pl["roofline"]["kernel"] == "k_dn_pool_mm" and pl["roofline"]["pooled_metric"] and 0.5 < pl["roofline"]["frac"] < 0.95
        assert pl["roofline"]["mfma"]["peak"] == 78.6 and 0.2 < pl["roofline"]["mfma"]["frac"] < 0.9 and pl["value"] > 3 * line["side"]["configs[4]_preset"]["value"]
    if device_diagnostics >= 6:
        # round 6: where the headline's seed sits -- the same command under the next two seeds, the default line's own run first
        sd = line["side"]["seeds"]
        assert list(line["side"])[0] == "seeds" and [r["seed"] for r in sd["runs"]] == [1843, 1844, 1845]
        assert sd["runs"][0]["leapfrogs_per_sec"] == line["value"] and sd["runs"][0]["seconds"] == line["seconds"]
        for k in ("leapfrogs_per_sec", "seconds", "ess_per_sec"):
            v = sorted(r[k] for r in sd["runs"])
            assert sd[k] == {"min": v[0], "median": v[1], "max": v[2]}
        assert all(len(r["treedepth_max_per_chain"]) == 8 and 7 <= max(r["treedepth_max_per_chain"]) <= 10 for r in sd["runs"])
        assert sd["deepest_tree_by_seed"] == [max(r["treedepth_max_per_chain"]) for r in sd["runs"]]
        assert line["config"]["posteriors"]["2016"]["treedepth_max_per_chain"] == sd["runs"][0]["treedepth_max_per_chain"]


def test_layout_plan_for_every_chain_count():
    """How potus_create resolves cus_per_chain = 0 / twin = -1 (potus_plan_cus_per_chain, potus_plan_sides: the product's own code, no
    device needed): 256 compute units, the reference's 2016 campaign of 254 days."""
    from us_potus_model_amd import sampler
    want = {1: (16, 2), 4: (16, 2), 8: (16, 2), 9: (14, 2), 10: (12, 2), 11: (11, 2), 12: (10, 2), 13: (16, 1), 16: (16, 1), 17: (8, 1),
            32: (8, 1), 33: (4, 1), 64: (4, 1), 65: (1, 2), 128: (1, 2), 129: (1, 1), 256: (1, 1), 1000: (1, 1)}
    for chains, kw in want.items():
        assert sampler.plan_layout(chains, 254) == kw, (chains, sampler.plan_layout(chains, 254))
    # explicit sizes: one cluster / workgroup unless a second one is asked for; a second one that does not fit is refused
    assert sampler.plan_layout(2, 254, cus_per_chain=16) == (16, 1) and sampler.plan_layout(2, 254, cus_per_chain=16, twin=1) == (16, 2)
    assert sampler.plan_layout(100, 254, cus_per_chain=1) == (1, 1) and sampler.plan_layout(100, 254, cus_per_chain=1, twin=1) == (1, 2)
    assert sampler.plan_layout(8, 254, twin=0) == (16, 1) and sampler.plan_layout(9, 254, twin=0) == (16, 1) and sampler.plan_layout(70, 254, twin=0) == (1, 1)
    for bad in (dict(chains=16, cus_per_chain=16, twin=1), dict(chains=129, cus_per_chain=1, twin=1), dict(chains=17, cus_per_chain=16),
                dict(chains=2, cus_per_chain=33), dict(chains=2, cus_per_chain=8, twin=1, metric="dense_e"), dict(chains=2, twin=2)):
        with pytest.raises(sampler.PotusError):
            sampler.plan_layout(T=254, **bad)
    # the dense metric never takes a second cluster; a smaller device changes the plan with its compute units
    assert sampler.plan_layout(8, 254, metric="dense_e") == (16, 1) and sampler.plan_layout(70, 254, metric="dense_e") == (1, 1)
    assert sampler.plan_layout(8, 254, n_cus=128) == (16, 1) and sampler.plan_layout(4, 254, n_cus=128) == (16, 2) and sampler.plan_layout(40, 254, n_cus=128) == (1, 2)
    # campaigns the one-workgroup kernels cannot hold (T > 256): a cluster with enough members for the days, or a refusal
    assert sampler.plan_layout(8, 600, one_workgroup_ok=False) == (16, 2) and sampler.plan_layout(16, 600, one_workgroup_ok=False) == (16, 1)
    assert sampler.plan_layout(12, 600, one_workgroup_ok=False) == (16, 1) and sampler.plan_layout(4, 1500, one_workgroup_ok=False) == (32, 2)
    with pytest.raises(sampler.PotusError):
        sampler.plan_layout(20, 600, one_workgroup_ok=False)                       # 600 days need 16 members: 20 x 16 do not fit
    assert sampler.plan_layout(40, 300, one_workgroup_ok=True) == (1, 2)          # four members hold 256 days at most: one workgroup each
    with pytest.raises(sampler.PotusError):
        sampler.plan_layout(40, 600, one_workgroup_ok=False)                       # 40 chains x 8 compute units do not fit
    with pytest.raises(sampler.PotusError):
        sampler.plan_layout(2, 600, cus_per_chain=1, one_workgroup_ok=False)


def test_bench_refuses_more_gpus_than_the_box_has():
    """`python bench.py --gpus N` with fewer than N visible GPUs exits non-zero and prints NO JSON line (here: no GPU at all; on a GPU box:
    fewer than 64) -- it must never fall back to an n_gpus = 1 line, which a scaling run would record as an N-GPU number."""
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "POTUS_DIST_BACKEND")}
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "64", "--steps", "2", "--warmup", "0"], capture_output=True, text=True, env=env, timeout=600, cwd=str(ROOT))
    assert out.returncode != 0 and "refusing to run" in out.stderr
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    # a launcher that started a different number of ranks than --gpus says is an error too, even a single rank
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2"], capture_output=True, text=True, timeout=600, cwd=str(ROOT),
                         env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert out.returncode != 0 and "WORLD_SIZE=1" in (out.stderr + out.stdout)
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]


def test_side_measurements_keep_the_line_alive(monkeypatch):
    """bench.py's side measurements (configs[0], configs[3], the configs[4] preset as child processes behind the default line) only ever run on the
    driver's GPU box; here their bookkeeping with the children faked: a good child's line is kept in compact form, a child that fails, prints
    garbage or times out costs its own entry only, and a spent time budget skips the rest -- the default line is printed whatever happens."""
    import importlib.util
    import json
    import subprocess
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("bench_mod2", ROOT / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    good = json.loads([ln for ln in (ROOT / "profiles" / "r05_bench_config4.json").read_text().splitlines() if ln.startswith("{")][0])
    head = json.loads([ln for ln in (ROOT / "profiles" / "r05_bench_line.json").read_text().splitlines() if ln.startswith("{")][0])
    head["config"]["posteriors"]["2016"]["treedepth_max_per_chain"] = [8] * 8
    calls = []

    def fake_run(cmd, **kw):
        cfg = cmd[cmd.index("--config") + 1]
        calls.append(cfg)
        assert "--no-side" in cmd and "--no-cpu-baseline" in cmd and "WORLD_SIZE" not in kw["env"] and cmd.count("--seed") == 1
        if cfg == "1":                                   # the headline configuration under another seed: a slower re-roll for the first, garbage for the second
            sd = int(cmd[cmd.index("--seed") + 1])
            if sd == 1845:
                return subprocess.CompletedProcess(cmd, 0, stdout="{not json\n", stderr="")
            slow = json.loads(json.dumps(head))
            slow["value"], slow["seconds"] = head["value"] * 0.8, head["seconds"] / 0.8
            slow["config"]["posteriors"]["2016"]["treedepth_max_per_chain"] = [8] * 7 + [9]
            return subprocess.CompletedProcess(cmd, 0, stdout=json.dumps(slow) + "\n", stderr="")
        if cfg == "0":
            return subprocess.CompletedProcess(cmd, 1, stdout="", stderr="boom")
        if cfg == "3":
            raise subprocess.TimeoutExpired(cmd, kw["timeout"])
        return subprocess.CompletedProcess(cmd, 0, stdout="noise\n" + json.dumps(good) + "\n", stderr="")

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setenv("WORLD_SIZE", "1")
    out = bench.side_measurements(1843, headline=bench.seed_summary({**head, "seed": 1843}))
    assert calls == ["0", "3", "1", "1", "4", "4", "4"] and list(out)[0] == "seeds" and out["configs[4]_pooled"]["value"] == good["value"]
    assert out["configs[4]_pooled_f32"]["value"] == good["value"]
    sd = out["seeds"]                                    # the default line's own run first; the child that printed garbage costs its own entry only
    assert [r["seed"] for r in sd["runs"]] == [1843, 1844] and "error" in out["seed_1845"] and sd["deepest_tree_by_seed"] == [8, 9]
    assert sd["leapfrogs_per_sec"]["max"] == head["value"] and abs(sd["leapfrogs_per_sec"]["min"] - 0.8 * head["value"]) < 1e-9
    assert abs(sd["seconds"]["median"] - 0.5 * (head["seconds"] + head["seconds"] / 0.8)) < 1e-9
    assert out["configs[0]"]["rc"] == 1 and "boom" in out["configs[0]"]["error"] and "timed out" in out["configs[3]"]["error"]
    e = out["configs[4]_preset"]
    assert e["baseline_config_index"] == 4 and e["value"] == good["value"] and e["roofline"]["kernel"] == "k_dn_symv" and e["max_depth"] == 7
    assert abs(e["roofline"]["frac"] - good["roofline"]["frac"]) < 1e-15 and "per_step" not in e["dense"] and e["dense"]["window_ends"] == 1
    calls.clear()
    out = bench.side_measurements(1843, budget_s=10.0)                    # nothing fits a budget of ten seconds
    assert calls == [] and all("skipped" in out[k] for k in ("configs[0]", "configs[3]", "seed_1844", "seed_1845", "configs[4]_preset", "configs[4]_pooled",
                                                         "configs[4]_pooled_f32")) and "seeds" not in out
