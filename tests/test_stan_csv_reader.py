"""The strict Stan-CSV reader (tests/stan_csv_reader.py, the stand-in for rstan::read_stan_csv) on hand-made files: it accepts
CmdStan 2.24's layout and rejects each way a writer can get it wrong.  The files libpotus_hmc writes are read by it on the GPU
box (tests/test_gpu_parity.py)."""
import numpy as np
import pytest

from stan_csv_reader import StanCsvError, read_stan_csv


def _csv(chain=1, nw=4, ns=3, save_warmup=0, metric="diag_e", matrix=True, seed=0):
    rng = np.random.default_rng(seed + chain)
    names = ["lp__", "accept_stat__", "stepsize__", "treedepth__", "n_leapfrog__", "divergent__", "energy__",
             "a.1", "a.2", "Z.1.1", "Z.2.1", "Z.1.2", "Z.2.2", "rho", "mu_b.1", "mu_b.2"]
    D = 7
    out = ["# stan_version_major = 2", "# stan_version_minor = 24", "# stan_version_patch = 1", "# model = toy_model", "# method = sample (Default)",
           "#   sample", f"#     num_samples = {ns}", f"#     num_warmup = {nw}", f"#     save_warmup = {save_warmup}", "#     thin = 1 (Default)",
           "#     adapt", "#       engaged = 1 (Default)", "#       gamma = 0.05", "#       delta = 0.8", "#     algorithm = hmc (Default)", "#       hmc",
           "#         engine = nuts (Default)", "#           nuts", "#             max_depth = 10",
           f"#         metric = {metric}" + (" (Default)" if metric == "diag_e" else ""), "#         stepsize = 1", f"# id = {chain}", "# random", "#   seed = 1843",
           ",".join(names)]
    adapt = ["# Adaptation terminated", "# Step size = 0.25"]
    if metric == "diag_e":
        adapt += ["# Diagonal elements of inverse mass matrix:", "# " + ", ".join(f"{x:.6g}" for x in rng.uniform(0.5, 2, D))]
    elif matrix:
        A = rng.standard_normal((D, D)); M = A @ A.T + np.eye(D)
        adapt += ["# Elements of inverse mass matrix:"] + ["# " + ", ".join(f"{x:.6g}" for x in r) for r in M]
    def row(eps):
        v = [-10.5, 0.9, eps, 3, 7, 0, 12.25] + list(rng.standard_normal(9))
        return ",".join(f"{x:.6g}" for x in v)
    if save_warmup:
        out += [row(0.1 + 0.01 * k) for k in range(nw)] + adapt
    else:
        out += adapt
    out += [row(0.25) for _ in range(ns)]
    out += ["# ", "#  Elapsed Time: 0.125 seconds (Warm-up)", "#                0.250 seconds (Sampling)", "#                0.375 seconds (Total)", "# "]
    return "\n".join(out) + "\n"


def _write(tmp_path, texts):
    paths = []
    for i, t in enumerate(texts):
        p = tmp_path / f"toy-{i + 1}.csv"
        p.write_text(t)
        paths.append(str(p))
    return paths


@pytest.mark.parametrize("kw", [dict(), dict(save_warmup=1), dict(metric="dense_e"), dict(metric="dense_e", matrix=False), dict(metric="dense_e", save_warmup=1)])
def test_reader_accepts_cmdstan_layout(tmp_path, kw):
    fit = read_stan_csv(_write(tmp_path, [_csv(1, **kw), _csv(2, **kw)]))
    assert fit.n_kept == 3 and fit.warmup2 == (4 if kw.get("save_warmup") else 0) and fit.model_name == "toy"
    assert fit.dims["Z"] == (2, 2) and fit.dims["a"] == (2,) and fit.dims["rho"] == ()
    assert fit.extract("Z").shape == (6, 2, 2) and fit.extract("lp__").shape == (6,)
    z = fit.extract("Z")
    assert z[0, 1, 0] == fit.samples[0, fit.warmup2, fit.fnames.index("Z.2.1")]           # column-major names
    assert fit.chains[0].stepsize == 0.25 and fit.chains[0].elapsed == (0.125, 0.25, 0.375)
    if kw.get("metric") == "dense_e":
        assert (fit.chains[0].inv_metric is None) == (kw.get("matrix") is False)
    else:
        assert fit.chains[0].inv_metric.shape == (7,)


@pytest.mark.parametrize("name,mutate", [
    ("row with a missing field", lambda t: t.replace(",12.25,", ",", 1)),
    ("non-numeric field", lambda t: t.replace("-10.5,0.9", "-10.5,NA", 1)),
    ("num_samples does not match the rows", lambda t: t.replace("num_samples = 3", "num_samples = 5")),
    ("no elapsed time", lambda t: t[:t.index("#  Elapsed")] ),
    ("elapsed time block of two lines", lambda t: t.replace("#                0.375 seconds (Total)\n", "")),
    ("step size line missing", lambda t: t.replace("# Step size = 0.25\n", "")),
    ("metric header without values", lambda t: t.replace("# Diagonal elements of inverse mass matrix:\n# ", "# Diagonal elements of inverse mass matrix:\n# x")),
    ("wrong number of metric elements", lambda t: __import__("re").sub(r"(# Diagonal elements of inverse mass matrix:\n# [^\n]*)", r"\1, 1.0", t)),
    ("names not column-major", lambda t: t.replace("Z.2.1,Z.1.2", "Z.1.2,Z.2.1")),
    ("duplicate column", lambda t: t.replace("a.2", "a.1")),
    ("thin not an integer", lambda t: t.replace("thin = 1 (Default)", "thin = 1.5")),
    ("unknown metric", lambda t: t.replace("metric = diag_e (Default)", "metric = banana")),
    ("sampling rows do not carry the adapted step size", lambda t: t.replace(",0.25,3,7", ",0.5,3,7", 1)),
    ("carriage returns", lambda t: t.replace("\n", "\r\n")),
    ("free text where the matrix should be", None),
])
def test_reader_rejects_broken_files(tmp_path, name, mutate):
    if mutate is None:
        good = _csv(1, metric="dense_e")
        bad = good[:good.index("# Elements of inverse mass matrix:\n")] + "# Elements of inverse mass matrix:\n# (7 x 7 matrix omitted)\n" + \
            good[good.index("-10.5"):]
    else:
        good = _csv(1)
        bad = mutate(good)
    assert bad != good, name
    with pytest.raises(StanCsvError):
        read_stan_csv(_write(tmp_path, [bad]))


def test_reader_checks_the_files_against_each_other(tmp_path):
    with pytest.raises(StanCsvError, match="chain ids"):
        read_stan_csv(_write(tmp_path, [_csv(1), _csv(1)]))
    with pytest.raises(StanCsvError, match="same parameters"):
        read_stan_csv(_write(tmp_path, [_csv(1), _csv(2).replace("rho", "rho2")]))
    with pytest.raises(StanCsvError, match="iter/warmups/thin|number of iterations"):
        read_stan_csv(_write(tmp_path, [_csv(1), _csv(2, ns=4)]))
