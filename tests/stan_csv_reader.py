"""A strict reader of CmdStan CSV files: TEST INFRASTRUCTURE standing in for rstan::read_stan_csv (final_2016.R:543).

R is not installed here, so no file this repo writes has ever been parsed by rstan itself.  This module restates, as
assertions, what rstan 2.21's read_stan_csv / read_csv_header / parse_stancsv_comments / paridx_fun / get_dims_from_fnames /
get_time_from_csv (rstan/R/stan_csv.R, rstan/R/misc.R -- third-party, not in the reference tree; restated from the published
source) require of a file before they yield a stanfit, plus the stricter grammar of CmdStan's own adaptation block that
cmdstanr::read_cmdstan_csv parses (step size as a number, one numeric comment line for diag_e, D lines of D values for dense_e).
What it cannot check: R's own number parser (scan) and the S4 object construction.

    fit = read_stan_csv([paths])      ->  StanCsvFit(chains, n_save, fnames, sampler_params, pars, dims, samples, comments ...)
"""
from __future__ import annotations

import math
import re
from dataclasses import dataclass, field

import numpy as np


class StanCsvError(ValueError):
    pass


@dataclass
class ChainCsv:
    header: list
    rows: np.ndarray
    comments: list
    values: dict
    adaptation_info: str
    time_info: list
    stepsize: float | None = None
    inv_metric: np.ndarray | None = None
    elapsed: tuple | None = None


@dataclass
class StanCsvFit:
    chains: list
    fnames: list
    sampler_params: list
    pars: list
    dims: dict
    n_save: int
    n_kept: int
    warmup2: int
    model_name: str
    samples: np.ndarray = field(repr=False, default=None)      # [chain][iteration][column]

    def extract(self, par, inc_warmup=False):
        """rstan::extract(fit, pars = par)[[1]] with chains merged (not permuted): [draws, *dims]."""
        cols = [i for i, n in enumerate(self.fnames) if n == par or n.startswith(par + ".")]
        x = self.samples[:, (0 if inc_warmup else self.warmup2):, :][:, :, cols]
        x = x.reshape(-1, len(cols))
        d = self.dims[par]
        return x[:, 0] if not d else x.reshape((x.shape[0],) + tuple(reversed(d))).transpose((0,) + tuple(range(len(d), 0, -1)))


_NUM = re.compile(r"^[+-]?(\d+\.?\d*|\.\d+)([eE][+-]?\d+)?$|^[+-]?(inf|nan)$", re.I)


def _read_one(path):
    """read_csv_header + the readBin loop of read_stan_csv: a line is a comment ('#'), the header (starts with 'l': lp__), empty, or
    a row of exactly len(header) numbers."""
    header, rows, comments = None, [], []
    with open(path, "rb") as f:
        raw = f.read()
    if b"\r" in raw:
        raise StanCsvError(f"{path}: carriage returns (scan() would read them into the last field)")
    if not raw.endswith(b"\n"):
        raise StanCsvError(f"{path}: the last line is not terminated")
    for ln_no, ln in enumerate(raw.decode("ascii").split("\n")[:-1], 1):
        if ln.startswith("#"):
            comments.append(ln)
            continue
        if ln.startswith("l"):                                   # char == 108: the header (also when CmdStan repeats it)
            names = ln.split(",")
            if header is None:
                header = names
            elif names != header:
                raise StanCsvError(f"{path}:{ln_no}: a second header line that differs from the first")
            continue
        if ln == "":
            continue
        if header is None:
            raise StanCsvError(f"{path}:{ln_no}: data before the header line (read_csv_header takes the first non-comment line as the header)")
        cells = ln.split(",")
        if len(cells) != len(header):
            raise StanCsvError(f"{path}:{ln_no}: {len(cells)} fields, header has {len(header)} (scan into the row buffer fails)")
        for c in cells:
            if not _NUM.match(c):
                raise StanCsvError(f"{path}:{ln_no}: field {c!r} is not a number")
        rows.append([float(c) for c in cells])
    if header is None:
        raise StanCsvError(f"{path}: no header line")
    if header[0] != "lp__":
        raise StanCsvError(f"{path}: the header does not start with lp__")
    if len(set(header)) != len(header):
        raise StanCsvError(f"{path}: duplicate column names")
    if any("#" in h for h in header):
        raise StanCsvError(f"{path}: '#' inside the header (read_csv_header greps for it anywhere in the line)")
    return header, np.asarray(rows, dtype=np.float64).reshape(len(rows), len(header)), comments


def _parse_comments(path, comments):
    """parse_stancsv_comments."""
    adapt = [i for i, c in enumerate(comments) if "Adaptation terminated" in c]
    tline = [i for i, c in enumerate(comments) if "Elapsed Time" in c]
    if len(adapt) > 1 or len(tline) > 1:
        raise StanCsvError(f"{path}: more than one 'Adaptation terminated' / 'Elapsed Time' line")
    n = len(comments)
    a = adapt[0] if adapt else n
    if not tline:
        raise StanCsvError(f"{path}: line with \"Elapsed Time\" not found (rstan warns; CmdStan always writes it)")
    if adapt and not a < tline[0]:
        raise StanCsvError(f"{path}: 'Elapsed Time' precedes 'Adaptation terminated'")
    adaptation_info = "\n".join(comments[a + 1:tline[0]]) if adapt else ""
    time_info = comments[tline[0]:]
    values = {}
    for c in comments[:a]:
        if "=" not in c:
            continue
        c = re.sub(r"^#+\s*|\s*|\(Default\)", "", c)
        k, v = c.split("=", 1)
        values[{"id": "chain_id", "num_warmup": "warmup", "num_samples": "iter"}.get(k, k)] = v      # later keys overwrite earlier ones, as in R's list
    for k in ("thin", "iter", "warmup", "chain_id", "save_warmup"):
        if k not in values:
            raise StanCsvError(f"{path}: comment '{k}' missing")
        if not re.match(r"^-?\d+$", values[k]):
            raise StanCsvError(f"{path}: {k} = {values[k]!r} is not an integer")
        values[k] = int(values[k])
    for k in ("max_depth",):
        if k in values:
            values[k] = int(values[k])
    for k in ("stepsize", "stepsize_jitter", "gamma", "kappa", "delta", "t0"):
        if k in values:
            values[k] = float(values[k])
    values["iter"] += values["warmup"]
    if values.get("algorithm") != "hmc" or values.get("engine") != "nuts" or values.get("metric") not in ("diag_e", "dense_e", "unit_e"):
        raise StanCsvError(f"{path}: algorithm / engine / metric = {values.get('algorithm')} / {values.get('engine')} / {values.get('metric')}: no sampler_t")
    values["sampler_t"] = f"NUTS({values['metric']})"
    return values, adaptation_info, time_info


def _floats(path, line):
    try:
        return [float(x) for x in line.lstrip("# ").split(",")]
    except ValueError:
        raise StanCsvError(f"{path}: a line of the adaptation block is not a list of numbers: {line[:60]!r}") from None


def _adaptation(path, info, metric, D):
    """CmdStan's block, as cmdstanr::read_cmdstan_csv reads it."""
    lines = info.split("\n") if info else []
    if not lines or not lines[0].startswith("# Step size = "):
        raise StanCsvError(f"{path}: '# Step size = ' does not follow 'Adaptation terminated'")
    eps = _floats(path, lines[0][len("# Step size = "):])[0]
    if not (eps > 0 and math.isfinite(eps)):
        raise StanCsvError(f"{path}: step size {eps}")
    body = [ln for ln in lines[1:] if ln.strip() not in ("#", "")]
    if metric == "diag_e":
        if len(body) != 2 or body[0] != "# Diagonal elements of inverse mass matrix:":
            raise StanCsvError(f"{path}: diag_e adaptation block is not 'Diagonal elements of inverse mass matrix:' + one line")
        v = _floats(path, body[1])
        if len(v) != D or not all(x > 0 for x in v):
            raise StanCsvError(f"{path}: {len(v)} inverse metric elements for {D} parameters (or a non-positive one)")
        return eps, np.asarray(v)
    if metric == "dense_e":
        if body and body[0] == "# Elements of inverse mass matrix:":
            if len(body) != 1 + D:
                raise StanCsvError(f"{path}: dense_e adaptation block has {len(body) - 1} matrix lines for D = {D}")
            M = np.asarray([_floats(path, ln) for ln in body[1:]])
            if M.shape != (D, D) or not np.allclose(M, M.T, rtol=1e-5, atol=1e-12):
                raise StanCsvError(f"{path}: inverse metric is not a symmetric {D} x {D} matrix")
            return eps, M
        if any("Elements of inverse mass matrix" in ln for ln in body):
            raise StanCsvError(f"{path}: matrix header without its D rows")
        return eps, None                                   # the matrix left out (large D): no header line either
    return eps, None


def _elapsed(path, tl):
    """get_time_from_csv: the first two lines of the block carry the warm-up and the sampling seconds."""
    tl = [t for t in tl if t.strip() != "#"]
    if len(tl) != 3:
        raise StanCsvError(f"{path}: the Elapsed Time block has {len(tl)} lines, not 3")
    w = re.sub(r"\s*seconds.*$", "", re.sub(r".*#\s*Elapsed.*:\s*", "", tl[0]))
    s = re.sub(r"\s*seconds.*$", "", re.sub(r".*#\s*", "", tl[1]))
    t = re.sub(r"\s*seconds.*$", "", re.sub(r".*#\s*", "", tl[2]))
    if "(Warm-up)" not in tl[0] or "(Sampling)" not in tl[1] or "(Total)" not in tl[2]:
        raise StanCsvError(f"{path}: Elapsed Time lines are not (Warm-up) / (Sampling) / (Total)")
    return float(w), float(s), float(t)


def _dims(path, fnames):
    """paridx_fun + unique_par + get_dims_from_fnames: names ending in '__' are sampler parameters; `par.i.j` columns of one
    parameter must form a complete column-major grid."""
    sp = [n for n in fnames if n.endswith("__") and n != "lp__"]
    pars, cols = [], {}
    for n in fnames:
        if n.endswith("__"):
            continue
        base = n.split(".")[0]
        if base not in cols:
            pars.append(base); cols[base] = []
        elif cols[base] is not None and fnames[fnames.index(n) - 1].split(".")[0] != base:
            raise StanCsvError(f"{path}: the columns of {base} are not contiguous")
        cols[base].append(n)
    dims = {}
    for p in pars:
        idx = [tuple(int(k) for k in n.split(".")[1:]) for n in cols[p]]
        if idx == [()]:
            dims[p] = ()
            continue
        nd = len(idx[0])
        if any(len(i) != nd or nd == 0 for i in idx):
            raise StanCsvError(f"{path}: mixed index depth in {p}")
        d = tuple(max(i[k] for i in idx) for k in range(nd))
        want = [tuple(reversed(t)) for t in np.ndindex(*reversed(d))]          # first index fastest
        want = [tuple(k + 1 for k in t) for t in want]
        if idx != want:
            raise StanCsvError(f"{path}: the columns of {p} are not the column-major grid 1..{d}")
        dims[p] = d
    dims["lp__"] = ()
    return sp, pars + ["lp__"], dims


def read_stan_csv(csvfiles):
    chains = []
    for path in csvfiles:
        header, rows, comments = _read_one(path)
        values, info, tl = _parse_comments(path, comments)
        chains.append(ChainCsv(header, rows, comments, values, info, tl))
    f0 = chains[0]
    for c in chains[1:]:
        if c.header != f0.header:
            raise StanCsvError("the CSV files do not have same parameters")
        if c.rows.shape[0] != f0.rows.shape[0]:
            raise StanCsvError("the number of iterations are not the same in all CSV files")
        for k in ("warmup", "thin", "iter", "save_warmup"):
            if c.values[k] != f0.values[k]:
                raise StanCsvError("not all iter/warmups/thin are the same in all CSV files")
    if len({c.values["chain_id"] for c in chains}) != len(chains):
        raise StanCsvError("chain ids are not distinct")
    sp, pars, dims = _dims(csvfiles[0], f0.header)
    want_sp = ["accept_stat__", "stepsize__", "treedepth__", "n_leapfrog__", "divergent__", "energy__"]
    if sp != want_sp:
        raise StanCsvError(f"sampler parameters {sp}, NUTS writes {want_sp}")
    v = f0.values
    n_save = f0.rows.shape[0]
    n_kept0 = 1 + (v["iter"] - v["warmup"] - 1) // v["thin"]
    warmup2 = (1 + (v["warmup"] - 1) // v["thin"]) if v["save_warmup"] == 1 and v["warmup"] > 0 else 0
    n_kept = n_save - warmup2
    if n_kept0 != n_kept:
        raise StanCsvError(f"the number of iterations after warmup found ({n_kept}) does not match iter/warmup/thin from CSV comments ({n_kept0})")
    for path, c in zip(csvfiles, chains):
        c.stepsize, c.inv_metric = _adaptation(path, c.adaptation_info, c.values["metric"], _n_unconstrained(c.header, dims, c))
        c.elapsed = _elapsed(path, c.time_info)
        if not np.isfinite(c.rows).all():
            raise StanCsvError(f"{path}: non-finite draws")
        samp = c.rows[warmup2:]
        if samp.size and not np.allclose(samp[:, 2], c.stepsize, rtol=1e-5):
            raise StanCsvError(f"{path}: stepsize__ of the sampling rows differs from '# Step size ='")
        if not (np.all(samp[:, 3] == np.floor(samp[:, 3])) and np.all(samp[:, 4] >= 1) and set(np.unique(c.rows[:, 5])) <= {0.0, 1.0}):
            raise StanCsvError(f"{path}: treedepth__ / n_leapfrog__ / divergent__ are not what NUTS writes")
    name = re.sub(r"(_\d+)*$", "", re.sub(r"\.csv$", "", csvfiles[0].split("/")[-1]))
    name = re.sub(r"-\d+$", "", name)
    return StanCsvFit(chains, list(f0.header), sp, pars, dims, n_save, n_kept, warmup2, name, np.stack([c.rows for c in chains]))


def _n_unconstrained(header, dims, c):
    """Length the inverse metric must have: CmdStan writes one element per UNCONSTRAINED parameter; for this model every
    parameter block has as many unconstrained as constrained elements, and the parameters block ends where the transformed
    parameters begin (mu_b)."""
    n = 0
    for h in header:
        if h.endswith("__"):
            continue
        if h.split(".")[0] == "mu_b":
            break
        n += 1
    return n
