"""CSV -> Stan `data` list for the poll model (host-side, not on the timed path).

Restates the data-preparation section of the reference run scripts so that
real-data fixtures can be produced without R:

  * polls -> indices/counts      scripts/model/final_2016.R:73-163
  * state weights                scripts/model/final_2016.R:165-193
  * state covariance             scripts/model/final_2016.R:196-332
  * scale constants              scripts/model/final_2016.R:334-342, 462-472
  * state priors                 scripts/model/final_2016.R:400-410
  * the `data` list itself       scripts/model/final_2016.R:435-514

  * 2012 / 2008 backtests        scripts/model/final_2012.R:63-556, final_2008.R:63-560
                                 (feed poll_model_2020_no_mode_adjustment.stan)

The output dict mirrors the `data{}` block of
scripts/model/poll_model_2020.stan:1-41 (same names, 1-based indices).

Known, harmless degrees of freedom (documented, not hidden):
  * factor level order of pollsters follows Python's code-point sort; R uses the
    session locale.  This only relabels pollsters (a permutation of raw_mu_c).
  * the order of polls inside the state/national vectors follows a code-point
    `arrange(state, t, polltype, two_party_sum)`; again only a relabelling of
    the per-poll noise parameters.
"""
from __future__ import annotations

import re
from pathlib import Path

import numpy as np
import pandas as pd

ADJUSTERS = ("ABC", "Washington Post", "Ipsos", "Pew", "YouGov", "NBC")  # final_2016.R:423-430


def _r_round(x):
    """R's round(): IEC 60559 half-to-even, same as numpy.rint."""
    return np.rint(x)


def cov_matrix(n: int, sigma2: float, rho: float) -> np.ndarray:
    """final_2016.R:29-35."""
    m = np.full((n, n), rho)
    np.fill_diagonal(m, 1.0)
    d = np.sqrt(sigma2) * np.eye(n)
    return d @ m @ d


def make_positive_definite(m: np.ndarray, tol: float | None = None) -> np.ndarray:
    """lqmm::make.positive.definite (third-party, unpinned; used at final_2016.R:286,295).

    Published algorithm: eigen-decompose, lift every eigenvalue below 2*tol up to
    2*tol, with tol = d * max|eig| * eps by default.
    """
    m = np.asarray(m, dtype=np.float64)
    d = m.shape[0]
    vals, vecs = np.linalg.eigh(m)
    if tol is None:
        tol = d * np.max(np.abs(vals)) * np.finfo(np.float64).eps
    delta = 2.0 * tol
    tau = np.maximum(0.0, delta - vals)
    return m + (vecs * tau) @ vecs.T


def logit(p):
    p = np.asarray(p, dtype=np.float64)
    return np.log(p / (1.0 - p))


def _extract_pollster(name: str) -> str:
    # str_extract(pollster, "[A-z0-9 ]+") then strip trailing blanks (final_2016.R:100)
    m = re.search(r"[A-z0-9 ]+", name)
    return re.sub(r"\s+$", "", m.group(0)) if m else name


def _state_context_2016(data_dir: Path):
    """final_2016.R:165-193."""
    st = pd.read_csv(data_dir / "2012.csv")
    st["score"] = st.obama_count / (st.obama_count + st.romney_count)
    st["national_score"] = st.obama_count.sum() / (st.obama_count + st.romney_count).sum()
    st["delta"] = st.score - st.national_score
    grown = st.total_count * (1.0 + st.adult_pop_growth_2011_15)
    st["share_national_vote"] = grown / grown.sum()
    st = st.sort_values("state", kind="stable").reset_index(drop=True)
    w = (st.share_national_vote / st.share_national_vote.sum()).to_numpy()
    return st, w


def state_correlation(data_dir: Path, dem_year: int = 2016) -> np.ndarray:
    """final_2016.R:196-295: demographic correlation, clipped and mixed."""
    res = pd.read_csv(data_dir / "potus_results_76_16.csv")
    res = res[res.year == dem_year][["state", "dem"]].dropna()
    long = [pd.DataFrame({"state": res.state, "variable": str(dem_year), "value": res.dem})]

    census = pd.read_csv(data_dir / "acs_2013_variables.csv")
    census = census[census.state.notna()].drop(columns=["state_fips", "pop_total", "pop_density"])
    long.append(census.melt(id_vars="state", var_name="variable", value_name="value"))

    urb = pd.read_csv(data_dir / "urbanicity_index.csv")[["state", "average_log_pop_within_5_miles"]]
    urb = urb.rename(columns={"average_log_pop_within_5_miles": "pop_density"})
    long.append(urb.melt(id_vars="state", var_name="variable", value_name="value"))

    ev = pd.read_csv(data_dir / "white_evangel_pct.csv")
    long.append(ev.melt(id_vars="state", var_name="variable", value_name="value"))

    sd = pd.concat(long, ignore_index=True)
    # min-max scale each variable across states (final_2016.R:251-256)
    g = sd.groupby("variable")["value"]
    sd["value"] = (sd.value - g.transform("min")) / (g.transform("max") - g.transform("min"))
    wide = sd.pivot(index="variable", columns="state", values="value").dropna()
    wide = wide[sorted(wide.columns)]
    C = np.corrcoef(wide.to_numpy(), rowvar=False)
    C[C < 0] = 0.0
    lam = 0.75
    new_C = make_positive_definite(lam * C + (1.0 - lam) * np.ones_like(C))
    return make_positive_definite(new_C)


def build_2016(data_dir: str | Path, run_date: str = "2016-11-08") -> dict:
    """The full `data` list for the 2016 backtest (final_2016.R:66-514)."""
    data_dir = Path(data_dir)
    RUN_DATE = pd.Timestamp(run_date)
    election_day = pd.Timestamp("2016-11-08")
    start_date = pd.Timestamp("2016-03-01")

    ap = pd.read_csv(data_dir / "all_polls.csv")
    ap = ap[["state", "pollster", "number.of.observations", "population", "mode", "start.date",
             "end.date", "clinton", "trump", "undecided", "other", "johnson", "mcmullin"]].copy()
    ap["end"] = pd.to_datetime(ap["end.date"])
    ap["begin"] = pd.to_datetime(ap["start.date"])
    ap = ap[ap.end <= RUN_DATE]

    df = ap.rename(columns={"number.of.observations": "n"})
    span = (df.end - df.begin).dt.days
    df["t"] = df.end - pd.to_timedelta((1 + span) // 2, unit="D")
    df = df[(df.t >= start_date) & df.t.notna()
            & df.population.isin(["Likely Voters", "Registered Voters", "Adults"])
            & (df.n > 1)].copy()

    df["pollster"] = df.pollster.map(_extract_pollster).replace({
        "Fox News": "FOX", "WashPost": "Washington Post", "ABC News": "ABC",
        "DHM Research": "DHM", "Public Opinion Strategies": "POS"})
    df["undecided"] = df.undecided.fillna(0)
    df["other"] = df.other.fillna(0) + df.johnson.fillna(0) + df.mcmullin.fillna(0)

    mode_l = df["mode"].astype(str).str.lower()
    df["mode"] = np.where(df["mode"] == "Internet", "Online poll",
                          np.where(mode_l.str.contains("live phone"), "Live phone component", "Other"))

    df["two_party_sum"] = df.clinton + df.trump
    df["polltype"] = df.population
    df["n_respondents"] = _r_round(df.n)
    df["n_clinton"] = _r_round(df.n * df.clinton / 100.0)
    df["n_trump"] = _r_round(df.n * df.trump / 100.0)

    state_abb_list = list(pd.read_csv(data_dir / "potus_results_76_16.csv").state.unique())
    levels = ["--"] + state_abb_list
    idx = df.state.map({s: i + 1 for i, s in enumerate(levels)})
    df["index_s"] = np.where(idx == 1, 52, idx - 1)
    tmin = df.t.min()
    df["poll_day"] = (df.t - tmin).dt.days + 1
    for col, src in (("index_p", "pollster"), ("index_m", "mode"), ("index_pop", "polltype")):
        lev = sorted(df[src].astype(str).unique())
        df[col] = df[src].astype(str).map({v: i + 1 for i, v in enumerate(lev)})

    df = df.sort_values(["state", "t", "polltype", "two_party_sum"], kind="stable")
    df = df.drop_duplicates(subset=["state", "t", "pollster"], keep="first").reset_index(drop=True)

    first_day = df.begin.min()
    T = int(round((election_day - first_day).days))
    pollsters = sorted(df.pollster.unique())

    st, state_weights = _state_context_2016(data_dir)
    states = list(st.state)
    assert states == sorted(states) and len(states) == 51

    corr = state_correlation(data_dir, 2016)
    state_covariance_0 = cov_matrix(51, 0.07 ** 2, 0.9) * corr

    days_til_election = (election_day - RUN_DATE).days
    expected_national_mu_b_T_error = 0.03 + (10 ** -6.6) * days_til_election ** 2
    polling_bias_scale = 0.013 * 4
    mu_b_T_scale = expected_national_mu_b_T_error * 4
    random_walk_scale = 0.05 / np.sqrt(300.0) * 4

    pri = pd.read_csv(data_dir / "state_priors_08_12_16.csv")
    pri["date"] = pd.to_datetime(pri["date"])
    pri = pri[pri.date <= RUN_DATE]
    pri = pri[pri.date == pri.groupby("state").date.transform("max")]
    pri = pri.sort_values("state", kind="stable")
    assert list(pri.state) == states, "prior/state order mismatch (final_2016.R:411)"
    mu_b_prior = logit(pri.pred.to_numpy())

    unadj = (~df.pollster.isin(ADJUSTERS)).astype(np.float64)
    nat = (df.index_s == 52).to_numpy()
    sta = ~nat
    i32 = lambda s: np.asarray(s, dtype=np.int32)

    data = dict(
        N_national_polls=int(nat.sum()), N_state_polls=int(sta.sum()),
        T=T, S=51, P=len(pollsters), M=int(df["mode"].nunique()), Pop=int(df.polltype.nunique()),
        state=i32(df.index_s[sta]), state_weights=state_weights,
        day_state=i32(df.poll_day[sta]), day_national=i32(df.poll_day[nat]),
        poll_state=i32(df.index_p[sta]), poll_national=i32(df.index_p[nat]),
        poll_mode_national=i32(df.index_m[nat]), poll_mode_state=i32(df.index_m[sta]),
        poll_pop_national=i32(df.index_pop[nat]), poll_pop_state=i32(df.index_pop[sta]),
        unadjusted_national=unadj[nat].to_numpy(), unadjusted_state=unadj[sta].to_numpy(),
        n_democrat_national=i32(df.n_clinton[nat]), n_democrat_state=i32(df.n_clinton[sta]),
        n_two_share_national=i32((df.n_trump + df.n_clinton)[nat]),
        n_two_share_state=i32((df.n_trump + df.n_clinton)[sta]),
        sigma_measure_noise_national=0.04, sigma_measure_noise_state=0.04,
        mu_b_prior=mu_b_prior, sigma_c=0.06, sigma_m=0.04, sigma_pop=0.04, sigma_e_bias=0.02,
        state_covariance_0=state_covariance_0,
        polling_bias_scale=float(polling_bias_scale), mu_b_T_scale=float(mu_b_T_scale),
        random_walk_scale=float(random_walk_scale),
    )
    meta = dict(states=states, pollsters=pollsters, ev_state=st.ev.to_numpy(),
                first_day=str(first_day.date()), election_day=str(election_day.date()))
    return dict(data=data, meta=meta)


def _state_context_2008(data_dir: Path):
    """final_2012.R:171-193 and final_2008.R:175-197 (both read data/2008.csv, a CR-terminated file)."""
    st = pd.read_csv(data_dir / "2008.csv", lineterminator="\r") if b"\n" not in (data_dir / "2008.csv").read_bytes() \
        else pd.read_csv(data_dir / "2008.csv")
    st = st[st.state.notna()].copy()
    st["state"] = st.state.astype(str).str.strip()
    st["score"] = st.obama_count / (st.obama_count + st.mccain_count)
    st["national_score"] = st.obama_count.sum() / (st.obama_count + st.mccain_count).sum()
    st["delta"] = st.score - st.national_score
    st["share_national_vote"] = st.total_count / st.total_count.sum()
    st = st.sort_values("state", kind="stable").reset_index(drop=True)
    w = (st.share_national_vote / st.share_national_vote.sum()).to_numpy()
    return st, w


_BACKTESTS = {
    # year: (polls file, Republican column, RUN_DATE = election_day as scripted, start_date)
    2012: ("all_polls_2012.csv", "romney", "2012-11-06", "2012-03-01"),   # final_2012.R:63-69,75
    2008: ("all_polls_2008.csv", "mccain", "2008-11-03", "2008-03-01"),   # final_2008.R:63-69,75
}


def build_backtest(data_dir: str | Path, year: int, run_date: str | None = None) -> dict:
    """The `data` list of the 2012 or 2008 backtest (final_2012.R:63-556 / final_2008.R:63-560), which the
    scripts feed to poll_model_2020_no_mode_adjustment.stan (final_2012.R:558).  Differences from 2016: HuffPost
    CSV with m/d/y dates (2008: UTF-8 BOM), no population filter, population recoded to 0/1/2 for the
    de-duplication order only, three pollster renames, state context from the 2008 results without population
    growth, no mode/population indices; the list carries an unused `sigma_a` that Stan ignores."""
    if year not in _BACKTESTS:
        raise ValueError("year must be 2012 or 2008 (2016: build_2016)")
    data_dir = Path(data_dir)
    fname, rep, eday, sday = _BACKTESTS[year]
    election_day = pd.Timestamp(eday)
    RUN_DATE = pd.Timestamp(run_date) if run_date else election_day
    start_date = pd.Timestamp(sday)

    ap = pd.read_csv(data_dir / fname, encoding="utf-8-sig")
    ap = ap[["state", "pollster", "number.of.observations", "mode", "population", "start.date", "end.date",
             "obama", rep, "undecided", "other"]].copy()
    ap["end"] = pd.to_datetime(ap["end.date"], format="%m/%d/%y")
    ap["begin"] = pd.to_datetime(ap["start.date"], format="%m/%d/%y")
    ap = ap[ap.end <= RUN_DATE]

    df = ap.rename(columns={"number.of.observations": "n"})
    span = (df.end - df.begin).dt.days
    df["t"] = df.end - pd.to_timedelta((1 + span) // 2, unit="D")
    df = df[(df.t >= start_date) & df.t.notna() & (df.n > 1)].copy()

    df["pollster"] = df.pollster.map(_extract_pollster).replace({"Fox News": "FOX", "WashPost": "Washington Post", "ABC News": "ABC"})
    df["undecided"] = df.undecided.fillna(0)
    df["other"] = df.other.fillna(0)
    df["two_party_sum"] = df.obama + df[rep]
    df["polltype"] = df.population.map({"Likely Voters": 0.0, "Registered Voters": 1.0, "Adults": 2.0})   # others NA (sorted last, as arrange does)
    df["n_dem"] = _r_round(df.n * df.obama / 100.0)
    df["n_rep"] = _r_round(df.n * df[rep] / 100.0)

    state_abb_list = list(pd.read_csv(data_dir / "potus_results_76_16.csv").state.unique())
    levels = ["--"] + state_abb_list
    idx = df.state.map({s: i + 1 for i, s in enumerate(levels)})
    df["index_s"] = np.where(idx == 1, 52, idx - 1)
    tmin = df.t.min()
    df["poll_day"] = (df.t - tmin).dt.days + 1
    lev = sorted(df.pollster.astype(str).unique())
    df["index_p"] = df.pollster.astype(str).map({v: i + 1 for i, v in enumerate(lev)})

    df = df.sort_values(["state", "t", "polltype", "two_party_sum"], kind="stable", na_position="last")
    df = df.drop_duplicates(subset=["state", "t", "pollster"], keep="first").reset_index(drop=True)

    first_day = df.begin.min()
    T = int(round((election_day - first_day).days))
    pollsters = sorted(df.pollster.unique())

    st, state_weights = _state_context_2008(data_dir)
    states = list(st.state)
    assert states == sorted(states) and len(states) == 51

    corr = state_correlation(data_dir, 2016)               # final_2012.R:212: the 2016 results column, as scripted
    state_covariance_0 = cov_matrix(51, 0.07 ** 2, 0.9) * corr

    days_til_election = (election_day - RUN_DATE).days
    expected_national_mu_b_T_error = 0.03 + (10 ** -6.6) * days_til_election ** 2
    polling_bias_scale = 0.013 * 4
    mu_b_T_scale = expected_national_mu_b_T_error * 4
    random_walk_scale = 0.05 / np.sqrt(300.0) * 4

    pri = pd.read_csv(data_dir / "state_priors_08_12_16.csv")
    pri["date"] = pd.to_datetime(pri["date"])
    pri = pri[pri.date <= RUN_DATE]
    pri = pri[pri.date == pri.groupby("state").date.transform("max")]
    pri = pri.sort_values("state", kind="stable")
    assert list(pri.state) == states, "prior/state order mismatch (final_2012.R:460)"
    mu_b_prior = logit(pri.pred.to_numpy())

    unadj = (~df.pollster.isin(ADJUSTERS)).astype(np.float64)
    nat = (df.index_s == 52).to_numpy()
    sta = ~nat
    i32 = lambda s: np.asarray(s, dtype=np.int32)
    data = dict(
        N_national_polls=int(nat.sum()), N_state_polls=int(sta.sum()),
        T=T, S=51, P=len(pollsters), M=int(df["mode"].nunique(dropna=False)), Pop=int(df.polltype.nunique(dropna=False)),
        state=i32(df.index_s[sta]), state_weights=state_weights,
        day_state=i32(df.poll_day[sta]), day_national=i32(df.poll_day[nat]),
        poll_state=i32(df.index_p[sta]), poll_national=i32(df.index_p[nat]),
        unadjusted_national=unadj[nat].to_numpy(), unadjusted_state=unadj[sta].to_numpy(),
        n_democrat_national=i32(df.n_dem[nat]), n_democrat_state=i32(df.n_dem[sta]),
        n_two_share_national=i32((df.n_rep + df.n_dem)[nat]), n_two_share_state=i32((df.n_rep + df.n_dem)[sta]),
        sigma_a=0.012,                                        # carried by the script, not declared by the Stan model
        sigma_measure_noise_national=0.04, sigma_measure_noise_state=0.04,
        mu_b_prior=mu_b_prior, sigma_c=0.06, sigma_m=0.04, sigma_pop=0.04, sigma_e_bias=0.02,
        state_covariance_0=state_covariance_0,
        polling_bias_scale=float(polling_bias_scale), mu_b_T_scale=float(mu_b_T_scale),
        random_walk_scale=float(random_walk_scale),
    )
    meta = dict(states=states, pollsters=pollsters, ev_state=st.ev.to_numpy(),
                first_day=str(first_day.date()), election_day=str(election_day.date()))
    return dict(data=data, meta=meta)


def build_2012(data_dir, run_date=None):
    return build_backtest(data_dir, 2012, run_date)


def build_2008(data_dir, run_date=None):
    return build_backtest(data_dir, 2008, run_date)


def save_npz(path: str | Path, built: dict) -> None:
    """Store a `data` list (+ meta) as one .npz fixture."""
    flat = {f"data__{k}": np.asarray(v) for k, v in built["data"].items()}
    flat.update({f"meta__{k}": np.asarray(v) for k, v in built["meta"].items()})
    np.savez_compressed(path, **flat)


def load_npz(path: str | Path) -> dict:
    z = np.load(path, allow_pickle=False)
    data, meta = {}, {}
    for k in z.files:
        kind, name = k.split("__", 1)
        v = z[k]
        if v.ndim == 0:
            v = v.item()
        (data if kind == "data" else meta)[name] = v
    return dict(data=data, meta=meta)
