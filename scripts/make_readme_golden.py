#!/usr/bin/env python3
"""Turn everything the reference PUBLISHED about its posteriors into committed fixtures.

The reference has no tests; the only numbers it publishes for this path are the rendered tables of
/root/reference/README.md (knitted by README.Rmd from saved fits that are not in the tree):

  README.md:83-136, 179-232, 279-332   election-day predicted_score per state + national ('--', rendered
                                        as an en dash): mean, 2.5 %, 97.5 %, P(> 0.5), se = (high - mean) / 1.96
  README.md:75, 169, 260               'economist (backtest)': EV-weighted Brier, unweighted Brier, states correct
  README.md:79, 175, 275               RMSE over the 50 states (DC left out) of the election-day posterior mean against the
                                        result, actual = dem / (dem + rep) (README.Rmd:392-401, 908-917, 1431-1440: the
                                        results come from politicaldata::pres_results, whose table the reference also
                                        commits as data/potus_results_76_16.csv) -- the only published figures with
                                        seven significant digits

plus the winner lists the Brier scores are computed against: README.Rmd:381, 892, 1416 (what the published
numbers used) and scripts/model/final_2008.R:926-927, final_2012.R:922-923, final_2016.R:929 (the run
scripts' own copies; the 2008 script carries the 2012 list).

Output: tests/golden/readme_{2008,2012,2016}.csv -- one row per state and one for the nation, columns
  state,mean,low,high,prob,se,won_readme,won_script,actual
and `# key = value` comment lines with the three performance figures and the RMSE.  Run in the build container only
(needs /root/reference); the GPU box reads the committed CSVs.
"""
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
REF = Path(sys.argv[1]) if len(sys.argv) > 1 else Path("/root/reference")
GOLD = ROOT / "tests" / "golden"
YEARS = (2008, 2012, 2016)


def state_tables(lines):
    """The three 'Predictions for each state' tables, in file order (2008, 2012, 2016)."""
    tables, i = [], 0
    while i < len(lines):
        if lines[i].startswith("| state |  mean |"):
            first, rows = i + 1, []
            i += 2
            while i < len(lines) and lines[i].startswith("|"):
                cells = [c.strip() for c in lines[i].strip().strip("|").split("|")]
                rows.append((cells[0], *[float(c) for c in cells[1:6]]))
                i += 1
            tables.append((first, i, rows))
        i += 1
    return tables


def performance(lines):
    out = []
    for i, ln in enumerate(lines):
        if ln.startswith("| economist (backtest)"):
            c = [x.strip() for x in ln.strip().strip("|").split("|")]
            out.append((i + 1, float(c[1]), float(c[2]), int(c[3])))
    return out


def winner_list(path, pattern="win_actual = ifelse"):
    """Every c('CA', ...) that follows a `*_win_actual = ifelse(state %in%` in the file, in order."""
    txt = Path(path).read_text(errors="replace")
    found = []
    for m in re.finditer(pattern + r"\(state %in% c\((.*?)\),\s*1,\s*0\)", txt, flags=re.S):
        found.append((txt[:m.start()].count("\n") + 1, re.findall(r"'([A-Z]{2})'", m.group(1))))
    return found


def rmse_lines(lines):
    """The three `## [1] 0.0...` outputs of `model_v_actual.<year>.rmse`, in file order."""
    return [(i + 1, ln.split("]")[1].strip()) for i, ln in enumerate(lines) if ln.strip().startswith("## [1] 0.0")]


def actual_results(year):
    """dem / (dem + rep) by state from data/potus_results_76_16.csv (README.Rmd:394-396)."""
    import csv
    out = {}
    with open(REF / "data" / "potus_results_76_16.csv") as f:
        for r in csv.DictReader(f):
            if int(r["year"]) == year:
                out[r["state"]] = float(r["dem"]) / (float(r["dem"]) + float(r["rep"]))
    return out


def main():
    lines = (REF / "README.md").read_text().splitlines()
    tables, perf, rmse = state_tables(lines), performance(lines), rmse_lines(lines)
    assert len(tables) == 3 and len(perf) == 3 and len(rmse) == 3, (len(tables), len(perf), rmse)
    rmd = winner_list(REF / "README.Rmd")
    assert len(rmd) == 3, rmd
    for k, year in enumerate(YEARS):
        first, last, rows = tables[k]
        assert len(rows) == 52, (year, len(rows))
        scr_line, scr = winner_list(REF / "scripts" / "model" / f"final_{year}.R")[0]
        rmd_line, won = rmd[k]
        pl, evb, ub, sc = perf[k]
        actual = actual_results(year)
        out = [f"# reference README.md:{first + 1}-{last} (election-day predicted_score), performance row README.md:{pl}",
               f"# winners: README.Rmd:{rmd_line} (won_readme), scripts/model/final_{year}.R:{scr_line} (won_script)",
               f"# ev_wtd_brier = {evb}", f"# unwtd_brier = {ub}", f"# states_correct = {sc}",
               f"# rmse source: README.md:{rmse[k][0]}; actual = dem / (dem + rep) from data/potus_results_76_16.csv, DC left out (README.Rmd:392-401)",
               f"# rmse_ex_dc = {rmse[k][1]}",
               "state,mean,low,high,prob,se,won_readme,won_script,actual"]
        for st, mean, low, high, prob, se in rows:
            st = "--" if st in ("–", "--", "-") else st
            nat = st == "--"
            out.append(f"{st},{mean:.3f},{low:.3f},{high:.3f},{prob:.3f},{se:.3f},"
                       f"{'' if nat else int(st in won)},{'' if nat else int(st in scr)},{'' if nat else repr(actual[st])}")
        tab = [(mean, actual[st]) for st, mean, *_ in rows if st not in ("–", "--", "-", "DC")]
        print(year, "rmse of the rounded table means", (sum((a - b) ** 2 for a, b in tab) / len(tab)) ** 0.5, "published", rmse[k][1])
        (GOLD / f"readme_{year}.csv").write_text("\n".join(out) + "\n")
        print(year, "rows", len(rows), "brier", evb, ub, sc, "winners", len(won), "/", len(scr))


if __name__ == "__main__":
    main()
