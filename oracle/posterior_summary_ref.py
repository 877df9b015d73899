"""CPU restatement (numpy) of the posterior summaries the run scripts compute from
rstan::extract(out, "predicted_score") -- TEST INFRASTRUCTURE ONLY (imported by tests/ to check
potus_posterior_summary; never by the product path).

Follows scripts/model/final_2016.R:
  :720-731  per state and day: quantile(x, 0.025), quantile(x, 0.975), mean(x), mean(x > 0.5)
  :742-762  national vote per draw and day = weighted.mean(score over states, state_weights), same four
  :799-823  electoral college: dem_ev = sum(ev * (score > 0.5)) per draw and day ->
            mean, median, quantile 0.975, quantile 0.025, mean(dem_ev >= 270)
R's quantile() default is type 7 = numpy's default 'linear' interpolation.
"""
import numpy as np


def posterior_summary(predicted_score, state_weights, ev):
    """predicted_score: [draws, T, S].  Returns dict(state [T,S,4], national [T,4], electoral_votes [T,5])."""
    ps = np.asarray(predicted_score, dtype=np.float64)
    w = np.asarray(state_weights, dtype=np.float64)
    ev = np.asarray(ev, dtype=np.float64)
    q = lambda x, p: np.quantile(x, p, axis=0)
    state = np.stack([q(ps, 0.025), q(ps, 0.975), ps.mean(axis=0), (ps > 0.5).mean(axis=0)], axis=-1)
    natl = (ps * w).sum(axis=2) / w.sum()
    national = np.stack([q(natl, 0.025), q(natl, 0.975), natl.mean(axis=0), (natl > 0.5).mean(axis=0)], axis=-1)
    dem_ev = ((ps > 0.5) * ev).sum(axis=2)
    evs = np.stack([dem_ev.mean(axis=0), np.median(dem_ev, axis=0), q(dem_ev, 0.975), q(dem_ev, 0.025), (dem_ev >= 270).mean(axis=0)], axis=-1)
    return dict(state=state, national=national, electoral_votes=evs)
