"""Host-side mirror of the reference's sampler call surface, over libpotus_hmc.so.

The reference drives CmdStan from R (R is not available in this image, so the host side
above the C ABI is Python; R/potus_sampling.R is the same thin shim for R users):

    model <- cmdstanr::cmdstan_model("scripts/model/poll_model_2020.stan")   # final_2016.R:532
    fit   <- model$sample(data=, seed=1843, parallel_chains=, chains=,
                          iter_warmup=, iter_sampling=, refresh=)            # final_2016.R:533-541
    out   <- rstan::read_stan_csv(fit$output_files())                        # final_2016.R:543
    rstan::extract(out, pars = "mu_b")[[1]]                                  # final_2016.R:556,...

becomes

    model = PotusModel("full")                       # or "no_mode_adjustment" (final_2012.R:558)
    fit   = model.sample(data=, seed=1843, chains=, iter_warmup=, iter_sampling=, refresh=)
    fit.extract("mu_b")                              # [draws, S, T], chains merged

`sampling()` is the rstan::sampling() spelling (iter includes warmup; final_2016.R:525-529).
There is no CPU fallback: without the HIP library and an MI355X every call raises.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

from . import _abi

_LIB = None
_HERE = Path(__file__).resolve().parent


class PotusError(RuntimeError):
    pass


def lib_path() -> Path:
    # POTUS_LIB selects a development build (e.g. the -DPOTUS_PROF one); default is the product library
    return Path(os.environ["POTUS_LIB"]) if os.environ.get("POTUS_LIB") else _HERE / "libpotus_hmc.so"


def load_library():
    """dlopen libpotus_hmc.so (built in-tree by __graft_entry__.build()). Fails loudly."""
    global _LIB
    if _LIB is not None:
        return _LIB
    p = lib_path()
    if not p.exists():
        raise PotusError(f"{p} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
    L = C.CDLL(str(p))
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    L.potus_version.restype = C.c_char_p
    L.potus_last_error.argtypes = [C.c_char_p, C.c_int]
    L.potus_default_opts.argtypes = [C.POINTER(_abi.PotusOpts)]
    L.potus_num_params.argtypes = [C.POINTER(_abi.PotusData), ip]
    L.potus_num_columns.argtypes = [C.POINTER(_abi.PotusData), ip]
    L.potus_column_name.argtypes = [C.POINTER(_abi.PotusData), C.c_int, C.c_char_p, C.c_int]
    L.potus_create.argtypes = [C.POINTER(_abi.PotusData), C.POINTER(_abi.PotusOpts), ip]
    L.potus_destroy.argtypes = [C.c_int]
    L.potus_log_prob_grad.argtypes = [C.c_int, dp, C.c_int, dp, dp]
    L.potus_init.argtypes = [C.c_int, dp]
    L.potus_run.argtypes = [C.c_int, C.c_int]
    L.potus_run_many.argtypes = [ip, C.c_int, C.c_int]
    L.potus_iterations_done.argtypes = [C.c_int, ip]
    L.potus_total_leapfrogs.argtypes = [C.c_int, C.POINTER(C.c_longlong)]
    L.potus_chain_status.argtypes = [C.c_int, ip, ip]
    L.potus_get_adaptation.argtypes = [C.c_int, dp, dp]
    L.potus_get_draws.argtypes = [C.c_int, dp, ip]
    L.potus_draws_device_ptr.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_longlong)]
    L.potus_write_array.argtypes = [C.c_int, C.c_int, C.c_int, dp]
    L.potus_write_stan_csv.argtypes = [C.c_int, C.c_char_p, C.c_char_p]
    L.potus_last_run_timing.argtypes = [C.c_int, dp, C.POINTER(C.c_longlong)]
    L.potus_posterior_summary.argtypes = [C.c_int, dp, dp, dp, dp]
    L.potus_posterior_summary_many.argtypes = [ip, C.c_int, dp, dp, dp, dp]
    L.potus_backtest_scores.argtypes = [dp, C.c_int, C.c_int, C.c_int, dp, ip, dp]
    L.potus_write_array_device.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.potus_diagnostics.argtypes = [ip, C.c_int, C.c_int, C.c_int, dp, dp]
    L.potus_diagnostics_device.argtypes = [C.c_int, C.c_void_p, C.c_longlong, C.c_int, C.c_int, dp, dp]
    if hasattr(L, "potus_check_convergence"):           # (development builds selected through POTUS_LIB may predate an export)
        L.potus_check_convergence.argtypes = [ip, C.c_int, C.c_double, C.c_double, ip, dp, dp]
    L.potus_get_dense_metric.argtypes = [C.c_int, C.c_int, dp]
    if hasattr(L, "potus_dense_pool_window"):
        L.potus_dense_pool_window.argtypes = [C.c_int, ip, dp, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_longlong)]
        L.potus_dense_pool_finish.argtypes = [C.c_int, C.c_double]
    L.potus_iterations_done.argtypes = [C.c_int, ip]
    L.potus_cus_per_chain.argtypes = [C.c_int, ip]
    L.potus_clusters_per_chain.argtypes = [C.c_int, ip]
    L.potus_plan_cus_per_chain.argtypes = [C.c_int] * 7 + [ip, ip]
    L.potus_plan_sides.argtypes = [C.c_int] * 7 + [ip]
    L.potus_twin_stats.argtypes = [C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
    _LIB = L
    return L


EXPORTS = [
    "potus_version", "potus_last_error", "potus_default_opts", "potus_num_params", "potus_num_columns",
    "potus_column_name", "potus_create", "potus_destroy", "potus_cus_per_chain", "potus_clusters_per_chain", "potus_plan_cus_per_chain", "potus_plan_sides", "potus_twin_stats", "potus_log_prob_grad", "potus_init", "potus_run", "potus_run_many",
    "potus_iterations_done", "potus_total_leapfrogs", "potus_chain_status", "potus_get_adaptation",
    "potus_get_dense_metric", "potus_dense_timing", "potus_dense_adapt_timing", "potus_dense_pool_window", "potus_dense_pool_finish", "potus_dense_check", "potus_get_draws", "potus_draws_device_ptr", "potus_write_array", "potus_write_array_device", "potus_extract_matrix", "potus_write_stan_csv",
    "potus_last_run_timing", "potus_posterior_summary", "potus_posterior_summary_many", "potus_backtest_scores",
    "potus_diagnostics", "potus_diagnostics_device", "potus_check_convergence",
    "potus_R_create", "potus_R_init", "potus_R_run", "potus_R_run_many", "potus_R_num_columns", "potus_R_saved_count",
    "potus_R_write_array", "potus_R_write_stan_csv", "potus_R_posterior_summary", "potus_R_diagnostics", "potus_R_check_convergence", "potus_R_backtest_scores", "potus_R_last_error", "potus_R_destroy",
]


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _check(L, rc):
    if rc != 0:
        buf = C.create_string_buffer(512)
        L.potus_last_error(buf, 512)
        raise PotusError(f"libpotus_hmc error {rc}: {buf.value.decode()}")


class Handle:
    """One sampler instance on one GPU (a set of chains)."""

    def __init__(self, data: dict, variant="full", **opts):
        self.L = load_library()
        self.data = data
        self.variant = variant
        self._d, self._keep = _abi.make_data(data, variant)
        o = _abi.PotusOpts()
        self.L.potus_default_opts(C.byref(o))
        for k, v in opts.items():
            if not hasattr(o, k):
                raise TypeError(f"unknown sampler option {k!r}")
            setattr(o, k, v)
        self.opts = o
        h = C.c_int(-1)
        _check(self.L, self.L.potus_create(C.byref(self._d), C.byref(o), C.byref(h)))
        self.h = h.value
        D, nc = C.c_int(), C.c_int()
        _check(self.L, self.L.potus_num_params(C.byref(self._d), C.byref(D)))
        _check(self.L, self.L.potus_num_columns(C.byref(self._d), C.byref(nc)))
        self.D, self.n_cols = D.value, nc.value
        self.layout, ncols2 = _abi.column_layout(data, variant)
        assert ncols2 == self.n_cols
        k = C.c_int()
        _check(self.L, self.L.potus_cus_per_chain(self.h, C.byref(k)))
        self.cus_per_chain = k.value
        _check(self.L, self.L.potus_clusters_per_chain(self.h, C.byref(k)))
        self.clusters_per_chain = k.value          # 2: one cluster per end of the trajectory (opts twin)

    def close(self):
        if getattr(self, "h", None) is not None and self.h >= 0:
            self.L.potus_destroy(self.h)
            self.h = -1

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def log_prob_grad(self, q):
        q = np.ascontiguousarray(np.atleast_2d(q), dtype=np.float64)
        n = q.shape[0]
        assert q.shape[1] == self.D
        lp, g = np.zeros(n), np.zeros((n, self.D))
        _check(self.L, self.L.potus_log_prob_grad(self.h, _dp(q), n, _dp(lp), _dp(g)))
        return lp, g

    def init(self, q0=None):
        p = None
        if q0 is not None:
            q0 = np.ascontiguousarray(q0, dtype=np.float64).reshape(self.opts.chains, self.D)
            p = _dp(q0)
        _check(self.L, self.L.potus_init(self.h, p))

    def run(self, n_iter):
        _check(self.L, self.L.potus_run(self.h, int(n_iter)))

    def iterations_done(self):
        n = C.c_int()
        _check(self.L, self.L.potus_iterations_done(self.h, C.byref(n)))
        return n.value

    def total_leapfrogs(self):
        n = C.c_longlong()
        _check(self.L, self.L.potus_total_leapfrogs(self.h, C.byref(n)))
        return n.value

    def last_run_timing(self):
        ms, n = C.c_double(), C.c_longlong()
        _check(self.L, self.L.potus_last_run_timing(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def chain_status(self):
        ch = self.opts.chains
        st, dv = (C.c_int * ch)(), (C.c_int * ch)()
        _check(self.L, self.L.potus_chain_status(self.h, st, dv))
        return list(st), list(dv)

    def adaptation(self):
        ch = self.opts.chains
        eps, minv = np.zeros(ch), np.zeros((ch, self.D))
        _check(self.L, self.L.potus_get_adaptation(self.h, _dp(eps), _dp(minv)))
        return eps, minv

    def draws_saved(self):
        n = C.c_int()
        _check(self.L, self.L.potus_get_draws(self.h, None, C.byref(n)))
        return n.value

    def draws(self):
        """[chains, n_saved, 7 + D] on the unconstrained scale."""
        n = C.c_int()
        _check(self.L, self.L.potus_get_draws(self.h, None, C.byref(n)))
        out = np.zeros((self.opts.chains, n.value, _abi.N_SAMPLER_COLS + self.D))
        if n.value:
            _check(self.L, self.L.potus_get_draws(self.h, _dp(out), C.byref(n)))
        return out

    def posterior_summary(self, ev):
        """Device-side summaries of predicted_score over all saved draws (final_2016.R:708-762, 799-823).

        Returns dict(state=[T,S,4] (low, high, mean, prob), national=[T,4], electoral_votes=[T,5]
        (mean, median, high, low, P(>=270))); `ev` = electoral votes per state, in state order."""
        return posterior_summary([self], ev)

    def write_array_device(self, col_begin, col_end, out_tensor):
        """write_array straight into a torch tensor on this handle's GPU ([n_saved, chains, col_end - col_begin],
        float64, contiguous): the buffer an RCCL all-gather sends, no trip through the host."""
        import torch
        if not (out_tensor.is_cuda and out_tensor.dtype == torch.float64 and out_tensor.is_contiguous()):
            raise TypeError("write_array_device needs a contiguous float64 tensor on the GPU")
        n = C.c_int()
        _check(self.L, self.L.potus_get_draws(self.h, None, C.byref(n)))
        if out_tensor.numel() != n.value * self.opts.chains * (col_end - col_begin):
            raise ValueError(f"tensor has {out_tensor.numel()} elements, {n.value} x {self.opts.chains} x {col_end - col_begin} expected")
        torch.cuda.current_stream(out_tensor.device).synchronize()      # the library writes on its own stream
        _check(self.L, self.L.potus_write_array_device(self.h, col_begin, col_end, C.c_void_p(out_tensor.data_ptr())))
        return out_tensor

    def twin_stats(self):
        """Twin mode: (leapfrogs counted in the trajectories, leaves integrated by the backward side, by the forward side)."""
        a, b, c = C.c_longlong(), C.c_longlong(), C.c_longlong()
        _check(self.L, self.L.potus_twin_stats(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def dense_timing(self):
        """metric = dense_e: (milliseconds in k_dn_matvec, matrix passes, bytes of matrix streamed, leaf rounds) so far."""
        L = self.L
        L.potus_dense_timing.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
        ms, n, b, r = C.c_double(), C.c_longlong(), C.c_longlong(), C.c_longlong()
        _check(L, L.potus_dense_timing(self.h, C.byref(ms), C.byref(n), C.byref(b), C.byref(r)))
        return ms.value, n.value, b.value, r.value

    def dense_adapt_timing(self):
        """metric = dense_e: dict(cov_ms, chol_ms, init_stepsize_ms, window_ends) of the warm-up's window ends so far."""
        L = self.L
        L.potus_dense_adapt_timing.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]
        a, b, c, n = C.c_double(), C.c_double(), C.c_double(), C.c_int()
        _check(L, L.potus_dense_adapt_timing(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(n)))
        return dict(cov_ms=a.value, chol_ms=b.value, init_stepsize_ms=c.value, window_ends=n.value)

    def iterations_done(self):
        n = C.c_int()
        _check(self.L, self.L.potus_iterations_done(self.h, C.byref(n)))
        return n.value

    def pool_window(self):
        """pooled_metric = 2: (pending, count, mean, m2) of a window end that waits for the host -- mean [D] and m2 [D, LD] are torch tensors ON the handle's
        GPU that alias the library's buffers (zero copy; m2[:, :D] is the matrix of centred outer products, both triangles)."""
        pend, cnt, pm, p2, ld = C.c_int(), C.c_double(), C.c_void_p(), C.c_void_p(), C.c_longlong()
        _check(self.L, self.L.potus_dense_pool_window(self.h, C.byref(pend), C.byref(cnt), C.byref(pm), C.byref(p2), C.byref(ld)))
        if not pend.value:
            return False, 0.0, None, None
        import torch

        class _Dev:                                        # __cuda_array_interface__: torch wraps the device pointer without copying
            def __init__(s_, ptr, shape):
                s_.__cuda_array_interface__ = {"shape": shape, "typestr": "<f8", "data": (ptr, False), "version": 2}
        dev = f"cuda:{self.opts.device}"
        mean = torch.as_tensor(_Dev(pm.value, (self.D,)), device=dev)
        m2 = torch.as_tensor(_Dev(p2.value, (self.D, ld.value)), device=dev)
        return True, cnt.value, mean, m2

    def pool_finish(self, n_total):
        _check(self.L, self.L.potus_dense_pool_finish(self.h, C.c_double(float(n_total))))

    def dense_check(self, chain=0, n_probe=2):
        """metric = dense_e: (max ||L L' x - M^-1 x|| / ||M^-1 x|| over n_probe random x, ||L' p - u|| / ||u|| of the momentum solve)."""
        L = self.L
        L.potus_dense_check.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
        out = np.zeros(2)
        _check(L, L.potus_dense_check(self.h, int(chain), int(n_probe), _dp(out)))
        return float(out[0]), float(out[1])

    def dense_metric(self, chain=0):
        """metric = dense_e: the adapted D x D inverse metric of one chain."""
        out = np.zeros((self.D, self.D))
        _check(self.L, self.L.potus_get_dense_metric(self.h, int(chain), _dp(out)))
        return out

    def draws_device_ptr(self):
        p, n = C.c_void_p(), C.c_longlong()
        _check(self.L, self.L.potus_draws_device_ptr(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def write_array(self, col_begin, col_end, n_saved):
        out = np.zeros((n_saved, self.opts.chains, max(col_end - col_begin, 1)))
        if n_saved:
            _check(self.L, self.L.potus_write_array(self.h, col_begin, col_end, _dp(out)))
        return out

    def write_stan_csv(self, directory, basename="poll_model_2020"):
        os.makedirs(directory, exist_ok=True)
        _check(self.L, self.L.potus_write_stan_csv(self.h, str(directory).encode(), basename.encode()))
        off = self.opts.chain_id_offset
        return [str(Path(directory) / f"{basename}-{off + c + 1}.csv") for c in range(self.opts.chains)]


def posterior_summary(handles, ev):
    """potus_posterior_summary_many: the summaries of final_2016.R:708-762, 799-823 over the pooled draws of every
    listed handle (the chains of one posterior, on one GPU or several)."""
    h0 = handles[0]
    S, T = int(h0.data["S"]), int(h0.data["T"])
    ev = np.ascontiguousarray(ev, dtype=np.float64).reshape(S)
    st, na, eo = np.zeros((S, T, 4)), np.zeros((T, 4)), np.zeros((T, 5))
    ids = (C.c_int * len(handles))(*[h.h for h in handles])
    _check(h0.L, h0.L.potus_posterior_summary_many(ids, len(handles), _dp(ev), _dp(st), _dp(na), _dp(eo)))
    return dict(state=np.ascontiguousarray(st.transpose(1, 0, 2)), national=na, electoral_votes=eo)


def run_pooled(handles, n_iter, coll_device=None):
    """potus_run_many for dense samplers with pooled_metric = 2 (one posterior, its chains spread over several handles / GPUs / ranks): every window end is
    pooled over ALL of them before it is finished -- the handles of this process by adding their moments on the first handle's GPU, the ranks of an initialised
    torch.distributed group by all-reduces (RCCL over xGMI when coll_device is a GPU; parallel.pool_window_moments) -- so that every GPU ends up with the
    same inverse metric, estimated from the window draws of every chain of the job.  A declared deviation from Stan (potus_opts.pooled_metric)."""
    import torch
    from . import parallel
    left = int(n_iter)
    while left > 0:
        it0 = handles[0].iterations_done()
        run_many(handles, left)
        done = handles[0].iterations_done() - it0
        left -= done
        wins = [h.pool_window() for h in handles]
        if not any(w[0] for w in wins):
            if done == 0:
                break                                      # the sampler has run all its iterations
            continue
        if not all(w[0] for w in wins):
            raise PotusError("run_pooled: the handles do not share a warm-up schedule (a window end is pending on some of them only)")
        dev0 = wins[0][2].device
        cnt = float(sum(w[1] for w in wins))
        msum = sum((w[1] * w[2]).to(dev0) for w in wins)                       # sum of count x mean over this process's handles
        n_tot, gmean = parallel.pool_window_mean(cnt, msum, coll_device)       # ... and over the ranks
        for _, c, mean, m2 in wins:                                           # Chan: M2 += count (mean - pooled mean)(mean - pooled mean)'
            d = mean - gmean.to(mean.device)
            m2[:, :mean.numel()].addr_(d, d, alpha=c)
        total = wins[0][3]
        for w in wins[1:]:
            total += w[3].to(dev0)
        parallel.pool_window_m2(total, coll_device)                            # in place: the sum over the ranks
        for w in wins[1:]:
            w[3].copy_(total)
        torch.cuda.synchronize()
        for h in handles:
            h.pool_finish(n_tot)


def device_diagnostics(handles, col_begin, col_end):
    """potus_diagnostics: rank-normalised split R-hat and bulk ESS of columns [col_begin, col_end) of the output row over the pooled
    chains of the listed handles (one posterior), computed on the first handle's GPU.  Returns (rhat, ess_bulk), each [col_end - col_begin]."""
    h0 = handles[0]
    n = int(col_end) - int(col_begin)
    rhat, ess = np.zeros(n), np.zeros(n)
    ids = (C.c_int * len(handles))(*[h.h for h in handles])
    _check(h0.L, h0.L.potus_diagnostics(ids, len(handles), int(col_begin), int(col_end), _dp(rhat), _dp(ess)))
    return rhat, ess


def _need_check_convergence(L):
    """A development build selected through POTUS_LIB may predate the export: say so up front, with the library's name, instead of an
    AttributeError in the middle of a run whose warm-up has already been paid for."""
    if not hasattr(L, "potus_check_convergence"):
        raise PotusError(f"{lib_path()} does not export potus_check_convergence (an older development build?): rhat_stop / check_convergence need it")


def check_convergence(handles, rhat_below=1.01, ess_at_least=400.0):
    """potus_check_convergence: (converged, rhat_max, ess_bulk_min) of lp__ and mu_b[:, T] over the post-warm-up draws the pooled chains of
    the handles have saved so far (the online early-stop check of SURVEY 8(f4); a deviation from Stan when acted upon)."""
    h0 = handles[0]
    _need_check_convergence(h0.L)
    ids = (C.c_int * len(handles))(*[h.h for h in handles])
    conv, r, e = C.c_int(0), C.c_double(), C.c_double()
    _check(h0.L, h0.L.potus_check_convergence(ids, len(handles), C.c_double(rhat_below), C.c_double(ess_at_least), C.byref(conv), C.byref(r), C.byref(e)))
    return bool(conv.value), r.value, e.value


def device_diagnostics_of_block(block):
    """potus_diagnostics_device on a torch tensor [draws, chains, columns] (float64, contiguous, on a GPU) -- e.g. the result of the
    all-gather of potus_write_array_device blocks.  Returns (rhat, ess_bulk) as numpy arrays [columns]."""
    import torch
    if not (block.is_cuda and block.dtype == torch.float64 and block.is_contiguous() and block.dim() == 3):
        raise TypeError("device_diagnostics_of_block needs a contiguous float64 [draws, chains, columns] tensor on the GPU")
    L = load_library()
    nd, nc, ncol = (int(x) for x in block.shape)
    rhat, ess = np.zeros(ncol), np.zeros(ncol)
    torch.cuda.current_stream(block.device).synchronize()
    _check(L, L.potus_diagnostics_device(int(block.device.index or 0), C.c_void_p(block.data_ptr()), nd, nc, ncol, _dp(rhat), _dp(ess)))
    return rhat, ess


def backtest_scores(summary, ev, won, day=0):
    """final_2016.R:925-945: (EV-weighted Brier, unweighted Brier, states called correctly) from the `state` block of
    a posterior summary; `won` = 1 where the Democrat carried the state; day: 1-based, 0 = election day (the last)."""
    L = load_library()
    st = np.ascontiguousarray(np.asarray(summary["state"], dtype=np.float64).transpose(1, 0, 2))    # [S][T][4]: cell t + T*s
    S, T = st.shape[0], st.shape[1]
    ev = np.ascontiguousarray(ev, dtype=np.float64).reshape(S)
    w = (C.c_int * S)(*[int(x) for x in won])
    out = np.zeros(3)
    _check(L, L.potus_backtest_scores(_dp(st), T, S, int(day), _dp(ev), w, _dp(out)))
    return dict(ev_wtd_brier=float(out[0]), unwtd_brier=float(out[1]), states_correct=int(out[2]))


def plan_layout(chains, T, n_cus=256, cus_per_chain=0, twin=-1, metric="diag_e", one_workgroup_ok=True):
    """(workgroups per chain, clusters / workgroups per chain side count) as potus_create would first plan them for a device of
    n_cus compute units -- potus_plan_cus_per_chain + potus_plan_sides, no device needed."""
    L = load_library()
    k, low, sides = C.c_int(0), C.c_int(0), C.c_int(0)
    m = _abi.METRICS[metric] if isinstance(metric, str) else int(metric)
    _check(L, L.potus_plan_cus_per_chain(chains, T, n_cus, cus_per_chain, twin, m, int(bool(one_workgroup_ok)), C.byref(k), C.byref(low)))
    _check(L, L.potus_plan_sides(chains, k.value, n_cus, 1, cus_per_chain, twin, m, C.byref(sides)))
    K = k.value
    if sides.value == 1 and low.value:
        K = 16                                      # the smaller clusters were chosen for the sake of the second one
    return K, sides.value


def run_many(handles, n_iter):
    """potus_run_many: advance several handles (other posteriors, other GPUs) concurrently from one host thread."""
    if not handles:
        return
    L = handles[0].L
    ids = (C.c_int * len(handles))(*[h.h for h in handles])
    _check(L, L.potus_run_many(ids, len(handles), int(n_iter)))


class StanFit:
    """What the scripts use of rstan's stanfit: extract(pars), model_name, sampler params."""

    def __init__(self, handle, model_name: str):
        # one handle, or one per device holding consecutive blocks of chains (PotusModel.sample(devices=...))
        self._hs = list(handle) if isinstance(handle, (list, tuple)) else [handle]
        self._h = self._hs[0]
        self.model_name = model_name  # out@model_name, final_2016.R:825
        # only the 7 sampler columns come to the host here (through write_array); the unconstrained draws
        # (chains x draws x D doubles: 1 GB for the 2016 run) stay on the device until unconstrained() asks for them
        self.n_saved = self._hs[0].draws_saved()
        self.chains = sum(h.opts.chains for h in self._hs)
        self._sampler = np.transpose(self._write_array(0, _abi.N_SAMPLER_COLS), (1, 0, 2))       # [chain, draw, 7]
        self._draws = None

    def _write_array(self, a, b):
        """[iter, chain, b - a] over all devices, chains in id order."""
        return np.concatenate([h.write_array(a, b, self.n_saved) for h in self._hs], axis=1)

    def sampler_params(self):
        """dict of [chains, draws] arrays: lp__, accept_stat__, ... (rstan::get_sampler_params)."""
        return {n: self._sampler[:, :, i] for i, n in enumerate(_abi.SAMPLER_COLS)}

    def unconstrained(self):
        """[chains, draws, D] on the unconstrained scale (copied from the devices on first use)."""
        if self._draws is None:
            self._draws = np.concatenate([h.draws() for h in self._hs], axis=0)
        return self._draws[:, :, _abi.N_SAMPLER_COLS:]

    def summary(self, ev):
        """Device-side posterior summaries over all chains of the fit (see posterior_summary)."""
        return posterior_summary(self._hs, ev)

    def extract(self, pars, permuted=False):
        """rstan::extract(out, pars=)[[1]]: array [draws, ...dims], chains merged.

        rstan additionally permutes the merged draws at random; the summaries the scripts
        take (means, quantiles, P(>0.5): final_2016.R:708-762) are permutation invariant, so
        the default keeps chain-major order.
        """
        single = isinstance(pars, str)
        names = [pars] if single else list(pars)
        out = {}
        for name in names:
            if name in _abi.SAMPLER_COLS:
                out[name] = self._sampler[:, :, _abi.SAMPLER_COLS.index(name)].reshape(-1)
                continue
            if name not in self._h.layout:
                raise KeyError(f"unknown parameter {name!r}")
            a, b, dims = self._h.layout[name]
            arr = self._write_array(a, b)                            # [iter, chain, n]
            arr = np.transpose(arr, (1, 0, 2)).reshape(self.chains * self.n_saved, b - a)
            if dims:
                arr = arr.reshape((arr.shape[0],) + tuple(reversed(dims))).transpose(
                    (0,) + tuple(range(len(dims), 0, -1)))            # column-major -> [draw, *dims]
            else:
                arr = arr[:, 0]
            out[name] = arr
        if permuted:
            perm = np.random.default_rng(self._h.opts.seed).permutation(self.chains * self.n_saved)
            out = {k: v[perm] for k, v in out.items()}
        return out[pars] if single else out

    def as_array(self, pars):
        """as.array(stanfit)[, , pars]: [iterations, chains, columns]."""
        a, b, _ = self._h.layout[pars]
        return self._write_array(a, b)

    def output_files(self, directory, basename=None):
        """fit$output_files(): writes CmdStan CSVs for rstan::read_stan_csv (final_2016.R:543)."""
        files = []
        for h in self._hs:                           # files are numbered by global chain id
            files += h.write_stan_csv(directory, basename or self.model_name.replace("_model", ""))
        return files


class PotusModel:
    """cmdstanr::cmdstan_model() for the two poll models (final_2016.R:532, final_2012.R:558)."""

    def __init__(self, stan_file_or_variant="full"):
        v = str(stan_file_or_variant)
        if v.endswith(".stan"):
            v = "no_mode_adjustment" if "no_mode_adjustment" in v else "full"
        if v not in _abi.VARIANTS:
            raise ValueError(f"unknown model {stan_file_or_variant!r}")
        self.variant = v
        self.model_name = "poll_model_2020_model" if v == "full" else "poll_model_2020_no_mode_adjustment_model"

    def sample(self, data, seed=1843, chains=4, parallel_chains=None, iter_warmup=1000, iter_sampling=1000,
               refresh=100, adapt_delta=0.8, max_treedepth=10, init=2.0, save_warmup=False, device=0,
               chain_id_offset=0, show_messages=False, inits=None, devices=None, metric="diag_e", cus_per_chain=0, twin=-1,
               metric_storage="f64", rhat_stop=None, ess_stop=400.0, pooled_metric=False):
        """`devices`: GPU ids; the chains are dealt to them in consecutive blocks and advance together under
        potus_run_many (one host thread).  Chain ids -- hence RNG streams and draws -- do not depend on the split.
        `rhat_stop` (off by default; a DEVIATION from Stan, which always runs iter_sampling iterations): after every `refresh` transitions
        of the sampling phase the pooled chains' rank-normalised split R-hat / bulk ESS of lp__ and mu_b[:, T] are taken on the device
        (potus_check_convergence) and sampling ends once every R-hat < rhat_stop and every bulk ESS >= ess_stop; the draws up to that
        point are those of the uninterrupted run.  `self.last_convergence` keeps the checks.
        `pooled_metric` (off by default; metric = "dense_e"; a DEVIATION from Stan): one inverse metric per GPU, adapted at every window end from
        the draws of all chains on it (potus_opts.pooled_metric)."""
        from . import parallel
        if rhat_stop is not None:
            _need_check_convergence(load_library())
        devs = [int(device)] if devices is None else [int(d) for d in devices]
        hs, first = [], 0
        for r, dev in enumerate(devs):
            off, n_loc = parallel.chain_block(int(chains), r, len(devs))
            if n_loc == 0:
                continue
            h = Handle(data, self.variant, chains=n_loc, chain_id_offset=int(chain_id_offset) + off,
                       num_warmup=int(iter_warmup), num_samples=int(iter_sampling), max_depth=int(max_treedepth),
                       delta=float(adapt_delta), init_radius=float(init), seed=int(seed), device=dev,
                       save_warmup=int(bool(save_warmup)), metric=_abi.METRICS[metric], cus_per_chain=int(cus_per_chain), twin=int(twin),
                       metric_storage={"f64": _abi.STORAGE_F64, "f32": _abi.STORAGE_F32}[metric_storage], pooled_metric=int(bool(pooled_metric)))
            h.init(None if inits is None else np.asarray(inits)[off:off + n_loc])
            hs.append(h)
        total = int(iter_warmup) + int(iter_sampling)
        chunk = max(1, int(refresh)) if refresh else total
        done = 0
        self.last_convergence = []
        while done < total:
            n = min(chunk, total - done)
            if done < iter_warmup:
                n = min(n, int(iter_warmup) - done)          # the first check falls on a whole chunk of sampling draws
            run_many(hs, n)
            done += n
            if show_messages:
                phase = "Warmup" if done <= iter_warmup else "Sampling"
                print(f"Iteration: {done:5d} / {total} [{100 * done // total:3d}%]  ({phase})", flush=True)
            if rhat_stop is not None and done > iter_warmup and done < total:
                conv, r, e = check_convergence(hs, float(rhat_stop), float(ess_stop))
                self.last_convergence.append({"iterations": done, "sampling_draws": done - int(iter_warmup), "rhat_max": r, "ess_bulk_min": e, "converged": conv})
                if conv:
                    if show_messages:
                        print(f"Stopped after {done - int(iter_warmup)} sampling iterations: R-hat {r:.4f} < {rhat_stop}, bulk ESS {e:.0f} >= {ess_stop}", flush=True)
                    break
        self.last_handle = hs[0]
        self.last_handles = hs
        return StanFit(hs, self.model_name)


def sampling(model: PotusModel, data, chains=4, iter=2000, warmup=None, refresh=None, seed=1843, control=None,
             **kw):
    """rstan::sampling(model, data=, chains=, iter=, warmup=, refresh=) (final_2016.R:525-529)."""
    warmup = iter // 2 if warmup is None else warmup
    control = control or {}
    return model.sample(data, seed=seed, chains=chains, iter_warmup=warmup, iter_sampling=iter - warmup,
                        refresh=refresh if refresh is not None else max(iter // 10, 1),
                        adapt_delta=control.get("adapt_delta", 0.8), max_treedepth=control.get("max_treedepth", 10),
                        metric=control.get("metric", "diag_e"), **kw)
