# potus_sampling.R -- R host shim over libpotus_hmc.so: plain .C() everywhere (no Rcpp, no Rinternals.h needed), and -- where R/src/potus_call.c has been
# built (R CMD SHLIB) and loaded -- .Call() for the calls that return draws: the result is then allocated once and filled in place (long vectors welcome),
# where .C() copies every argument in and out and the shim permutes the block once more.
#
# Drop-in for the sampler call of the reference scripts:
#
#   scripts/model/final_2016.R:532-543
#     model <- cmdstanr::cmdstan_model("scripts/model/poll_model_2020.stan", compile=TRUE, force=TRUE)
#     fit   <- model$sample(data = data, seed = 1843, parallel_chains = n_cores, chains = n_chains,
#                           iter_warmup = n_warmup, iter_sampling = n_sampling, refresh = n_refresh)
#     out   <- rstan::read_stan_csv(fit$output_files())
#
# becomes
#
#   source("R/potus_sampling.R"); potus_load("us_potus_model_amd/libpotus_hmc.so")
#   fit <- potus_sample(data, variant = "full", seed = 1843, chains = n_chains,
#                       iter_warmup = n_warmup, iter_sampling = n_sampling, refresh = n_refresh)
#   mu_b <- potus_extract(fit, "mu_b")             # = rstan::extract(out, pars = "mu_b")[[1]]
#   out  <- rstan::read_stan_csv(potus_output_files(fit, tempdir()))   # the literal plumbing, if a stanfit is needed
#
# NOTE: R is not installed in the build image, so this file itself has never run; every .C() call below is
# replayed argument for argument (int* / double* / char** only, outputs through the same buffers) by
# tests/test_gpu_boundary.py::test_r_entry_points_replay_the_shim with ctypes.

# path: libpotus_hmc.so; call_path: R/src/potus_call.so (optional: without it every call goes through .C())
potus_load <- function(path, call_path = NULL) {
  dyn.load(path, local = FALSE)                  # (global: potus_call.so resolves the C ABI against it)
  if (!is.null(call_path)) dyn.load(call_path)
  invisible(is.loaded("potus_call_extract"))
}

.potus_check <- function(status) {
  if (status != 0L) {
    msg <- .C("potus_R_last_error", buf = paste(rep(" ", 512), collapse = ""), len = 512L)$buf
    stop(sprintf("libpotus_hmc error %d: %s", status, trimws(msg)), call. = FALSE)
  }
}

# gpus: device ids; the chains are dealt to them in contiguous blocks (chain ids, hence RNG streams and draws, do not
# depend on the number of GPUs) and all devices advance together under potus_R_run_many.
potus_sample <- function(data, variant = c("full", "no_mode_adjustment"), seed = 1843, chains = 4,
                         parallel_chains = chains, iter_warmup = 1000, iter_sampling = 1000, refresh = 100,
                         adapt_delta = 0.8, max_treedepth = 10, init = 2, device = 0, chain_id_offset = 0,
                         save_warmup = FALSE, gpus = device, cus_per_chain = 0, metric = c("diag_e", "dense_e"), twin = -1,
                         metric_storage = c("f64", "f32"), rhat_stop = NULL, ess_stop = 400, pooled_metric = FALSE) {
  # pooled_metric (off by default; metric = "dense_e" only; a DEVIATION from Stan / CmdStan, whose chains are separate processes): ONE inverse metric
  # per GPU, adapted at every window end from the draws of all chains on it (potus_opts.pooled_metric, include/potus_hmc.h).
  # rhat_stop (off by default; a DEVIATION from Stan / the reference, which always run iter_sampling iterations, final_2016.R:539): after every
  # `refresh` transitions of the sampling phase the pooled chains' rank-normalised split R-hat / bulk ESS of lp__ and mu_b[, T] are taken on the
  # device (potus_R_check_convergence) and sampling ends once every R-hat < rhat_stop and every bulk ESS >= ess_stop; the draws up to that point
  # are those of the uninterrupted run.
  metric_storage <- match.arg(metric_storage)
  metric <- match.arg(metric)
  variant <- match.arg(variant)
  full <- variant == "full"
  iv <- function(x, n) if (is.null(x)) integer(max(n, 1)) else as.integer(x)
  dv <- function(x, n) if (is.null(x)) double(max(n, 1)) else as.double(x)
  Ns <- as.integer(data$N_state_polls); Nn <- as.integer(data$N_national_polls)
  dims <- as.integer(c(Nn, Ns, data$T, data$S, data$P, if (full) data$M else 0L, if (full) data$Pop else 0L,
                       if (full) 0L else 1L))
  scalars <- as.double(c(data$sigma_c, if (full) data$sigma_m else 0, if (full) data$sigma_pop else 0,
                         data$sigma_measure_noise_national, data$sigma_measure_noise_state,
                         if (full) data$sigma_e_bias else 0, data$random_walk_scale, data$mu_b_T_scale,
                         data$polling_bias_scale))
  gpus <- as.integer(gpus)
  per <- rep(chains %/% length(gpus), length(gpus)) + (seq_along(gpus) <= chains %% length(gpus))   # chains per device
  first <- cumsum(c(0L, per))[seq_along(per)]
  handles <- integer(0); counts <- integer(0)
  for (g in seq_along(gpus)) {
    if (per[g] == 0L) next
    res <- .C("potus_R_create", dims,
              iv(data$state, Ns), iv(data$day_state, Ns), iv(data$day_national, Nn), iv(data$poll_state, Ns),
              iv(data$poll_national, Nn), iv(data$poll_mode_state, Ns), iv(data$poll_mode_national, Nn),
              iv(data$poll_pop_state, Ns), iv(data$poll_pop_national, Nn),
              iv(data$n_democrat_national, Nn), iv(data$n_two_share_national, Nn),
              iv(data$n_democrat_state, Ns), iv(data$n_two_share_state, Ns),
              dv(data$unadjusted_national, Nn), dv(data$unadjusted_state, Ns),
              as.double(data$mu_b_prior), as.double(data$state_weights), scalars,
              as.double(data$state_covariance_0),           # column-major, as R stores it
              as.integer(c(per[g], chain_id_offset + first[g], iter_warmup, iter_sampling, max_treedepth, gpus[g],
                           as.integer(save_warmup), cus_per_chain, if (metric == "dense_e") 1L else 0L, twin,
                           if (metric_storage == "f32") 1L else 0L, if (isTRUE(pooled_metric)) 1L else 0L)),
              as.double(c(adapt_delta, 0.05, 0.75, 10, 1, init, seed)),   # the seed as a double: exact to 2^53
              handle = integer(1), status = integer(1))
    .potus_check(res$status)
    .potus_check(.C("potus_R_init", res$handle, status = integer(1))$status)
    handles <- c(handles, res$handle); counts <- c(counts, per[g])
  }
  total <- iter_warmup + iter_sampling
  done <- 0L
  chunk <- if (is.null(refresh) || refresh <= 0) total else as.integer(refresh)
  convergence <- NULL
  while (done < total) {                     # chunked so that R can print progress / be interrupted
    n <- min(chunk, total - done)
    if (done < iter_warmup) n <- min(n, iter_warmup - done)
    .potus_check(.C("potus_R_run_many", as.integer(handles), length(handles), as.integer(n), status = integer(1))$status)
    done <- done + n
    message(sprintf("Iteration: %5d / %d [%3d%%]  (%s)", done, total, as.integer(100 * done / total),
                    if (done <= iter_warmup) "Warmup" else "Sampling"))
    if (!is.null(rhat_stop) && done > iter_warmup && done < total) {
      cv <- .C("potus_R_check_convergence", as.integer(handles), length(handles), as.double(c(rhat_stop, ess_stop)), converged = integer(1), out = double(2),
               status = integer(1))
      .potus_check(cv$status)
      convergence <- rbind(convergence, data.frame(iterations = done, rhat_max = cv$out[1], ess_bulk_min = cv$out[2], converged = cv$converged == 1L))
      if (cv$converged == 1L) {
        message(sprintf("Stopped after %d sampling iterations: R-hat %.4f < %g, bulk ESS %.0f >= %g", done - iter_warmup, cv$out[1], rhat_stop, cv$out[2], ess_stop))
        break
      }
    }
  }
  info <- .C("potus_R_num_columns", handles[1], D = integer(1), n_cols = integer(1), status = integer(1))
  .potus_check(info$status)
  saved <- .C("potus_R_saved_count", handles[1], n_saved = integer(1), status = integer(1))   # what the library holds, not what was asked for
  .potus_check(saved$status)
  structure(list(handle = handles[1], handles = handles, chains_per_handle = counts, D = info$D, n_cols = info$n_cols, chains = chains,
                 n_saved = saved$n_saved, data = data, variant = variant, convergence = convergence,
                 model_name = if (full) "poll_model_2020_model" else "poll_model_2020_no_mode_adjustment_model"),
            class = "potus_fit")
}

# rstan::sampling() surface (final_2016.R:525-529, scripts/deprecated/R/Refactored/poll_run_v9.R:387-390): `iter` counts
# warm-up + sampling, `warmup` defaults to iter / 2, control = list(adapt_delta =, max_treedepth =).
potus_sampling <- function(data, variant = c("full", "no_mode_adjustment"), chains = 4, iter = 2000, warmup = floor(iter / 2),
                           seed = 1843, refresh = max(iter %/% 10, 1), init = 2, control = list(), gpus = 0L, ...) {
  potus_sample(data, variant = match.arg(variant), seed = seed, chains = chains, iter_warmup = warmup, iter_sampling = iter - warmup,
               refresh = refresh, init = if (is.numeric(init)) init else 2,
               adapt_delta = if (is.null(control$adapt_delta)) 0.8 else control$adapt_delta,
               max_treedepth = if (is.null(control$max_treedepth)) 10 else control$max_treedepth, gpus = gpus, ...)
}

# column ranges of the CmdStan row, 0-based [begin, end): same arithmetic as _abi.column_layout
.potus_layout <- function(fit) {
  d <- fit$data; full <- fit$variant == "full"
  S <- d$S; T <- d$T; P <- d$P; Ns <- d$N_state_polls; Nn <- d$N_national_polls
  blocks <- list(raw_mu_b_T = S, raw_mu_b = c(S, T), raw_mu_c = P)
  if (full) blocks <- c(blocks, list(raw_mu_m = d$M, raw_mu_pop = d$Pop, mu_e_bias = integer(0), rho_e_bias = integer(0), raw_e_bias = T))
  blocks <- c(blocks, list(raw_measure_noise_national = Nn, raw_measure_noise_state = Ns, raw_polling_bias = S,
                           mu_b = c(S, T), mu_c = P))
  if (full) blocks <- c(blocks, list(mu_m = d$M, mu_pop = d$Pop, e_bias = T))
  blocks <- c(blocks, list(polling_bias = S, national_mu_b_average = T, national_polling_bias_average = integer(0)))
  if (full) blocks <- c(blocks, list(sigma_rho = integer(0)))
  blocks <- c(blocks, list(logit_pi_democrat_state = Ns, logit_pi_democrat_national = Nn, predicted_score = c(T, S)))
  col <- 7L; out <- list()
  for (nm in names(blocks)) { n <- as.integer(prod(blocks[[nm]])); out[[nm]] <- list(begin = col, end = col + n, dims = blocks[[nm]]); col <- col + n }
  out
}

# rstan::extract(out, pars = name)[[1]] : array [draws, dims...], chains merged (chain-major)
potus_extract <- function(fit, name) {
  b <- .potus_layout(fit)[[name]]
  if (is.null(b)) stop("unknown parameter ", name)
  n <- b$end - b$begin
  if (is.loaded("potus_call_extract")) {          # .Call(): ONE allocation, filled in place in R's own order [draws, columns] (R/src/potus_call.c)
    a <- .Call("potus_call_extract", as.integer(fit$handles), as.integer(b$begin), as.integer(b$end))
    return(if (length(b$dims)) { dim(a) <- c(nrow(a), b$dims); a } else as.vector(a))
  }
  parts <- lapply(seq_along(fit$handles), function(g) {
    ch <- fit$chains_per_handle[g]
    res <- .C("potus_R_write_array", fit$handles[g], as.integer(b$begin), as.integer(b$end),
              out = double(fit$n_saved * ch * n), status = integer(1))
    .potus_check(res$status)
    a <- aperm(array(res$out, c(n, ch, fit$n_saved)), c(3, 2, 1))         # [iter, chain, col]
    matrix(a, fit$n_saved * ch, n)                                          # chain-major merge of this device's chains
  })
  a <- do.call(rbind, parts)                                                # devices hold consecutive chain ids
  if (length(b$dims)) array(a, c(nrow(a), b$dims)) else as.vector(a)
}

potus_output_files <- function(fit, dir, basename = "poll_model_2020") {
  for (h in fit$handles) .potus_check(.C("potus_R_write_stan_csv", h, as.character(dir), as.character(basename), status = integer(1))$status)
  file.path(dir, sprintf("%s-%d.csv", basename, seq_len(fit$chains)))       # files are numbered by chain id
}

# Posterior summaries of predicted_score computed on the device (replaces final_2016.R:708-762 and :799-823:
# no 8000 x 12954 array ever reaches R), pooled over every chain of the fit whatever the number of GPUs.
# ev: electoral votes per state in state order (states2012$ev).
# Returns list(state = array [T, S, 4] (low, high, mean, prob), national = [T, 4],
#              electoral_votes = [T, 5] (mean, median, high, low, prob >= 270)).
potus_summary <- function(fit, ev) {
  S <- fit$data$S; T <- fit$data$T
  r <- .C("potus_R_posterior_summary", as.integer(fit$handles), length(fit$handles), as.double(ev), state = double(T * S * 4),
          natl = double(T * 4), ev_out = double(T * 5), status = integer(1))
  .potus_check(r$status)
  list(state = aperm(array(r$state, c(4, T, S)), c(2, 3, 1)),       # C order [s][t][4] -> [t, s, 4]
       national = t(matrix(r$natl, nrow = 4)), electoral_votes = t(matrix(r$ev_out, nrow = 5)),
       state_raw = r$state)
}

# Rank-normalised split R-hat and bulk ESS (Vehtari et al. 2021) of the columns `pars` names, over every chain of the fit, computed on
# the device (the reference never inspects a diagnostic, final_2016.R:543-556; rstan::monitor would need the draws in R).
# Returns data.frame(column, rhat, ess_bulk); columns are 0-based positions in the CmdStan row, as in potus_extract.
potus_diagnostics <- function(fit, col_begin, col_end) {
  n <- col_end - col_begin
  if (is.loaded("potus_call_diagnostics")) {
    m <- .Call("potus_call_diagnostics", as.integer(fit$handles), as.integer(col_begin), as.integer(col_end))
    return(data.frame(column = seq(col_begin, col_end - 1L), rhat = m[, 1], ess_bulk = m[, 2]))
  }
  r <- .C("potus_R_diagnostics", as.integer(fit$handles), length(fit$handles), as.integer(c(col_begin, col_end)), rhat = double(n), ess = double(n),
          status = integer(1))
  .potus_check(r$status)
  data.frame(column = seq(col_begin, col_end - 1L), rhat = r$rhat, ess_bulk = r$ess)
}

# Backtest scores of final_2016.R:925-945: EV-weighted Brier, unweighted Brier, states called correctly on `day`
# (1-based; 0 = the last day).  won: 1 where the Democrat carried the state, in state order.
potus_backtest_scores <- function(fit, summary, ev, won, day = 0L) {
  r <- .C("potus_R_backtest_scores", summary$state_raw, as.integer(c(fit$data$T, fit$data$S, day)), as.double(ev), as.integer(won),
          out = double(3), status = integer(1))
  .potus_check(r$status)
  c(ev_wtd_brier = r$out[1], unwtd_brier = r$out[2], states_correct = r$out[3])
}

potus_free <- function(fit) invisible(lapply(fit$handles, function(h) .C("potus_R_destroy", h, status = integer(1))))
