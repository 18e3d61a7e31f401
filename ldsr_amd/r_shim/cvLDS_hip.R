# All folds of cvLDS in ONE GPU call.  Replaces the fold loop of the reference,
#   foreach(z = Z) %dopar% one_lds_cv(z, instPeriod, mu, y, u, v, ...)   (R/LDS_reconstruction.R:373-375),
# i.e. per fold: y[instPeriod][z] <- NA (:274), fresh make_init (:275), LDS_EM_restart (:276),
# fit$Y[instPeriod] + mu (:283) -- or, with use.raw, propagate(result$theta, u, v, y)$Y[instPeriod] + mu
# (:279-281).  Returns the same thing that loop returns: a list with one numeric vector
# (length(instPeriod)) per fold.  Load the side-car DLL once: dyn.load("ldsrhip.so")
cv_folds_hip <- function(Z, instPeriod, mu, y, u, v, num.restarts = 20, niter = 1000, tol = 1e-6,
                         use.raw = FALSE) {
  Y <- sapply(Z, function(z) { yz <- y; yz[instPeriod][z] <- NA; as.numeric(yz) })  # T x length(Z)
  inits <- lapply(Z, function(z) ldsr::make_init(nrow(u), nrow(v), num.restarts))
  storage.mode(Y) <- "double"; storage.mode(u) <- "double"; storage.mode(v) <- "double"
  fits <- .Call(if (use.raw) "ldsrhip_LDS_EM_grid_raw" else "ldsrhip_LDS_EM_grid",
                Y, u, v, inits, as.integer(niter), as.double(tol))
  lapply(fits, function(m) as.numeric((if (use.raw) m$raw$Y else m$fit$Y)[instPeriod] + mu))
}

# The nested loop of cvLDS for ensembles (R/LDS_reconstruction.R:377-381),
#   foreach(z = Z) %:% foreach(i = seq_along(u), .combine = cbind, .final = rowMeans) %dopar% one_lds_cv(...),
# in ONE call: folds x members x restarts.  u, v: lists of input matrices.  Returns, per fold, the
# mean over the members of their predictions -- what `.final = rowMeans` leaves.
cv_folds_ensemble_hip <- function(Z, instPeriod, mu, y, u, v, num.restarts = 20, niter = 1000,
                                  tol = 1e-6, use.raw = FALSE) {
  Y <- sapply(Z, function(z) { yz <- y; yz[instPeriod][z] <- NA; as.numeric(yz) })
  storage.mode(Y) <- "double"
  inits <- lapply(seq_along(u), function(i)
    lapply(Z, function(z) ldsr::make_init(nrow(u[[i]]), nrow(v[[i]]), num.restarts)))
  res <- .Call("ldsrhip_LDS_EM_groups", Y, lapply(u, function(m) { storage.mode(m) <- "double"; m }),
               lapply(v, function(m) { storage.mode(m) <- "double"; m }), inits,
               as.integer(niter), as.double(tol), use.raw)
  lapply(seq_along(Z), function(f)
    rowMeans(sapply(res, function(member) {
      m <- member[[f]]
      as.numeric((if (use.raw) m$raw$Y else m$fit$Y)[instPeriod] + mu)
    })))
}

# Drop-in for ldsr:::one_lds_cv (R/LDS_reconstruction.R:270-285, method = 'EM'): same signature and
# return, use.raw included; a single fold is a grid of one column.
one_lds_cv_hip <- function(z, instPeriod, mu, y, u, v, method = "EM", num.restarts = 20,
                           ub = NULL, lb = NULL, num.islands = 4, pop.per.island = 100,
                           niter = 1000, tol = 1e-6, use.raw = FALSE) {
  stopifnot(method == "EM")
  cv_folds_hip(list(z), instPeriod, mu, y, u, v, num.restarts, niter, tol, use.raw)[[1]]
}
