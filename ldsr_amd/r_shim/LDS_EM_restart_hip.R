# Drop-in replacement for ldsr::LDS_EM_restart (reference R/LDS_reconstruction.R:42-62):
# same signature, same return shape (theta, fit, liks, lik, [init]); every restart runs in one
# GPU launch instead of one LDS_EM() per foreach task.  Load the side-car DLL once:
#   dyn.load("ldsrhip.so")
# and either call this function directly or install it over the package's own:
#   assignInNamespace("LDS_EM_restart", LDS_EM_restart_hip, ns = "ldsr")
# after which LDS_reconstruction() and cvLDS() use the GPU path unchanged.
# return.raw = TRUE additionally attaches $raw = propagate(theta, u, v, y) of the winner (what
# LDS_reconstruction's format_results computes for return.raw, R/LDS_reconstruction.R:219-222)
# from the same call.
LDS_EM_restart_hip <- function(y, u, v, init, niter = 1000, tol = 1e-5, return.init = TRUE,
                               return.raw = FALSE) {
  storage.mode(y) <- "double"; storage.mode(u) <- "double"; storage.mode(v) <- "double"
  res <- .Call(if (return.raw) "ldsrhip_LDS_EM_batch_raw" else "ldsrhip_LDS_EM_batch",
               y, u, v, init, as.integer(niter), as.double(tol))
  ans <- res[c("theta", "fit", "liks", "lik")]
  if (return.raw) ans$raw <- res$raw
  if (return.init) ans$init <- init[[res$index]]
  ans
}

# The ensemble loop of LDS_reconstruction (R/LDS_reconstruction.R:242-246),
#   foreach(i = seq_along(u)) %dopar% call_method(y, u[[i]], v[[i]], method, init[[i]], ...),
# in ONE call: u, v are lists of input matrices (members may differ in p and q), init a list of
# init lists.  Returns a list with one LDS_EM_restart result per member; the members run
# concurrently on the GPU(s).
LDS_EM_restart_ensemble_hip <- function(y, u, v, init, niter = 1000, tol = 1e-5, return.init = TRUE,
                                        return.raw = FALSE) {
  Y <- matrix(as.double(y), ncol = 1)                       # T x 1: one "fold"
  res <- .Call("ldsrhip_LDS_EM_groups", Y, lapply(u, function(m) { storage.mode(m) <- "double"; m }),
               lapply(v, function(m) { storage.mode(m) <- "double"; m }),
               lapply(init, function(ii) list(ii)), as.integer(niter), as.double(tol), return.raw)
  lapply(seq_along(res), function(i) {
    m <- res[[i]][[1]]
    ans <- m[c("theta", "fit", "liks", "lik")]
    if (return.raw) ans$raw <- m$raw
    if (return.init) ans$init <- init[[i]][[m$index]]
    ans
  })
}
