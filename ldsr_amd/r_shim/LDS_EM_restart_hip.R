# Drop-in replacement for ldsr::LDS_EM_restart (reference R/LDS_reconstruction.R:42-62):
# same signature, same return shape (theta, fit, liks, lik, [init]); every restart runs in one
# GPU launch instead of one LDS_EM() per foreach task.  Load the side-car DLL once:
#   dyn.load("ldsrhip.so")
# and either call this function directly or install it over the package's own:
#   assignInNamespace("LDS_EM_restart", LDS_EM_restart_hip, ns = "ldsr")
# after which LDS_reconstruction() and cvLDS() use the GPU path unchanged.
LDS_EM_restart_hip <- function(y, u, v, init, niter = 1000, tol = 1e-5, return.init = TRUE) {
  storage.mode(y) <- "double"; storage.mode(u) <- "double"; storage.mode(v) <- "double"
  res <- .Call("ldsrhip_LDS_EM_batch", y, u, v, init, as.integer(niter), as.double(tol))
  ans <- res[c("theta", "fit", "liks", "lik")]
  if (return.init) ans$init <- init[[res$index]]
  ans
}
