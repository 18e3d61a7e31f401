"""PCIe-inclusive rates of the host-pointer entry points on config 2, next to the device-resident
rate of the same work (never bench.py's `value`; the numbers go into DESIGN.md).

  A  ldsr_em_batch             niter=100, tol=0   (the bench's work through host pointers)
  B  ldsr_em_batch_device      same, operands resident in HBM (what bench.py times)
  C  ldsr_em_restart_grid      niter=1000, tol=1e-5: the R shim's exact call pattern -- all
                               restarts, selection, the winner's trace and fit in ONE call
  D  ldsr_em_batch_device      same convergence run, operands resident, no selection / fit
  E  the round-1 pattern for C: ldsr_em_batch with the full [n x niter] trace + ldsr_smooth_batch
"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402  (before the library: see tests/conftest.py)
import ldsr_amd  # noqa: E402
from ldsr_amd import _lib, synth  # noqa: E402

mask = sys.argv[1] if len(sys.argv) > 1 else "dense"
T, p, q, n = 1000, 1, 2, 4096
y, u, v = synth.make_series(T, p, q, mask=mask)
th0 = synth.make_init_packed(p, q, n, seed=1)
L = _lib.lib()
dev = torch.device("cuda", 0)


def timeit(f, reps=20, warm=3):
    for _ in range(warm):
        r = f()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(reps):
        r = f()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / reps, r


d = {"y": torch.from_numpy(y[None].copy()).to(dev), "u": torch.from_numpy(np.ascontiguousarray(u.T)).to(dev),
     "v": torch.from_numpy(np.ascontiguousarray(v.T)).to(dev), "th0": torch.from_numpy(th0).to(dev),
     "th": torch.empty((n, 6 + p + q), dtype=torch.float64, device=dev),
     "lik": torch.empty(n, dtype=torch.float64, device=dev),
     "nit": torch.empty(n, dtype=torch.int32, device=dev), "st": torch.empty(n, dtype=torch.int32, device=dev)}
wsb = L.ldsr_em_workspace_bytes(1, T, p, q, n, 0)
ws = torch.empty(wsb + 256, dtype=torch.uint8, device=dev)
ws_ptr = (ws.data_ptr() + 255) & ~255
off = (C.c_int * 2)(0, n)
stream = torch.cuda.current_stream(dev)


def device_call(niter, tol):
    _lib.check(L.ldsr_em_batch_device(0, C.c_void_p(stream.cuda_stream), 1, T, p, q, d["y"].data_ptr(),
                                      d["u"].data_ptr(), d["v"].data_ptr(), 0, off, d["th0"].data_ptr(),
                                      niter, tol, 0, d["th"].data_ptr(), d["lik"].data_ptr(),
                                      d["nit"].data_ptr(), d["st"].data_ptr(), None, C.c_void_p(ws_ptr), wsb))
    torch.cuda.synchronize(dev)


tA, rA = timeit(lambda: ldsr_amd.em_batch(y, u, v, th0, niter=100, tol=0.0))
tB, _ = timeit(lambda: device_call(100, 0.0))
uA = n * 100
print("[%s] A ldsr_em_batch        niter=100 tol=0 : %.3f ms/call = %.3g units/s" % (mask, tA * 1e3, uA / tA))
print("[%s] B device-resident      niter=100 tol=0 : %.3f ms/call = %.3g units/s   (A/B = %.3f)"
      % (mask, tB * 1e3, uA / tB, tA / tB))

tC, rC = timeit(lambda: ldsr_amd.em_restart_grid(y, u, v, th0, niter=1000, tol=1e-5), reps=10)
units = int(rC["all"]["n_iter"].sum())
tD, _ = timeit(lambda: device_call(1000, 1e-5), reps=10)


def round1_pattern():
    r = ldsr_amd.em_batch(y, u, v, th0, niter=1000, tol=1e-5, return_liks=True)
    k = ldsr_amd.select_restart(r["lik"], r["theta"], p, q)
    return ldsr_amd.smooth_batch(y, u, v, r["theta"][k:k + 1])


tE, _ = timeit(round1_pattern, reps=5)
print("[%s] C ldsr_em_restart_grid niter=1000 tol=1e-5 (%d E-steps, winner %d): %.3f ms/call = %.3g units/s"
      % (mask, units, int(rC["winner"][0]), tC * 1e3, units / tC))
print("[%s] D device-resident      same run, no selection/fit: %.3f ms/call = %.3g units/s   (C/D = %.3f)"
      % (mask, tD * 1e3, units / tD, tC / tD))
print("[%s] E round-1 pattern (full trace D2H + separate smoother call): %.3f ms/call   (E/D = %.3f)"
      % (mask, tE * 1e3, tE / tD))
