"""GPU box: distribution of the iteration counts of a workload run to convergence (what bounds the launch?)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from ldsr_amd import _lib
L = _lib.lib()
dev = torch.device("cuda:0")
for name in sys.argv[1:] or ["cfg5", "cfg4", "cfg3"]:
    w = bench.WORKLOADS[name]
    prob = bench.build_problem(name, "dense", 1, 0, "strong")
    job = bench.Job(L, torch, dev, 0, prob, w["T"], w["p"], w["q"], 1000, 1e-5, 0)
    job.step(); torch.cuda.synchronize()
    nit = job.d_nit.cpu().numpy()
    off = job.loc_off
    print(name, "cells", nit.size, "units", int(nit.sum()), "mean %.1f" % nit.mean(), "max", int(nit.max()),
          "pcts 50/90/99/99.9:", [int(x) for x in np.percentile(nit, [50, 90, 99, 99.9])],
          "cells >= 256: %d, >= 384: %d, >= 512: %d" % ((nit >= 256).sum(), (nit >= 384).sum(), (nit >= 512).sum()))
    per = [int(nit[off[s]:off[s + 1]].sum()) for s in range(len(off) - 1)]
    mx = [int(nit[off[s]:off[s + 1]].max()) for s in range(len(off) - 1)]
    print("   per series units min/mean/max:", min(per), int(np.mean(per)), max(per), " per series max n_iter:", sorted(mx)[-8:])
