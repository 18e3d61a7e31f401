"""Host-side pieces of bench.py that run without a GPU: workload construction for every
BASELINE config at 1 and 8 ranks, and the cpu_baseline leg (so a refactor cannot silently break
the driver's default `python bench.py` invocation)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_workloads_shard_exactly():
    import bench
    for name, w in bench.WORKLOADS.items():
        for world in (1, 2, 8):
            total = 0
            for rank in range(world):
                Y, U, V, shared, off, th0, n_global = bench.build_problem(name, "dense", world, rank)
                assert off[0] == 0 and off[-1] == th0.shape[0] and len(off) == Y.shape[0] + 1
                assert th0.shape[1] == 6 + w["p"] + w["q"]
                assert U.shape[0] == (1 if shared or w["series"] == 1 else Y.shape[0])
                total += th0.shape[0]
            expect = w["series"] * w["restarts"] * (world if w["scaling"] == "weak" else 1)
            assert total == expect == n_global
    # restart r of the global grid is the same numbers whichever rank owns it
    a = bench.build_problem("cfg2", "dense", 1, 0)[5]
    b = bench.build_problem("cfg2", "dense", 2, 0)[5]
    assert np.array_equal(a, b)


def test_cpu_baseline_leg_runs():
    import bench
    Y, U, V, shared, off, th0, n_global = bench.build_problem("cfg2", "dense", 1, 0)
    r = bench.cpu_baseline(1, 2, 2, Y, U, V, seed=1)
    assert r["kind"] == "port" and r["value"] > 0 and r["cores"] >= 1
    assert bench.bytes_per_unit(1000, 1, 2) == 96000
