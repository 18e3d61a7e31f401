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


def test_pmc_summary_is_only_used_for_the_kernel_it_was_measured_on():
    """roofline.traffic / issue_frac come from committed rocprofv3 passes: an entry counts only
    for the workload AND kernel instantiation AND iteration setup it was measured on; anything
    else is flagged stale (round 1 printed constants of an older kernel unconditionally)."""
    import json
    import bench
    entries = json.load(open(os.path.join(ROOT, "profiles", "pmc_summary.json")))["entries"]
    e = sorted([x for x in entries if x["workload"] == "cfg2"], key=lambda x: x["round"])[-1]   # the latest round's
    got, stale = bench.load_pmc("cfg2", "dense", e["kernel"], e["niter"], e["tol"])
    assert got is e or got == e
    assert stale is False
    got, stale = bench.load_pmc("cfg2", "dense", "em_scan_kernel<9, 9, 99, 1, false, false, false>", 100, 0.0)
    assert stale is True and got["workload"] == "cfg2"
    got, stale = bench.load_pmc("cfg2", "dense", e["kernel"], 1000, 1e-5)
    assert stale is True
    assert bench.load_pmc("custom", "dense", e["kernel"], 100, 0.0) == (None, None)
    assert bench.flops_per_unit(1000, 1, 2) == 68000


def test_plain_multi_gpu_invocation_spawns_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` without WORLD_SIZE starts N ranks as a child
    torch.distributed.run (no exec from a GPU-initialised process) on 127.0.0.1."""
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0
    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3"])
    assert bench.spawn_ranks(8) == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "8", "--steps", "3"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_pmc_entry_of_the_latest_round_counts():
    """bench.py quotes counter-measured fields (traffic, issue_frac, fp64_executed_frac) from the committed
    rocprofv3 passes of the SAME kernel: when several rounds profiled a kernel of that name, the latest
    round's entry -- not the first in the file -- and every committed bench line names the passes of
    its own round."""
    import json
    import os
    import bench
    e, stale = bench.load_pmc("cfg2", "dense", "em_pair_kernel<1, 2, 32, 32, false, false>", 100, 0.0)
    assert e is not None and stale is False
    rounds = sorted({x["round"] for x in json.load(open(os.path.join(bench.ROOT, "profiles", "pmc_summary.json")))["entries"]
                     if x.get("workload") == "cfg2"})
    assert e["round"] == rounds[-1]
    e2, stale2 = bench.load_pmc("cfg2", "dense", "some_other_kernel", 100, 0.0)
    assert e2 is not None and stale2 is True and e2["round"] == rounds[-1]
    for w in ("cfg2", "cfg3", "cfg4", "cfg5"):
        line = json.loads(open(os.path.join(bench.ROOT, "profiles", "%s_%s_bench.json" % (rounds[-1], w))).read().strip().splitlines()[-1])
        src = line["roofline"].get("pmc_source") or []
        assert src and all(("/%s_" % rounds[-1]) in s_ for s_ in src), (w, src)
        assert line["roofline"]["pmc_stale"] is False and line["verified"]["ok"] is True


def test_sustained_clock_fields_of_the_roofline():
    """roofline.peak assumes the 2.4 GHz boost clock; the chip holds less under fp64 load.  The
    committed PMC passes carry the clock they saw (GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration) and
    the bench line prices the same rate against the peak at THAT clock too."""
    import json
    import bench
    entries = json.load(open(os.path.join(ROOT, "profiles", "pmc_summary.json")))["entries"]
    latest = sorted({e["round"] for e in entries})[-1]
    for e in entries:
        if e["round"] == latest:
            assert 1.5 < e["sustained_clock_ghz"] < 2.45, e["workload"]
    src = open(os.path.join(ROOT, "bench.py")).read()
    for field in ("sustained_clock_ghz", "frac_at_sustained_clock", "peak_at_sustained_clock", "host_entry"):
        assert field in src
    assert bench.FP64_VALU_PEAK_TFLOPS == 78.6 and bench.CLOCK_HZ == 2.4e9


def test_host_entry_leg_builds_the_call_it_times(monkeypatch):
    """The host_entry leg hands ldsr_em_batch the SAME operands as the device-resident job (host
    copies of them) and reports units from the returned iteration counts -- checked here against a
    stand-in library (no GPU): argument order, shapes and the reported fields."""
    import types
    import bench
    Y, U, V, shared, off, th0, n_global = bench.build_problem("cfg5", "dense", 8, 3)
    seen = {}

    class FakeTensor:
        def __init__(self, a):
            self.a = a

        def cpu(self):
            return self

        def numpy(self):
            return self.a

    job = types.SimpleNamespace(d_y=FakeTensor(Y), d_u=FakeTensor(U), d_v=FakeTensor(V), d_th0=FakeTensor(th0),
                                S=Y.shape[0], loc_off=np.asarray(off), shared_uv=shared, local_rank=0)

    def fake_em_batch(device, S, T, p, q, y, u, v, shared_uv, off_, th0_, niter, tol, algo, th, lik, nit, st, liks):
        import ctypes as C
        seen.update(S=S, T=T, p=p, q=q, shared=shared_uv, niter=niter, off=[off_[i] for i in range(S + 1)])
        n = off_[S]
        assert isinstance(y, C.POINTER(C.c_double)) and isinstance(nit, C.POINTER(C.c_int))   # _lib.SIGNATURES' types
        src = np.full(n, niter, np.int32)
        C.memmove(nit, src.ctypes.data, 4 * n)
        return 0
    L = types.SimpleNamespace(ldsr_em_batch=fake_em_batch)
    from ldsr_amd import _lib
    monkeypatch.setattr(_lib, "check", lambda rc: None)
    w = bench.WORKLOADS["cfg5"]
    r = bench.host_entry_rate(L, job, w["T"], w["p"], w["q"], 7, 0.0, 0, th0.shape[0] * 7, steps=2, warmup=1)
    assert seen["S"] == 48 and seen["T"] == 813 and seen["niter"] == 7 and seen["off"] == [int(x) for x in off]
    assert r["units_per_call"] == th0.shape[0] * 7 and r["same_units_as_device_entry"] is True
    assert r["value"] > 0 and r["bytes_in"] == Y.nbytes + U.nbytes + V.nbytes + th0.nbytes
