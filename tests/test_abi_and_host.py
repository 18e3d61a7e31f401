"""CPU-side checks: the C-ABI library loads and exports every symbol that include/ldsr_hip.h
declares, argument errors surface with messages, the product path refuses to run without a GPU
(no CPU fallback), and host-side helpers behave like the reference's R code."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "ldsr_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(ldsr_[a-z_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from ldsr_amd import _lib
    assert os.path.exists(_lib.SO_PATH), "run __graft_entry__.build() first"
    names = _declared_symbols()
    assert len(names) >= 12
    L = C.CDLL(_lib.SO_PATH)
    for n in names:
        assert hasattr(L, n), "missing symbol %s" % n
    assert set(names) == set(_lib.SIGNATURES), "ctypes table and header disagree"
    assert b"gfx950" in _lib.lib().ldsr_version()


def test_code_object_is_gfx950_only():
    """Every device code object in the library targets gfx950 and nothing else.  The objects are
    compressed offload bundles (hipcc --offload-compress: 93.7 -> 40 MB, the library travels to the
    GPU box on every run): header `CCOB`, version, method, total size, uncompressed size, hash,
    then one zstd frame."""
    import struct

    import pyarrow as pa
    from ldsr_amd import _lib
    blob = open(_lib.SO_PATH, "rb").read()
    assert os.path.getsize(_lib.SO_PATH) < 64 << 20
    triples, n, i = set(), 0, 0
    while True:
        i = blob.find(b"CCOB", i)
        if i < 0:
            break
        total, usize, _ = struct.unpack_from("<QQQ", blob, i + 8)
        raw = pa.Codec("zstd").decompress(blob[i + 32:i + total], usize).to_pybytes()
        triples |= set(re.findall(rb"hip[a-z0-9]*-amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", raw[:4096]))
        n += 1
        i += total
    assert n >= 50 and triples == {b"gfx950"}, (n, triples)


def test_argument_validation_without_gpu():
    from ldsr_amd import _lib
    L = _lib.lib()
    dp = C.POINTER(C.c_double)
    y = (C.c_double * 4)(0, 1, 2, 3)
    off = (C.c_int * 2)(0, 1)
    th = (C.c_double * 8)()
    out = (C.c_double * 8)()
    lik = (C.c_double * 1)()
    it = (C.c_int * 1)()
    st = (C.c_int * 1)()
    # T < 2
    rc = L.ldsr_em_batch(0, 1, 1, 1, 1, y, None, None, 0, off, th, 10, 1e-5, 0, out, lik, it, st, None)
    assert rc == 1 and b"T must be" in L.ldsr_last_error()
    # p > 16
    rc = L.ldsr_em_batch(0, 1, 4, 17, 1, y, None, None, 0, off, th, 10, 1e-5, 0, out, lik, it, st, None)
    assert rc == 2
    # niter < 2: the reference reads lik[1] unconditionally (src/EM.cpp:256)
    rc = L.ldsr_em_batch(0, 1, 4, 1, 1, y, None, None, 0, off, th, 1, 1e-5, 0, out, lik, it, st, None)
    assert rc == 1 and b"niter" in L.ldsr_last_error()
    # bad offsets
    bad = (C.c_int * 2)(1, 1)
    rc = L.ldsr_em_batch(0, 1, 4, 1, 1, y, None, None, 0, bad, th, 10, 1e-5, 0, out, lik, it, st, None)
    assert rc == 1
    assert L.ldsr_em_workspace_bytes(1, 1000, 1, 2, 4096, 0) > 0
    assert L.ldsr_em_workspace_bytes(1, 1000, 17, 2, 4096, 0) == 0
    assert L.ldsr_em_workspace_bytes(1, 1000, 12, 2, 64, 0) > 2 * 1000 * 64 * 8    # wide input: serial kernel
    # serial needs the [t][cell] strip, scan does not
    assert L.ldsr_em_workspace_bytes(1, 1000, 1, 2, 4096, 1) >= 2 * 1000 * 4096 * 8
    assert L.ldsr_em_workspace_bytes(1, 1000, 1, 2, 4096, 2) < 1 << 20
    assert L.ldsr_em_workspace_bytes(1, 5000, 1, 2, 64, 2) > 0      # four waves per cell
    assert L.ldsr_em_workspace_bytes(1, 9000, 1, 2, 64, 2) == 0     # scan kernel: T <= 8192
    assert L.ldsr_em_workspace_bytes(1, 9000, 1, 2, 64, 0) > 0      # AUTO: serial kernel


def test_launch_plan_of_every_shape():
    """ldsr_em_plan is host logic (no GPU): which kernel a device-filling launch of a shape gets.
    Four cells per wave up to T = 512, two up to 1024 (narrow inputs, where eight waves per CU fit),
    one to four waves per cell up to T = 8192 and p, q <= 8, one thread per cell beyond; with
    tol > 0 AUTO reports what the device entry runs (scan kernel + work queue)."""
    import ctypes as C
    from ldsr_amd import _lib
    L = _lib.lib()

    def plan(T, p, q, tol=0.0, algo=0):
        buf = C.create_string_buffer(160)
        a = L.ldsr_em_plan(T, p, q, 100, float(tol), algo, buf, 160)
        return a, buf.value.decode()

    assert plan(1000, 1, 2) == (3, "em_pair_kernel<1, 2, 32, 32, false, false>")          # BASELINE config 2
    assert plan(813, 1, 3) == (3, "em_pair_kernel<1, 4, 26, 32, false, false>")           # config 5
    assert plan(1000, 4, 8) == (2, "em_scan_kernel<4, 8, 16, 1, false, false, false>")   # config 3
    assert plan(2000, 1, 4) == (2, "em_scan_kernel<1, 4, 32, 1, false, false, false>")   # config 4
    assert plan(1000, 1, 2, 1e-5) == (2, "em_scan_kernel<1, 2, 16, 1, true, false, false>")
    assert plan(1000, 1, 2, 1e-5, 3) == (3, "em_pair_kernel<1, 2, 32, 32, true, false>")
    # ... except for short series (four-wave workgroups, chunks of <= 13 steps): two cells per wave
    # (short series keep the pair family with early stopping whatever the mask; a launch that fills
    # the device -- what the plan assumes -- gets four cells per wave: tools/auto_regret.py)
    assert plan(400, 1, 2, 1e-5) == (4, "em_pair_kernel<1, 2, 25, 16, true, false>")
    # (runs to convergence with padded p + q >= 8 stay at two cells per wave -- four lose, T = 260 (4,4) 20 000 cells
    # 1.13 against 0.97 ms -- and take the scan kernel beyond T = 512: profiles/r04_auto_regret.txt)
    assert plan(120, 4, 4, 1e-5) == (3, "em_pair_kernel<4, 4, 4, 32, true, false>")
    assert plan(120, 4, 4) == (4, "em_pair_kernel<4, 4, 8, 16, false, false>")
    assert plan(417, 1, 2, 1e-5)[0] == 2 and plan(500, 1, 4, 1e-5)[0] == 2
    # ... and when the caller of the device entry says every y_t is observed (lead_steps = -1)
    buf = C.create_string_buffer(160)
    assert L.ldsr_em_plan_lead(1000, 1, 2, 100, 1e-5, 0, -1, buf, 160) == 3
    assert buf.value.decode() == "em_pair_kernel<1, 2, 32, 32, true, false>"
    assert L.ldsr_em_plan_lead(1000, 1, 2, 100, 1e-5, 0, 0, buf, 160) == 2
    assert plan(85, 1, 2) == (4, "em_pair_kernel<1, 2, 6, 16, false, false>")
    assert plan(213, 3, 3) == (4, "em_pair_kernel<4, 4, 14, 16, false, false>")           # the NP test slice
    assert plan(85, 7, 7)[1].startswith("em_scan_kernel<8, 8,")                     # the P1 known-answer case
    assert plan(4000, 2, 2)[1].startswith("em_scan_kernel<2, 2, 32, 2, true, true, false>")
    assert plan(9000, 1, 1)[0] == 1 and plan(500, 9, 1)[0] == 1                      # serial kernel
    # the pair family's limits: T, widths, and the eight-waves-per-CU rule for wide inputs
    for T, p, q, a3, a4 in ((64, 1, 2, -1, -1), (65, 1, 2, 3, 4), (512, 1, 2, 3, 4), (513, 1, 2, 3, -1),
                            (1024, 1, 2, 3, -1), (1025, 1, 2, -1, -1), (1000, 1, 4, -1, -1), (928, 1, 4, 3, -1),
                            (800, 4, 4, 3, -1), (900, 4, 4, -1, -1), (512, 3, 3, 3, 4), (400, 3, 3, 3, 4),   # (512, 3, 3) x four: since round 4's unpadded image
                            (1000, 5, 1, -1, -1), (300, 1, 5, -1, -1)):
        assert plan(T, p, q, 0.0, 3)[0] == a3, (T, p, q)
        assert plan(T, p, q, 0.0, 4)[0] == a4, (T, p, q)
    # every T of the family maps its steps onto the lanes exactly (1 <= rp <= nl <= lanes per cell)
    for lpc, algo, Tmax in ((32, 3, 1024), (16, 4, 512)):
        for T in range(65, Tmax + 1):
            a, name = plan(T, 1, 2, 0.0, algo)
            assert a == algo, (T, algo)
            Lc = int(name.split(",")[2])
            nl = -(-T // Lc)
            rp = T - nl * (Lc - 1)
            assert nl <= lpc and 1 <= rp <= nl and Lc <= 32, (T, Lc, nl, rp)


def test_no_cpu_fallback():
    """Without a GPU the product path must fail loudly, never compute on the host."""
    import ldsr_amd
    from ldsr_amd import _lib, synth
    if _lib.lib().ldsr_device_count() > 0:
        pytest.skip("GPU present")
    y, u, v = synth.make_series(50, 1, 2)
    with pytest.raises(_lib.LdsrError):
        ldsr_amd.em_batch(y, u, v, synth.make_init_packed(1, 2, 2), niter=5)
    with pytest.raises(_lib.LdsrError):
        ldsr_amd.Kalman_smoother(y, u, v, synth.make_init_packed(1, 2, 1)[0])


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "ldsr_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".inc", ".c", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle|oracle/|libldsr_oracle", src,
                                     flags=re.M), "%s uses the oracle" % os.path.join(dirpath, f)


def test_select_restart_matches_reference_rule():
    """R/LDS_reconstruction.R:50-58 through the C ABI (host-only function)."""
    import ldsr_amd
    p, q = 1, 2
    th = np.zeros((4, 9))
    th[:, 2] = [0.5, -0.2, 0.1, 0.3]                 # C
    assert ldsr_amd.select_restart([1.0, 9.0, 3.0, np.nan], th, p, q) == 2
    th[:, 2] = -1.0
    assert ldsr_amd.select_restart([1.0, 9.0, 3.0, np.nan], th, p, q) == 1
    assert ldsr_amd.select_restart([np.nan] * 4, th, p, q) == -1


def test_make_init_distribution():
    """R/LDS_reconstruction.R:14-30: A,C ~ U(0,1); B,D ~ U(-1,1); Q=R=V1=1; mu1=0."""
    import ldsr_amd
    init = ldsr_amd.make_init(3, 2, 2000, seed=123)
    assert len(init) == 2000 and init[0]["B"].shape == (1, 3) and init[0]["D"].shape == (1, 2)
    A = np.array([t["A"][0, 0] for t in init])
    B = np.concatenate([t["B"].ravel() for t in init])
    Cc = np.array([t["C"][0, 0] for t in init])
    D = np.concatenate([t["D"].ravel() for t in init])
    assert 0 <= A.min() and A.max() < 1 and abs(A.mean() - 0.5) < 0.03
    assert 0 <= Cc.min() and Cc.max() < 1 and abs(Cc.mean() - 0.5) < 0.03
    assert -1 <= B.min() and B.max() < 1 and abs(B.mean()) < 0.04 and B.min() < -0.9
    assert -1 <= D.min() and D.max() < 1 and abs(D.mean()) < 0.04
    assert all(t["Q"][0, 0] == 1 and t["R"][0, 0] == 1 and t["mu1"][0, 0] == 0
               and t["V1"][0, 0] == 1 for t in init)
    th = ldsr_amd.pack_theta(init[0], 3, 2)
    back = ldsr_amd.unpack_theta(th, 3, 2)
    assert all(np.array_equal(back[k], init[0][k]) for k in back)


def test_synthetic_generator_is_shard_independent():
    from ldsr_amd import synth
    a = synth.make_init_packed(1, 2, 100, seed=1)
    b = synth.make_init_packed(1, 2, 40, seed=1, first=60)
    assert np.array_equal(a[60:], b)
    y, u, v = synth.make_series(300, 1, 2, mask="paleo")
    assert np.isnan(y[:270]).all() and np.isfinite(y[270:]).all()
    assert abs(np.nanmean(y)) < 1e-12


def test_r_compatible_uniforms_match_published_r_draws():
    """set.seed(k); runif(3) in R (widely published first draws of the default Mersenne-Twister)."""
    from ldsr_amd.rrng import RUniform, make_init_packed_r
    np.testing.assert_allclose(RUniform(1).runif(3), [0.2655087, 0.3721239, 0.5728534], atol=5e-8)
    np.testing.assert_allclose(RUniform(42).runif(3), [0.9148060, 0.9370754, 0.2861395], atol=5e-8)
    np.testing.assert_allclose(RUniform(123).runif(3), [0.2875775, 0.7883051, 0.4089769], atol=5e-8)
    th = make_init_packed_r(2, 1, 2, r_seed=1)
    # draw order per restart: A, B[0..p-1] on (-1,1), C, D on (-1,1)   (R/LDS_reconstruction.R:18-21)
    u = RUniform(1).unif_rand(10)
    assert th[0, 0] == u[0] and th[0, 1] == -1 + 2 * u[1] and th[0, 2] == -1 + 2 * u[2]
    assert th[0, 3] == u[3] and th[0, 4] == -1 + 2 * u[4] and th[1, 0] == u[5]
    assert np.all(th[:, 5:] == [1, 1, 0, 1])


def test_make_Z_and_metrics_follow_the_reference():
    from ldsr_amd import cv
    obs = np.arange(46, dtype=float)
    Z = cv.make_Z(obs, nRuns=30, frac=0.25, contiguous=True, rng=np.random.default_rng(0))
    assert len(Z) == 30 and all(len(z) == 12 for z in Z)          # k+1 = floor(46*.25)+1 points
    assert all(np.all(np.diff(z) == 1) for z in Z) and max(z[-1] for z in Z) <= 45
    Z1 = cv.make_Z(obs, frac=1)
    assert len(Z1) == 46
    rng = np.random.default_rng(1)
    y = rng.normal(5, 1, 40)
    yhat = y + rng.normal(0, 0.3, 40)
    assert cv.NSE(y, y) == 1.0 and cv.RE(y, y, 0.0) == 1.0 and abs(cv.KGE(y, y) - 1.0) < 1e-12
    assert abs(cv.corr(y, yhat) - np.corrcoef(y, yhat)[0, 1]) < 1e-12
    m = cv.calculate_metrics(yhat, y, np.arange(10, 20))
    assert set(m) == {"R2", "RE", "CE", "nRMSE", "KGE"} and m["CE"] <= m["RE"] + 1e-12


def test_cv_grid_host_logic_with_oracle_engine():
    """Fold masking, per-fold selection and winner-fit extraction of cv_grid, with the CPU
    oracle standing in for the GPU engine (host logic only)."""
    from ldsr_amd import cv, synth
    from oracle import oracle as O

    T, p, q = 90, 1, 2
    y, u, v = synth.make_series(T, p, q, series_id=77, mask="paleo", n_tail=40)
    inst = np.arange(50, 90)
    Z = [np.arange(0, 5), np.arange(20, 26), np.arange(34, 40)]

    def em_batch(Y, u_, v_, th0, cell_offsets=None, niter=1000, tol=1e-5):
        S = Y.shape[0]
        soc = np.repeat(np.arange(S), np.diff(cell_offsets)).astype(np.int32)
        U = np.repeat(np.ascontiguousarray(u_.T)[None], S, axis=0)
        V = np.repeat(np.ascontiguousarray(v_.T)[None], S, axis=0)
        th, lik, nit, st = O.em_batch(Y, U, V, soc, th0, niter, tol, n_threads=2)
        return {"theta": th, "lik": lik, "n_iter": nit, "status": st}

    def smooth_batch(Y, u_, v_, th, cell_offsets=None):
        fits = [O.kalman_smoother(Y[f], u_, v_, th[f]) for f in range(Y.shape[0])]
        return {k: np.stack([f_[k] for f_ in fits]) for k in "XYVJ"}

    eng = {"em_batch": em_batch, "smooth_batch": smooth_batch,
           "select": lambda l, t, p_, q_: O.select(l, t[:, 1 + p_])}
    r = cv.cv_grid(y, u, v, inst, Z, num_restarts=4, niter=30, tol=1e-5, seed=3, mu=2.5, engine=eng)
    assert r["Ycv"].shape == (3, 40) and np.all(np.isfinite(r["Ycv"]))
    # fold 1 by hand
    y1 = y.copy()
    y1[inst[Z[1]]] = np.nan
    th0 = synth.make_init_packed(p, q, 12, seed=3)[4:8]
    best, best_lik = None, -np.inf
    for t0 in th0:
        f = O.lds_em(y1, u, v, t0, 30, 1e-5)
        if f["theta"][1 + p] > 0 and f["lik"] > best_lik:
            best, best_lik = f, f["lik"]
    np.testing.assert_allclose(r["Ycv"][1], best["fit"]["Y"][inst] + 2.5, rtol=1e-12)
    m = cv.calculate_metrics(r["Ycv"][1], y[inst] + 2.5, Z[1])
    assert np.isfinite(list(m.values())).all()


def test_library_holds_exactly_the_kernels_a_plan_can_return():
    """Round 3's library had grown to 44 MB: chunk length, lanes per cell, schedule, lead and image
    form are template parameters, and members no launch plan could select were compiled too (LDS
    forms of images that do not fit the LDS, global-image forms of images that do, two-cells-per-wave
    members whose image and strips exceed 160 KiB).  Now the launchers instantiate with the SAME
    constexpr predicates the plans use (scan_uses_gimg, pair_member_fits) and
    ldsr_kernel_inventory() lists the result; this test enumerates the plans -- ldsr_em_plan for
    every T of the supported domain, every padded width, both schedules, AUTO and the explicit
    algorithms; ldsr_em_plan_lead over the tails of a closed-form lead; ldsr_smooth_plan for the FIT
    forms -- and asserts reachable == compiled."""
    import ctypes as C
    from ldsr_amd import _lib
    L = _lib.lib()
    n = L.ldsr_kernel_inventory(None, 0)
    buf = C.create_string_buffer(n)
    L.ldsr_kernel_inventory(buf, n)
    compiled = set(buf.value.decode().split())            # names contain ", ": re-join below
    compiled = set(l for l in buf.value.decode().split("\n") if l)
    assert len(compiled) > 1000 and all(k.startswith(("em_scan_kernel<", "em_pair_kernel<")) for k in compiled)

    reach = set()
    name = C.create_string_buffer(160)
    widths = (1, 2, 4, 8)
    for T in range(2, 8193):
        for p in widths:
            for q in widths:
                if L.ldsr_smooth_plan(T, p, q, name, 160) == 2:
                    reach.add(name.value.decode())
                for tol in (0.0, 1e-5):
                    for algo in (0, 2, 3, 4):
                        if (algo in (3, 4) and (T > 1024 or p > 4 or q > 4)) or (algo == 4 and T > 512):
                            continue                       # (outside the family: the plan says -1)
                        a = L.ldsr_em_plan(T, p, q, 100, tol, algo, name, 160)
                        if a in (2, 3, 4):
                            reach.add(name.value.decode())
    # closed-form leads: the tail is max(T - lead, 80) rounded up to a multiple of 16, at most 512 steps
    for T in (600, 700, 813, 1000, 1024, 1200, 1536, 2000, 3000, 4000, 8000, 8192):
        for tail in range(64, 529, 8):
            lead = T - tail
            if lead < 1:
                continue
            for p in widths:
                for q in widths:
                    for tol in (0.0, 1e-5):
                        a = L.ldsr_em_plan_lead(T, p, q, 100, tol, 0, lead, name, 160)
                        if a in (2, 3, 4):
                            reach.add(name.value.decode())
    # AUTO reports the member a device-filling launch gets; smaller launches of a lead fall back from
    # four to two cells per wave (same tails, LPC = 32), which the explicit plans above do not cover:
    # the LEAD forms at two cells per wave exist for every tail a four-cell form exists for
    missing = sorted(reach - compiled)
    assert not missing, missing[:10]
    extra = sorted(compiled - reach)
    lead32 = [k for k in extra if k.startswith("em_pair_kernel<") and k.endswith(", true>") and ", 32, " in k]
    assert sorted(set(extra) - set(lead32)) == [], sorted(set(extra) - set(lead32))[:20]
    # ... and those are reachable through smaller launches: the tail plan at lpc = 32 is the same rule
    for k in lead32:
        PP, QQ, Lc, lpc = [int(x) for x in k[len("em_pair_kernel<"):].split(", ")[:4]]
        assert lpc == 32 and 3 <= Lc <= 16
