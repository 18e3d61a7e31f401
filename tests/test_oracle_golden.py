"""Pin the CPU oracle against the reference's own known-answer test
(/root/reference/tests/testthat/test-LDS-EM.R:21-41, tolerance 1e-6 as in the reference)
and against 17-digit regression values recorded by SURVEY.md section 8(c)."""
import numpy as np
import pytest

from oracle import oracle as O


def _th(th, p=7, q=7):
    return O.unpack_theta(th, p, q)


def test_first_two_iterations_match_reference_known_answers(p1case):
    c = p1case
    s1 = O.kalman_smoother(c["y"], c["u"], c["v"], c["theta0"])
    t1 = O.mstep(c["y"], c["u"], c["v"], s1)
    s2 = O.kalman_smoother(c["y"], c["u"], c["v"], t1)
    t2 = O.mstep(c["y"], c["u"], c["v"], s2)
    tol = 1e-6   # testthat expect_equal(tolerance = 1e-6)
    assert s1["lik"] == pytest.approx(-11.678657, rel=tol)
    assert s1["X"][0] == pytest.approx(1.293356, rel=tol)
    assert s1["X"][84] == pytest.approx(-0.987671, rel=tol)
    assert _th(t1)["A"] == pytest.approx(0.606066, rel=tol)
    assert _th(t1)["C"] == pytest.approx(-0.005995, abs=1e-6)
    assert _th(t1)["Q"] == pytest.approx(3.640236, rel=tol)
    assert s2["lik"] == pytest.approx(-0.114224, abs=1e-6)
    assert _th(t2)["A"] == pytest.approx(0.603945, rel=tol)
    assert _th(t2)["C"] == pytest.approx(-0.012004, abs=1e-6)
    assert _th(t2)["Q"] == pytest.approx(3.644322, rel=tol)


def test_convergence_matches_reference_known_answer(p1case):
    c = p1case
    fit = O.lds_em(c["y"], c["u"], c["v"], c["theta0"], 100, 1e-5)
    assert len(fit["liks"]) == 68
    assert fit["lik"] == pytest.approx(-0.039093, abs=1e-6)
    d = np.abs(np.diff(fit["liks"]))[-3:]
    assert d[0] > 1e-5 and d[1] < 1e-5 and d[2] < 1e-5     # the stop rule of src/EM.cpp:272


def test_seventeen_digit_regression_pins(p1case):
    c = p1case
    s1 = O.kalman_smoother(c["y"], c["u"], c["v"], c["theta0"])
    assert s1["lik"] == pytest.approx(-11.678656588814256, rel=1e-12)
    assert s1["V"][0] == pytest.approx(0.76393202250021031, rel=1e-12)
    assert s1["J"][0] == pytest.approx(0.33333333333333337, rel=1e-12)
    t1 = _th(O.mstep(c["y"], c["u"], c["v"], s1))
    assert t1["R"] == pytest.approx(0.074517883051681055, rel=1e-12)
    assert t1["mu1"] == pytest.approx(1.293355756908821, rel=1e-12)
    fit = O.lds_em(c["y"], c["u"], c["v"], c["theta0"], 100, 1e-5)
    th = _th(fit["theta"])
    assert th["A"] == pytest.approx(0.59893129323481986, rel=1e-11)
    assert th["C"] == pytest.approx(-0.046724849851945006, rel=1e-11)
    assert th["Q"] == pytest.approx(7.1349600981551093, rel=1e-11)
    assert th["R"] == pytest.approx(0.043075629153553945, rel=1e-11)


@pytest.mark.parametrize("case,n_it,lik,A", [
    ("u_absent", 62, -0.0691245254151, 0.821083992367),
    ("v_absent", 100, -0.315132864108, 0.260408153182),
    ("nan_mask", 100, 0.106697962128, 0.721974756064),
    ("p2_q7", 44, -0.0603096124706, 0.789382169484),
])
def test_unpinned_branches_regression(p1case, case, n_it, lik, A):
    """Branches the reference only smoke-tests (absent u / v, NA in y, p != q)."""
    c = p1case
    y, u, v = c["y"].copy(), c["u"], c["v"]
    if case == "u_absent":
        u = None
    if case == "v_absent":
        v = None
    if case == "nan_mask":
        y[[0, 10, 11, 84]] = np.nan
    if case == "p2_q7":
        u = u[:2]
    p = 1 if u is None else u.shape[0]
    q = 1 if v is None else v.shape[0]
    th0 = np.concatenate([[0.5], np.full(p, 0.5), [0.5], np.full(q, 0.5), [1, 1, 1, 1.0]])
    fit = O.lds_em(y, u, v, th0, 100, 1e-5)
    assert len(fit["liks"]) == n_it
    assert fit["lik"] == pytest.approx(lik, rel=1e-9)
    assert fit["theta"][0] == pytest.approx(A, rel=1e-9)


def test_propagate_is_open_loop_forward_pass(p1case):
    c = p1case
    pr = O.propagate(c["theta0"], c["u"], c["v"], c["y"])
    # with R -> infinity the filter never updates, so Kalman prior == propagate
    A, B = 0.5, np.full(7, 0.5)
    x = np.empty(85)
    x[0] = 1.0
    for t in range(1, 85):
        x[t] = A * x[t - 1] + B @ c["u"][:, t - 1]
    np.testing.assert_allclose(pr["X"], x, rtol=1e-13)
    assert np.isfinite(pr["lik"])


def test_selection_rule():
    # R/LDS_reconstruction.R:50-58
    assert O.select([1.0, 3.0, 2.0], [1.0, -1.0, 1.0]) == 2     # best among C>0
    assert O.select([1.0, 3.0, 2.0], [-1.0, -1.0, -1.0]) == 1   # no C>0: which.max
    assert O.select([np.nan, 0.5], [1.0, 1.0]) == 1             # na.rm
    assert O.select([np.nan, np.nan], [1.0, -1.0]) == -1


def test_np_bundled_data_reaches_published_optimum(npcase, refdata):
    """Config 1 plumbing: the bundled NPlds theta is a fixed point neighbourhood:
    one E-step at NPlds$theta reproduces NPlds$lik (vignette value 0.8249...)."""
    c = npcase(1200)
    th = refdata["NPlds"]["theta"]
    theta = O.pack_theta(th["A"][0], th["B"], th["C"][0], th["D"], th["Q"][0], th["R"][0],
                         th["mu1"][0], th["V1"][0])
    s = O.kalman_smoother(c["y"], c["u"], c["v"], theta)
    assert s["lik"] == pytest.approx(refdata["NPlds"]["lik"][0], rel=1e-9)
