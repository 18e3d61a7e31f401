"""GPU parity of the read-ahead rings (em_scan_impl.h / em_pair_impl.h: SPF) and of the pipelined walk of the
closed-form lead (em_pair_impl.h lead_walk) against the CPU oracle -- the index cases of the rings: every chunk
length of the short-chunk scan kernel with an odd and an even number of values per step (the odd value of a step
shares a 16-byte pair with its neighbour's: slot parity), ring depths 1 and 2, chunks with and without the
predicated last step, fully observed series (the reversed F1) and masked ones (the forward F1); and leads whose
per-lane step counts hit every remainder of the walk's eight-step iteration.  Same bar as everywhere: identical
n_iter, then |d| <= 1e-6 |ref| + 1e-9."""
import numpy as np
import pytest

from conftest import parity_close

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-6, 1e-9
SCAN, PAIR, QUAD = 2, 3, 4


@pytest.fixture(scope="module")
def eng():
    import ldsr_amd
    return ldsr_amd


def _case(T, p, q, sid, mask):
    from ldsr_amd import synth
    y, u, v = synth.make_series(T, p, q, series_id=sid)
    y = y.copy()
    if mask == "holes":
        y[::5] = np.nan
        y[1] = np.nan
        y[T - 1] = np.nan
    return y, u, v


def _check(eng, y, u, v, th0, algo, niter, tol, what, **kw):
    from oracle import oracle as O
    ref_th, ref_lik, ref_it, _ = O.em_batch(y[None], np.ascontiguousarray(u.T[None]), np.ascontiguousarray(v.T[None]),
                                            np.zeros(len(th0), np.int32), th0, niter, tol, n_threads=16)
    r = eng.em_batch(y, u, v, th0, niter=niter, tol=tol, algo=algo, **kw)
    ok = np.isfinite(ref_lik)
    assert np.array_equal(r["n_iter"][ok], ref_it[ok]), what
    assert parity_close(r["lik"][ok], ref_lik[ok], RTOL, ATOL), what
    assert parity_close(r["theta"][ok], ref_th[ok], RTOL, ATOL), what


# (chunk length of the scan kernel, a T that leaves lanes with and without the predicated step)
SCAN_T = [(2, 101), (3, 150), (4, 213), (6, 333), (8, 470), (10, 601), (12, 707), (13, 813), (14, 850), (15, 931), (16, 1000)]


@pytest.mark.parametrize("L,T", SCAN_T)
@pytest.mark.parametrize("p,q", [(1, 1), (1, 2), (2, 2), (1, 4), (2, 4), (4, 4)])     # K = 3, 4, 5, 6, 7, 9 values per step
def test_scan_kernel_rings(eng, L, T, p, q):
    from ldsr_amd import synth
    th0 = synth.make_init_packed(p, q, 9, seed=L * 10 + p + q)
    for mask in ("dense", "holes"):
        y, u, v = _case(T, p, q, 300 + L, mask)
        _check(eng, y, u, v, th0, SCAN, 40, 1e-5, "scan L=%d T=%d (%d,%d) %s" % (L, T, p, q, mask))


@pytest.mark.parametrize("p,q", [(7, 7), (4, 8), (1, 8)])       # wide inputs: the ring only with chunks of <= 4 steps
@pytest.mark.parametrize("T", [85, 150, 250, 300])
def test_scan_kernel_rings_wide(eng, T, p, q):
    from ldsr_amd import synth
    th0 = synth.make_init_packed(p, q, 5, seed=T + p)
    for mask in ("dense", "holes"):
        y, u, v = _case(T, p, q, 400 + T, mask)
        _check(eng, y, u, v, th0, SCAN, 30, 1e-5, "scan T=%d (%d,%d) %s" % (T, p, q, mask))


@pytest.mark.parametrize("T", [97, 130, 161, 200, 230, 270, 300, 333, 365, 400, 430, 460, 490, 512])   # L = 4 .. 16 at 32 lanes
@pytest.mark.parametrize("p,q", [(1, 1), (2, 2), (2, 4), (4, 4)])
def test_pair_family_rings(eng, T, p, q):
    from ldsr_amd import synth
    th0 = synth.make_init_packed(p, q, 13, seed=T + q)
    for mask in ("dense", "holes"):
        y, u, v = _case(T, p, q, 500 + T, mask)
        _check(eng, y, u, v, th0, PAIR, 40, 1e-5, "pair T=%d (%d,%d) %s" % (T, p, q, mask))
        if T <= 256:                                       # four cells per wave: chunks of <= 16 steps
            _check(eng, y, u, v, th0, QUAD, 40, 1e-5, "quad T=%d (%d,%d) %s" % (T, p, q, mask))


def _last_kernel():
    import ctypes as C
    from ldsr_amd import _lib
    buf = C.create_string_buffer(160)
    assert _lib.lib().ldsr_last_em_kernel(0, buf, 160) == 0
    return buf.value.decode()


@pytest.mark.parametrize("lead", [192, 200, 208, 216, 224, 232, 240, 250, 263, 277, 290, 301, 333, 352])
@pytest.mark.parametrize("p", [1, 2])
def test_lead_walk_remainders(eng, lead, p):
    """A lead of `lead` steps is walked by 16 lanes (four cells per wave): 12 .. 22 steps per lane, the last lane fewer --
    every remainder of the walk's eight- (p = 1) and four-step (p = 2) iterations, static and work-queue schedule."""
    from ldsr_amd import synth
    q, tail = 2, 90
    T = lead + tail
    y, u, v = synth.make_series(T, p, q, series_id=600 + lead)
    y = y.copy()
    y[:lead] = np.nan
    th0 = synth.make_init_packed(p, q, 8192, seed=lead)
    # (AUTO takes the LEAD form from ~1536 cells with a fixed iteration count, from 7/8 of a device's worth with tol > 0)
    for n, niter, tol in ((2048, 30, 0.0), (8192, 60, 1e-5)):
        _check(eng, y, u, v, th0[:n], 0, niter, tol, "lead=%d p=%d tol=%g" % (lead, p, tol))
        assert _last_kernel().endswith("true>"), _last_kernel()       # a LEAD form ran


# half-stored long chunks (17 .. 32 steps): F1, F2, the first half's re-run and both segments of B2 read through the ring;
# wide inputs where the registers allow (L = 20: padded p + q <= 12, L = 24: <= 10), the others without it
@pytest.mark.parametrize("T", [1100, 1280, 1400, 1536, 1700, 1792, 1900, 2048])            # L = 20, 20, 24, 24, 28, 28, 32, 32
@pytest.mark.parametrize("p,q", [(1, 1), (1, 2), (2, 2), (2, 4), (4, 4), (1, 8), (7, 2), (4, 8)])
def test_scan_kernel_rings_long_chunks(eng, T, p, q):
    from ldsr_amd import synth
    th0 = synth.make_init_packed(p, q, 5, seed=T + p + q)
    for mask in ("dense", "holes"):
        y, u, v = _case(T, p, q, 700 + T, mask)
        _check(eng, y, u, v, th0, SCAN, 25, 1e-5, "scan T=%d (%d,%d) %s" % (T, p, q, mask))


# the global image of the multi-wave cells (16-byte buffer loads two or three steps ahead, four slots for the odd values' pairs)
@pytest.mark.parametrize("T", [2100, 3000, 3585, 4097, 5555, 8192])
@pytest.mark.parametrize("p,q", [(1, 1), (1, 2), (2, 2), (3, 3)])
def test_scan_kernel_rings_global_image(eng, T, p, q):
    from ldsr_amd import synth
    th0 = synth.make_init_packed(p, q, 3, seed=T + q)
    for mask in ("dense", "holes"):
        y, u, v = _case(T, p, q, 800 + T % 97, mask)
        _check(eng, y, u, v, th0, SCAN, 12, 1e-5, "scan T=%d (%d,%d) %s" % (T, p, q, mask))
