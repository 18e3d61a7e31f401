import json
import os
import sys

import numpy as np
import pytest

try:        # PyTorch's bundled HIP runtime must initialise before libldsr_hip.so's (tests that hand
    import torch  # noqa: F401  torch device buffers to the C ABI); the reverse order leaves torch without a GPU
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The built libraries are git-ignored: build them once if a fresh checkout lacks them
    (hipcc cross-compiles gfx950 without a GPU; ~1.5 min)."""
    so = os.path.join(ROOT, "ldsr_amd", "libldsr_hip.so")
    oracle_so = os.path.join(ROOT, "oracle", "libldsr_oracle.so")
    if not (os.path.exists(so) and os.path.exists(oracle_so)):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def refdata():
    with open(os.path.join(ROOT, "tests", "golden", "reference_data.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def p1case(refdata):
    """Inputs of the reference's known-answer test (tests/testthat/test-LDS-EM.R:3-16):
    y = t(log(P1annual$Qa) - mean), uInst = vInst = t(P1pc[322:406]), fixed theta0."""
    obs = np.log(np.array(refdata["P1annual"]["Qa"]))
    y = obs - obs.mean()
    pc = np.array(refdata["P1pc"]["data"])          # 7 x 406
    uinst = np.ascontiguousarray(pc[:, 321:406])     # R rows 322..406
    theta0 = np.concatenate([[0.5], np.full(7, 0.5), [0.5], np.full(7, 0.5), [1, 1, 1, 1.0]])
    return {"y": y, "u": uinst, "v": uinst, "theta0": theta0, "p": 7, "q": 7}


@pytest.fixture(scope="session")
def npcase(refdata):
    """Nakhon Phanom bundled data as LDS_reconstruction builds it
    (R/LDS_reconstruction.R:164-183): y = log(Qa) - mean, NA outside the instrumental years;
    u = v = t(NPpc) for years 1200..2012."""
    qa = np.array(refdata["NPannual"]["Qa"])
    years = np.array(refdata["NPannual"]["year"])
    pcs = np.array(refdata["NPpc"]["data"])          # 3 x 813, years 1200..2012
    obs = np.log(qa)
    mu = obs.mean()

    def make(start_year):
        first = start_year - 1200
        u = np.ascontiguousarray(pcs[:, first:])
        T = u.shape[1]
        y = np.full(T, np.nan)
        i0 = years[0] - start_year
        y[i0:i0 + len(obs)] = obs - mu
        return {"y": y, "u": u, "v": u, "p": 3, "q": 3, "mu": mu}

    return make


def parity_close(a, b, rtol=1e-6, atol=1e-9):
    """SURVEY Appendix B criterion: |d| <= rtol*|ref| + atol, NaN == NaN."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    both_nan = np.isnan(a) & np.isnan(b)
    ok = np.abs(a - b) <= rtol * np.abs(b) + atol
    return bool(np.all(ok | both_nan))
