"""Synthetic workload generator shared by bench.py and the tests.

Counter-based: every number is a pure function of (seed, stream, index) through SplitMix64,
so a cell's initial theta does not depend on how cells are sharded over GPUs.

Series follow the generative model the reference simulates from (R/stochastics.R:34-37):
    x_{t+1} = A x_t + B u_t + N(0,Q),   y_t = C x_t + D v_t + N(0,R)
with u, v ~ iid N(0,1) and truth A=0.8, B=0.3, C=0.5, D_j=0.1(-1)^j, Q=0.5, R=0.1, x_1=0.
y is centred on its observed mean (R/LDS_reconstruction.R:180-182).  Initial thetas follow
make_init's distribution (R/LDS_reconstruction.R:14-30)."""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def uniform(seed, stream, n):
    """n doubles in [0,1): index i of (seed, stream)."""
    with np.errstate(over="ignore"):
        base = _splitmix64(np.uint64(seed) ^ _splitmix64(np.uint64(stream)))
        idx = np.arange(n, dtype=np.uint64)
        z = _splitmix64(base + idx * np.uint64(0x9E3779B97F4A7C15))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def normal(seed, stream, n):
    u1 = uniform(seed, 2 * stream, n)
    u2 = uniform(seed, 2 * stream + 1, n)
    return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)


def make_series(T, p, q, series_id=0, mask="dense", n_tail=None):
    """Returns y [T] (NaN = missing), u [p,T], v [q,T].  mask: 'dense' | 'paleo' (only the
    last n_tail = T//10 steps observed, as R/LDS_reconstruction.R:181-183 pads with NA)."""
    seed = 20260101 + series_id
    u = normal(seed, 1, p * T).reshape(T, p).T.copy()
    v = normal(seed, 2, q * T).reshape(T, q).T.copy()
    w = normal(seed, 3, T) * np.sqrt(0.5)
    e = normal(seed, 4, T) * np.sqrt(0.1)
    A, B, Cc = 0.8, np.full(p, 0.3), 0.5
    D = 0.1 * (-1.0) ** np.arange(q)
    x = np.empty(T)
    x[0] = 0.0
    for t in range(T - 1):
        x[t + 1] = A * x[t] + B @ u[:, t] + w[t]
    y = Cc * x + D @ v + e
    if mask == "paleo":
        n_tail = T // 10 if n_tail is None else n_tail
        y[:T - n_tail] = np.nan
    elif mask != "dense":
        raise ValueError(mask)
    y = y - np.nanmean(y)
    return y, u, v


def make_init_packed(p, q, n, seed=1, first=0):
    """make_init's distribution (A~U(0,1), B~U(-1,1)^p, C~U(0,1), D~U(-1,1)^q, Q=R=1, mu1=0,
    V1=1), packed [n, 6+p+q]; restart r uses stream `first + r` so shards agree."""
    P = 6 + p + q
    th = np.empty((n, P))
    for r in range(n):
        z = uniform(seed, 1000 + first + r, 2 + p + q)
        th[r, 0] = z[0]
        th[r, 1:1 + p] = 2.0 * z[1:1 + p] - 1.0
        th[r, 1 + p] = z[1 + p]
        th[r, 2 + p:2 + p + q] = 2.0 * z[2 + p:2 + p + q] - 1.0
    th[:, 2 + p + q] = 1.0
    th[:, 3 + p + q] = 1.0
    th[:, 4 + p + q] = 0.0
    th[:, 5 + p + q] = 1.0
    return th
