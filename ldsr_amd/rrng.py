"""R-compatible uniform random numbers: R's default generator is
Mersenne-Twister seeded by set.seed() through an LCG scrambler (R sources, src/main/RNG.c:
RNG_Init / MT_genrand / fixup).  With it `make_init(p, q, n, r_seed=k)` reproduces what
`set.seed(k); make_init(p, q, n)` draws in R (reference R/LDS_reconstruction.R:14-30), so a
reconstruction can be replayed without shipping the init list across the boundary.

R is not available in this image; the implementation is pinned in tests by the widely
published first draws of set.seed(1), set.seed(42) and set.seed(123)."""
import numpy as np

_I2_32M1 = 2.328306437080797e-10      # 1/(2^32 - 1), R's fixup constant
_SCALE = 2.3283064365386963e-10       # 2^-32


class RUniform:
    def __init__(self, seed):
        s = np.uint32(int(seed) & 0xFFFFFFFF)
        with np.errstate(over="ignore"):
            for _ in range(50):                       # initial scrambling
                s = np.uint32(69069) * s + np.uint32(1)
            key = np.empty(625, dtype=np.uint32)
            for j in range(625):
                s = np.uint32(69069) * s + np.uint32(1)
                key[j] = s
        # FixupSeeds: dummy[0] = mti = 624 -> regenerate on first use; mt = dummy + 1
        self._bg = np.random.MT19937()
        self._bg.state = {"bit_generator": "MT19937", "state": {"key": key[1:], "pos": 624}}

    def unif_rand(self, n=1):
        x = self._bg.random_raw(n).astype(np.float64) * _SCALE
        x = np.where(x <= 0.0, 0.5 * _I2_32M1, x)
        x = np.where(1.0 - x <= 0.0, 1.0 - 0.5 * _I2_32M1, x)
        return x

    def runif(self, n, a=0.0, b=1.0):
        return a + (b - a) * self.unif_rand(n)


def make_init_packed_r(p, q, n, r_seed):
    """Packed [n, 6+p+q] thetas drawn exactly as R's make_init after set.seed(r_seed):
    per restart runif(1), runif(p,-1,1), runif(1), runif(q,-1,1)."""
    g = RUniform(r_seed)
    th = np.empty((n, 6 + p + q))
    for r in range(n):
        th[r, 0] = g.runif(1)[0]
        th[r, 1:1 + p] = g.runif(p, -1.0, 1.0)
        th[r, 1 + p] = g.runif(1)[0]
        th[r, 2 + p:2 + p + q] = g.runif(q, -1.0, 1.0)
    th[:, 2 + p + q] = 1.0
    th[:, 3 + p + q] = 1.0
    th[:, 4 + p + q] = 0.0
    th[:, 5 + p + q] = 1.0
    return th
