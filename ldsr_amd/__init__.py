"""ldsr_amd -- MI355X-native engine for ldsr's EM/Kalman restart path.

Only the hot path lives here: the HIP kernels + C ABI (csrc/, include/ldsr_hip.h) and the
host-side mirror of the reference's operator interface (api.py)."""
from .api import (ALGO_AUTO, ALGO_PAIR, ALGO_QUAD, ALGO_SCAN, ALGO_SERIAL, Kalman_smoother, LDS_EM,  # noqa: F401
                  LDS_EM_restart, Mstep, em_batch, em_restart_grid, ensemble_restart, make_init, pack_theta, penalized_likelihood,
                  propagate,
                  select_restart, smooth_batch, unpack_theta)
from . import cv, shard  # noqa: F401,E402
