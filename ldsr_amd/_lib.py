"""ctypes loader for libldsr_hip.so (built in-tree by __graft_entry__.build() or
`make -C ldsr_amd/csrc`).  There is NO CPU fallback: if the library is missing or does not
export a symbol of include/ldsr_hip.h, importing callers fail loudly."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# LDSR_HIP_SO overrides the library path (A/B runs of two builds on the same GPU box)
SO_PATH = os.environ.get("LDSR_HIP_SO") or os.path.join(_HERE, "libldsr_hip.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_vp = C.c_void_p

# name -> (restype, argtypes); mirrors include/ldsr_hip.h one to one
SIGNATURES = {
    "ldsr_last_error": (C.c_char_p, []),
    "ldsr_version": (C.c_char_p, []),
    "ldsr_source_hash": (C.c_char_p, []),
    "ldsr_device_count": (C.c_int, []),
    "ldsr_shutdown": (None, []),
    "ldsr_em_batch": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp,
                                C.c_int, _ip, _dp, C.c_int, C.c_double, C.c_int, _dp, _dp, _ip,
                                _ip, _dp]),
    "ldsr_em_batch_multi": (C.c_int, [C.c_int, _ip, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp,
                                      C.c_int, _ip, _dp, C.c_int, C.c_double, C.c_int, _dp, _dp,
                                      _ip, _ip, _dp]),
    "ldsr_em_restart_grid": (C.c_int, [C.c_int, _ip, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp,
                                       C.c_int, _ip, _dp, C.c_int, C.c_double, C.c_int, _dp, _dp,
                                       _ip, _ip, _ip, _dp, _dp, _ip, _dp, _dp, _dp, _dp, _dp]),
    "ldsr_em_restart_groups": (C.c_int, [C.c_int, _ip, C.c_int, _vp, C.c_int, C.c_double, C.c_int]),
    "ldsr_em_plan": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_char_p,
                               C.c_size_t]),
    "ldsr_em_plan_lead": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int,
                                    C.c_char_p, C.c_size_t]),
    "ldsr_em_batch_device_lead": (C.c_int, [C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp,
                                            _vp, C.c_int, _ip, _vp, C.c_int, C.c_double, C.c_int, _vp,
                                            _vp, _vp, _vp, _vp, _vp, C.c_size_t, C.c_int]),
    "ldsr_smooth_plan": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_size_t]),
    "ldsr_kernel_inventory": (C.c_size_t, [C.c_char_p, C.c_size_t]),
    "ldsr_last_em_kernel": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t]),
    "ldsr_set_interrupt_callback": (C.c_int, [_vp, _vp]),
    "ldsr_em_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ldsr_em_batch_device": (C.c_int, [C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp,
                                       _vp, C.c_int, _ip, _vp, C.c_int, C.c_double, C.c_int, _vp,
                                       _vp, _vp, _vp, _vp, _vp, C.c_size_t]),
    "ldsr_smooth_batch": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp,
                                    C.c_int, _ip, _dp, C.c_int, _dp, _dp, _dp, _dp, _dp]),
    "ldsr_mstep_batch": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp,
                                   C.c_int, _ip, _dp, _dp, _dp, _dp, _ip]),
    "ldsr_propagate_batch": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp,
                                       C.c_int, _ip, _dp, C.c_int, _dp, _dp, _dp, _dp]),
    "ldsr_penalized_lik_batch": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp,
                                           C.c_int, _ip, _dp, C.c_double, _dp]),
    "ldsr_profile_enable": (None, [C.c_int]),
    "ldsr_profile_collect": (C.c_int, [_dp, _ip]),
    "ldsr_select_restart": (C.c_int, [C.c_int, _dp, _dp, C.c_int, C.c_int]),
    "ldsr_metric_nse": (C.c_double, [C.c_int, _dp, _dp]),
    "ldsr_metric_nrmse": (C.c_double, [C.c_int, _dp, _dp, C.c_double]),
    "ldsr_metric_corr": (C.c_double, [C.c_int, _dp, _dp]),
    "ldsr_metric_kge": (C.c_double, [C.c_int, _dp, _dp]),
    "ldsr_metric_re": (C.c_double, [C.c_int, _dp, _dp, C.c_double]),
}



class Group(C.Structure):
    """struct ldsr_group of include/ldsr_hip.h (one member of a heterogeneous ensemble)."""
    _fields_ = [("n_series", C.c_int), ("T", C.c_int), ("p", C.c_int), ("q", C.c_int),
                ("shared_uv", C.c_int),
                ("y", _dp), ("u", _dp), ("v", _dp), ("cell_offsets", _ip), ("theta0", _dp),
                ("theta_all", _dp), ("lik_all", _dp), ("n_iter_all", _ip), ("status_all", _ip),
                ("winner", _ip), ("theta_w", _dp), ("lik_w", _dp), ("n_iter_w", _ip),
                ("liks_w", _dp), ("X", _dp), ("Y", _dp), ("V", _dp), ("J", _dp), ("rc", C.c_int)]


_LIB = None


class LdsrError(RuntimeError):
    pass


def tree_source_hash():
    """The hash `make` stamps into the library (ldsr_amd/csrc/Makefile SRCHASH), recomputed from the
    tree next to this file; None when the sources are not there (an installed binary)."""
    import glob
    import hashlib
    csrc = os.path.join(_HERE, "csrc")
    names = sorted(os.path.basename(f) for pat in ("*.hip", "*.h", "*.inc") for f in glob.glob(os.path.join(csrc, pat)))
    files = [os.path.join(csrc, n) for n in names if n != "source_hash.h"]
    files += [os.path.join(csrc, "Makefile"), os.path.join(_HERE, "..", "include", "ldsr_hip.h")]
    if not names or not all(os.path.exists(f) for f in files):
        return None
    h = hashlib.sha256()
    for f in files:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(SO_PATH):
            raise LdsrError(
                "libldsr_hip.so not found at %s -- build it with `python -c 'import "
                "__graft_entry__ as g; g.build()'` or `make -C ldsr_amd/csrc`; there is no CPU "
                "fallback" % SO_PATH)
        L = C.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        # a prebuilt library that does not match the sources it sits next to must not pass for them
        # (LDSR_HIP_SO: an explicitly chosen other build, e.g. the A/B runs of tools/ab.sh)
        want = None if os.environ.get("LDSR_HIP_SO") else tree_source_hash()
        got = L.ldsr_source_hash().decode()
        if want is not None and got != want:
            raise LdsrError("%s was built from other sources (library %s, tree %s): rebuild it with "
                            "`make -C ldsr_amd/csrc`" % (SO_PATH, got, want))
        _LIB = L
    return _LIB


def check(rc):
    if rc != 0:
        raise LdsrError("libldsr_hip error %d: %s" % (rc, lib().ldsr_last_error().decode()))
