"""ctypes wrapper of oracle/_build/libba_oracle.so (oracle/ba_cpu.cpp) — TEST INFRASTRUCTURE and the timed
single-thread CPU baseline of bench.py.  Uses the POD struct definitions of the product's ctypes binding
(they mirror include/vinsgpu.h); no product arithmetic is involved."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libba_oracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
        _lib = C.CDLL(_LIB)
    return _lib


def optimize(prob, margin_flag=2, packed=None):
    """CPU Estimator::optimization() restatement: returns (state, summary, new_prior)."""
    from vins_mono_amd import ba
    L = lib()
    p = packed if packed is not None else ba.PackedProblem(prob)
    out = ba._Out(p.K, p.L, p.has_relo, margin_flag != ba.VG_MARGIN_NONE)
    sm = ba.Summary()
    L.oracle_ba_optimize.argtypes = [C.POINTER(ba.Problem), C.c_int, C.POINTER(ba.State), C.POINTER(ba.Summary), C.POINTER(ba.Prior)]
    rc = L.oracle_ba_optimize(C.byref(p.struct), int(margin_flag), C.byref(out.state), C.byref(sm),
                              C.byref(out.prior) if out.prior is not None else None)
    assert rc == 0
    return out.state_dict(p.has_relo), ba.summary_dict(sm), out.prior_dict()


def time_optimize(packed_list, margin_flags, repeats=1):
    """Wall time (s) of running optimize over a list of pre-packed problems, single thread."""
    import time
    from vins_mono_amd import ba
    L = lib()
    L.oracle_ba_optimize.argtypes = [C.POINTER(ba.Problem), C.c_int, C.POINTER(ba.State), C.POINTER(ba.Summary), C.POINTER(ba.Prior)]
    outs = [ba._Out(p.K, p.L, p.has_relo, True) for p in packed_list]
    sm = ba.Summary()
    t0 = time.perf_counter()
    for _ in range(repeats):
        for p, o, mf in zip(packed_list, outs, margin_flags):
            L.oracle_ba_optimize(C.byref(p.struct), int(mf), C.byref(o.state), C.byref(sm), C.byref(o.prior))
    return time.perf_counter() - t0


def triangulate(Ps, Rs, tic, ric, start, nobs, obs_off, points, init_depth=5.0):
    """FeatureManager::triangulate, C++ restatement (Gram-matrix Jacobi) — see ba_numpy.triangulate for the SVD one."""
    import numpy as np
    Ps = np.ascontiguousarray(Ps, np.float64); Rs = np.ascontiguousarray(Rs, np.float64).reshape(-1, 9)
    st = np.ascontiguousarray(start, np.int32); nb = np.ascontiguousarray(nobs, np.int32); oo = np.ascontiguousarray(obs_off, np.int32)
    pts = np.ascontiguousarray(points, np.float64)
    out = np.zeros(max(len(st), 1))
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    f = lib().oracle_triangulate
    f.restype = None
    f(len(Ps), Ps.ctypes.data_as(dp), Rs.ctypes.data_as(dp), np.ascontiguousarray(tic, np.float64).ctypes.data_as(dp),
      np.ascontiguousarray(ric, np.float64).ctypes.data_as(dp), len(st), st.ctypes.data_as(ip), nb.ctypes.data_as(ip), oo.ctypes.data_as(ip),
      pts.ctypes.data_as(dp), C.c_double(init_depth), out.ctypes.data_as(dp))
    return out[:len(st)]
