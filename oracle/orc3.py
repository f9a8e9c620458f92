"""ctypes front of oracle/cfdh_oracle3.c: the C restatement of the tetrahedral element tensors, with the call signature of
`np_twin_nd.element_tensors` so that `np_twin_nd.Problem` can assemble with either.  TEST INFRASTRUCTURE ONLY (see the C
file's header): imported by tests/ only."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import orc as _orc

_L = None


def _lib():
    global _L
    if _L is None:
        L = C.CDLL(_orc.build())
        dp, ip, bp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_ubyte)
        L.orc3_element_tensors.argtypes = [C.c_int, dp, ip, dp, dp, dp, dp, bp, C.c_double, C.c_double, C.c_double, C.c_double, dp,
                                           C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_double, dp, dp]
        _L = L
    return _L


def element_tensors(x, cells, u, un, p, prm, facet_flags=None, want_jac=True, un2=None):
    """Same contract as np_twin_nd.element_tensors for d = 3: (Fe [nc,16], Je [nc,16,16] or None)."""
    assert x.shape[1] == 3
    L = _lib()
    dp, ip, bp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_ubyte)
    xa = np.ascontiguousarray(x, dtype=np.float64)
    ca = np.ascontiguousarray(cells, dtype=np.int32)
    ua, una, pa = (np.ascontiguousarray(a, dtype=np.float64) for a in (u, un, p))
    u2 = None if un2 is None else np.ascontiguousarray(un2, dtype=np.float64)
    ff = None if facet_flags is None else np.ascontiguousarray(facet_flags, dtype=np.uint8)
    nc = len(ca)
    Fe = np.empty((nc, 16))
    Je = np.empty((nc, 16, 16)) if want_jac else None
    f3 = np.zeros(3)
    f3[: len(prm.f)] = prm.f
    rc = L.orc3_element_tensors(nc, xa.ctypes.data_as(dp), ca.ctypes.data_as(ip), ua.ctypes.data_as(dp), una.ctypes.data_as(dp),
                                u2.ctypes.data_as(dp) if u2 is not None else None, pa.ctypes.data_as(dp),
                                ff.ctypes.data_as(bp) if ff is not None else None, prm.dt, prm.rho, prm.mu, prm.mu_facet,
                                f3.ctypes.data_as(dp), prm.theta, prm.a0, prm.a1, prm.a2, int(prm.ds_terms), prm.beta_backflow,
                                Fe.ctypes.data_as(dp), Je.ctypes.data_as(dp) if Je is not None else None)
    assert rc == 0
    return Fe, Je
