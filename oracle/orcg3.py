"""ctypes front of oracle/cfdh_oracle_gen3.c: the C restatement of the element tensors of Q1 hexahedra and P2 tetrahedra (and, as
a cross-check, P1 tetrahedra by quadrature), with the call signature of `np_twin_gen3.element_tensors` so that
`np_twin_gen3.Problem` can assemble with either.  TEST INFRASTRUCTURE ONLY (see the C file's header)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import orc as _orc
from .np_twin_gen3 import element as _element

_L = None


class _Params(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("dt", "rho", "mu", "muf")] + [("f", C.c_double * 3)] + \
               [(n, C.c_double) for n in ("theta", "a0", "a1", "a2", "beta")] + [("ds_terms", C.c_int32), ("pad", C.c_int32)]


def _lib():
    global _L
    if _L is None:
        L = C.CDLL(_orc.build())
        dp = C.POINTER(C.c_double)
        L.orcg3_element_tensors.argtypes = [C.c_int, C.c_int64, C.POINTER(C.c_int64), dp, dp, dp, dp, dp, C.POINTER(_Params),
                                            C.POINTER(C.c_uint16), C.c_int, dp, dp]
        L.orcg3_element_tensors.restype = None
        _L = L
    return _L


def element_tensors(etype, x, cells, u, un, p, prm, facet_flags=None, want_jac=True, un2=None):
    """Same contract as np_twin_gen3.element_tensors: (Fe [nc, 4 nloc], Je [nc, 4 nloc, 4 nloc] or None)."""
    L = _lib()
    dp = C.POINTER(C.c_double)
    nl = _element(etype).nloc
    xa = np.ascontiguousarray(x, dtype=np.float64)
    ca = np.ascontiguousarray(cells, dtype=np.int64)
    ua, una, pa = (np.ascontiguousarray(a, dtype=np.float64) for a in (u, un, p))
    u2 = None if un2 is None else np.ascontiguousarray(un2, dtype=np.float64)
    ff = None if facet_flags is None else np.ascontiguousarray(facet_flags, dtype=np.uint16)
    nc = len(ca)
    Fe = np.empty((nc, 4 * nl))
    Je = np.empty((nc, 4 * nl, 4 * nl)) if want_jac else None
    P = _Params(prm.dt, prm.rho, prm.mu, prm.mu_facet, (C.c_double * 3)(*[float(v) for v in prm.f][:3]), prm.theta, prm.a0, prm.a1, prm.a2,
                prm.beta_backflow, int(prm.ds_terms), 0)
    L.orcg3_element_tensors(etype, nc, ca.ctypes.data_as(C.POINTER(C.c_int64)), xa.ctypes.data_as(dp), ua.ctypes.data_as(dp),
                            una.ctypes.data_as(dp), u2.ctypes.data_as(dp) if u2 is not None else None, pa.ctypes.data_as(dp), C.byref(P),
                            ff.ctypes.data_as(C.POINTER(C.c_uint16)) if ff is not None else None, 1 if want_jac else 0,
                            Fe.ctypes.data_as(dp), Je.ctypes.data_as(dp) if Je is not None else None)
    return Fe, Je
