"""ctypes binding of the C oracle  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
(see oracle/cfdh_oracle.c header: parity unpinned; only tests/, smoke() and
bench.py's cpu_baseline leg may import this.)"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libcfdh_oracle.so")
    srcs = [os.path.join(_HERE, n) for n in ("cfdh_oracle.c", "cfdh_oracle3.c", "cfdh_oracle_gen.c")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(q) for q in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return so


class Opts(C.Structure):
    _fields_ = [("snes_rtol", C.c_double), ("snes_atol", C.c_double), ("snes_stol", C.c_double),
                ("snes_max_it", C.c_int), ("ksp_rtol", C.c_double), ("ksp_atol", C.c_double),
                ("ksp_max_it", C.c_int), ("ksp_restart", C.c_int), ("sub_rtol", C.c_double),
                ("sub_max_it", C.c_int), ("sub_restart", C.c_int), ("remove_p_mean", C.c_int),
                ("verbose", C.c_int), ("pc_kind", C.c_int), ("cheb_degree", C.c_int), ("cheb_ratio", C.c_double),
                ("amg_smooth_degree", C.c_int), ("amg_smooth_ratio", C.c_double), ("amg_theta", C.c_double),
                ("amg_max_coarse", C.c_int), ("cc_smooth_degree", C.c_int), ("schur_upper", C.c_int), ("ksp_guess", C.c_int)]


class Stats(C.Structure):
    _fields_ = [("newton_its", C.c_int), ("krylov_its", C.c_int), ("reason", C.c_int), ("sub_its", C.c_int),
                ("fnorm0", C.c_double), ("fnorm", C.c_double), ("ms_assemble", C.c_double),
                ("ms_solve", C.c_double)]


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_int, C.c_int, ip, dp, C.c_int, ip, ip]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_set_params.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double, dp]
        L.orc_set_threads.argtypes = [C.c_void_p, C.c_int]
        L.orc_get_threads.argtypes = [C.c_void_p]
        L.orc_get_threads.restype = C.c_int
        L.orc_clear_bcs.argtypes = [C.c_void_p]
        L.orc_add_bc.argtypes = [C.c_void_p, C.c_int, C.c_int, ip, dp]
        L.orc_set_un.argtypes = [C.c_void_p, dp]
        L.orc_set_un2.argtypes = [C.c_void_p, dp]
        L.orc_set_boundary_terms.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int, ip]
        L.orc_set_scheme.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double]
        L.orc_assemble.argtypes = [C.c_void_p, dp, C.c_int, dp]
        L.orc_get_csr.argtypes = [C.c_void_p, ip, C.POINTER(ip), C.POINTER(ip), C.POINTER(dp)]
        L.orc_default_opts.argtypes = [C.POINTER(Opts)]
        L.orc_solve_step.argtypes = [C.c_void_p, dp, C.POINTER(Opts), C.POINTER(Stats)]
        L.orc_solve_step.restype = C.c_int
        L.orc_last_error.argtypes = [C.c_void_p]
        L.orc_last_error.restype = C.c_char_p
        L.orc_spmv.argtypes = [C.c_void_p, dp, dp]
        L.orc_functional.argtypes = [C.c_void_p, dp, C.c_int, C.c_int, ip, C.c_double]
        L.orc_functional.restype = C.c_double
        L.orc_wss.argtypes = [C.c_void_p, dp, C.c_double, dp]
        L.orc_element.argtypes = [C.c_double, C.c_double, C.c_double, C.c_double, dp, dp, dp, dp, dp, C.c_int, dp, dp]
        _LIB = L
    return _LIB


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def default_opts(**kw):
    o = Opts()
    lib().orc_default_opts(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


class Oracle:
    """Plain-array problem: x [nv,2], cells [nc,3] (or x [nv,3], cells [nc,4]: tetrahedra), exterior facets (cell, local)."""

    def __init__(self, x, cells, facet_cells, facet_local, dt, rho, mu, f=(0.0, 0.0), mu_facet=None, etg=0):
        """etg != 0: a nodal element beyond P1 (the element codes of np_twin_gen.py / np_twin_gen3.py: 1 P2 triangles, 2 Q1
        parallelograms, 4 P2 tetrahedra, 5 Q1 hexahedra) -- x are node coordinates, cells list the nodes in the DOLFINx local order;
        Newton / FGMRES / Cahouet-Chabard + AMG of cfdh_oracle.c over the element routines of cfdh_oracle_gen*.c (pc_kind 2 only)."""
        L = lib()
        self.x = np.ascontiguousarray(x, dtype=np.float64)
        self.cells = np.ascontiguousarray(cells, dtype=np.int32)
        self.fc = np.ascontiguousarray(facet_cells, dtype=np.int32)
        self.fl = np.ascontiguousarray(facet_local, dtype=np.int32)
        self.nv, self.nc = len(self.x), len(self.cells)
        self.etg = int(etg)
        # triangles: 2; tetrahedra: 3 (element tensors from cfdh_oracle3.c, pc_kind 2 only); generic elements: from the coordinates
        self.dim = self.x.shape[1] if self.etg else self.cells.shape[1] - 1
        assert self.x.shape[1] == self.dim
        self.ndof = (self.dim + 1) * self.nv
        sig = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_create_d.restype = C.c_void_p
        L.orc_create_d.argtypes = sig
        L.orc_create_gen.restype = C.c_void_p
        L.orc_create_gen.argtypes = [C.c_int] + sig
        if self.etg:
            self.h = L.orc_create_gen(self.dim, self.etg, self.nv, self.nc, _ip(self.cells), _dp(self.x), len(self.fc), _ip(self.fc), _ip(self.fl))
            if not self.h:
                raise ValueError("orc_create_gen: element code %d does not exist for gdim %d" % (self.etg, self.dim))
        else:
            self.h = L.orc_create_d(self.dim, self.nv, self.nc, _ip(self.cells), _dp(self.x), len(self.fc), _ip(self.fc), _ip(self.fl))
        ff = np.zeros(3)
        ff[: len(np.atleast_1d(f))] = np.asarray(f, dtype=np.float64)
        L.orc_set_params(self.h, dt, rho, mu, mu if mu_facet is None else mu_facet, _dp(ff))
        self.mu = mu

    def __del__(self):
        try:
            lib().orc_destroy(self.h)
        except Exception:
            pass

    def set_threads(self, n):
        lib().orc_set_threads(self.h, int(n))

    def threads(self):
        return lib().orc_get_threads(self.h)

    def clear_bcs(self):
        lib().orc_clear_bcs(self.h)

    def add_bc_u(self, nodes, values):
        nodes = np.ascontiguousarray(nodes, dtype=np.int32)
        values = np.ascontiguousarray(values, dtype=np.float64).reshape(-1, self.dim)
        lib().orc_add_bc(self.h, 0, len(nodes), _ip(nodes), _dp(values))

    def add_bc_p(self, nodes, values):
        nodes = np.ascontiguousarray(nodes, dtype=np.int32)
        values = np.ascontiguousarray(values, dtype=np.float64).reshape(-1)
        lib().orc_add_bc(self.h, 1, len(nodes), _ip(nodes), _dp(values))

    def set_un(self, un):
        un = np.ascontiguousarray(un, dtype=np.float64).reshape(-1)
        lib().orc_set_un(self.h, _dp(un))

    def set_un2(self, un2):
        un2 = np.ascontiguousarray(un2, dtype=np.float64).reshape(-1)
        lib().orc_set_un2(self.h, _dp(un2))

    def set_boundary_terms(self, ds_terms, backflow_facets=None, beta=0.0):
        """ds_terms False + backflow facets = stabilized_schur_backflow.py:107,158-176."""
        bf = np.ascontiguousarray(backflow_facets if backflow_facets is not None else [], dtype=np.int32)
        lib().orc_set_boundary_terms(self.h, int(bool(ds_terms)), float(beta), len(bf), _ip(bf))

    def set_scheme(self, theta, a0, a1, a2):
        """(1/2; 1,-1,0) = stabilized_schur.py; (1; 1,-1,0) / (1; 1.5,-2,.5) = stabilized_schur_bdf2.py:95-110,298-305."""
        lib().orc_set_scheme(self.h, float(theta), float(a0), float(a1), float(a2))

    def assemble(self, xv, want_jac=True):
        xv = np.ascontiguousarray(xv, dtype=np.float64)
        F = np.empty(self.ndof)
        lib().orc_assemble(self.h, _dp(xv), int(want_jac), _dp(F))
        return F

    def csr(self):
        import scipy.sparse as sp
        nnz = C.c_int()
        rp, cl, vl = C.POINTER(C.c_int)(), C.POINTER(C.c_int)(), C.POINTER(C.c_double)()
        lib().orc_get_csr(self.h, C.byref(nnz), C.byref(rp), C.byref(cl), C.byref(vl))
        rowptr = np.ctypeslib.as_array(rp, (self.ndof + 1,)).copy()
        col = np.ctypeslib.as_array(cl, (nnz.value,)).copy()
        val = np.ctypeslib.as_array(vl, (nnz.value,)).copy()
        return sp.csr_matrix((val, col, rowptr), shape=(self.ndof, self.ndof))

    def solve_step(self, xv, opts=None):
        opts = opts or default_opts()
        st = Stats()
        xv = np.ascontiguousarray(xv, dtype=np.float64)
        r = lib().orc_solve_step(self.h, _dp(xv), C.byref(opts), C.byref(st))
        if r < 0:
            raise RuntimeError("Did not converge, reason: %d. %s" % (r, lib().orc_last_error(self.h).decode()))
        return xv, st

    def functional(self, xv, kind, facets=None):
        xv = np.ascontiguousarray(xv, dtype=np.float64)
        fa = np.ascontiguousarray(facets if facets is not None else [], dtype=np.int32)
        return lib().orc_functional(self.h, _dp(xv), int(kind), len(fa), _ip(fa), float(self.mu))


def _wss(self, xv):
    """Wall shear stress field [2*nv] of solverBase.py:163-195 for the monolithic state xv."""
    xv = np.ascontiguousarray(xv, dtype=np.float64)
    out = np.empty(2 * self.nv)
    lib().orc_wss(self.h, _dp(xv), float(self.mu), _dp(out))
    return out


Oracle.wall_shear_stress = _wss


def element(dt, rho, mu, muf, f, xe, ue, une, pe, fflag):
    Fe = np.empty(9)
    Je = np.empty((9, 9))
    a = [np.ascontiguousarray(v, dtype=np.float64) for v in (f, xe, ue, une, pe)]
    lib().orc_element(dt, rho, mu, muf, _dp(a[0]), _dp(a[1]), _dp(a[2]), _dp(a[3]), _dp(a[4]), int(fflag), _dp(Fe), _dp(Je))
    return Fe, Je
