"""ctypes binding of libcfdh.so (include/cfdh.h).  No CPU fallback: a missing
library or a missing GPU is an error, never a silent detour."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcfdh.so")
_LIB = None

dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int32)
lp = C.POINTER(C.c_int64)


class Options(C.Structure):
    _fields_ = [
        ("snes_rtol", C.c_double), ("snes_atol", C.c_double), ("snes_stol", C.c_double), ("snes_max_it", C.c_int32),
        ("ksp_rtol", C.c_double), ("ksp_atol", C.c_double), ("ksp_max_it", C.c_int32), ("ksp_restart", C.c_int32),
        ("cheb_degree", C.c_int32), ("cheb_ratio", C.c_double), ("schur_full", C.c_int32),
        ("amg_smooth_degree", C.c_int32), ("amg_smooth_ratio", C.c_double), ("amg_theta", C.c_double),
        ("amg_max_coarse", C.c_int32), ("pc_refresh", C.c_int32), ("remove_p_mean", C.c_int32), ("verbose", C.c_int32),
        ("pc_type", C.c_int32), ("cc_smooth_degree", C.c_int32), ("ksp_guess", C.c_int32),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("newton_its", C.c_int32), ("krylov_its", C.c_int32), ("reason", C.c_int32), ("pc_refreshes", C.c_int32),
        ("fnorm0", C.c_double), ("fnorm", C.c_double), ("ms_assemble", C.c_double), ("ms_solve", C.c_double),
        ("ms_pc_setup", C.c_double), ("ms_total", C.c_double),
    ]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, dp, C.c_int, C.c_int)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, dp, dp)

# every symbol include/cfdh.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "cfdh_create", "cfdh_create_elem", "cfdh_create_elem_part", "cfdh_set_facet_markers", "cfdh_destroy", "cfdh_last_error", "cfdh_abi_version", "cfdh_set_params", "cfdh_default_options",
    "cfdh_set_options", "cfdh_clear_dirichlet", "cfdh_add_dirichlet", "cfdh_update_dirichlet", "cfdh_set_state", "cfdh_get_solution",
    "cfdh_get_previous", "cfdh_get_residual", "cfdh_advance", "cfdh_advance_field", "cfdh_set_time_scheme", "cfdh_set_previous2", "cfdh_get_previous2",
    "cfdh_shift_history", "cfdh_set_boundary_terms", "cfdh_assemble", "cfdh_get_csr", "cfdh_spmv", "cfdh_solve_step",
    "cfdh_functional", "cfdh_wall_shear_stress", "cfdh_set_global_pressure_space", "cfdh_set_halo", "cfdh_comm_unique_id", "cfdh_comm_init", "cfdh_comm_set_callbacks",
    "cfdh_profile_enable", "cfdh_profile_get", "cfdh_profile_reset", "cfdh_info",
]


def build(force=False):
    """Compile libcfdh.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    src = os.path.join(_HERE, "csrc")
    newest = max(os.path.getmtime(os.path.join(src, f)) for f in os.listdir(src)
                 if f.endswith((".hip", ".cpp", ".hpp", ".h")))
    hdr = os.path.join(os.path.dirname(_HERE), "include", "cfdh.h")
    if os.path.exists(hdr):
        newest = max(newest, os.path.getmtime(hdr))
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < newest:
        if not os.path.exists("/opt/rocm/bin/hipcc") and os.path.exists(_SO):
            return _SO
        subprocess.check_call(["make", "-C", src, "-s", "-j4"])
    return _SO


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(_SO):
        raise RuntimeError(
            "libcfdh.so is missing (%s): build it with `python -c 'import __graft_entry__ as g; g.build()'`; "
            "this package has no CPU fallback" % _SO)
    L = C.CDLL(_SO)
    vp = C.c_void_p
    L.cfdh_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, ip, dp, C.c_int64, ip, ip, ip]
    L.cfdh_create_elem.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, ip, dp, C.c_int64, ip, ip, ip]
    L.cfdh_create_elem_part.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, ip, dp, C.c_int64, ip, ip, ip]
    L.cfdh_set_facet_markers.argtypes = [vp, C.c_int64, ip]
    L.cfdh_destroy.argtypes = [vp]
    L.cfdh_destroy.restype = None
    L.cfdh_last_error.argtypes = [vp]
    L.cfdh_last_error.restype = C.c_char_p
    L.cfdh_set_params.argtypes = [vp, C.c_double, C.c_double, C.c_double, C.c_double, dp]
    L.cfdh_default_options.argtypes = [C.POINTER(Options)]
    L.cfdh_set_options.argtypes = [vp, C.POINTER(Options)]
    L.cfdh_clear_dirichlet.argtypes = [vp]
    L.cfdh_add_dirichlet.argtypes = [vp, C.c_int, C.c_int64, ip, dp]
    L.cfdh_update_dirichlet.argtypes = [vp, C.c_int, C.c_int64, ip, dp]
    L.cfdh_set_state.argtypes = [vp, dp, dp, dp, dp]
    L.cfdh_get_solution.argtypes = [vp, dp, dp]
    L.cfdh_set_time_scheme.argtypes = [vp, C.c_double, C.c_double, C.c_double, C.c_double]
    L.cfdh_set_previous2.argtypes = [vp, dp]
    L.cfdh_get_previous2.argtypes = [vp, dp]
    L.cfdh_shift_history.argtypes = [vp]
    L.cfdh_wall_shear_stress.argtypes = [vp, dp]
    L.cfdh_set_boundary_terms.argtypes = [vp, C.c_int, C.c_int, C.c_double]
    L.cfdh_get_residual.argtypes = [vp, dp, dp]
    L.cfdh_get_previous.argtypes = [vp, dp, dp]
    L.cfdh_advance.argtypes = [vp]
    L.cfdh_advance_field.argtypes = [vp, C.c_int]
    L.cfdh_assemble.argtypes = [vp, C.c_int]
    L.cfdh_get_csr.argtypes = [vp, lp, ip, ip, dp]
    L.cfdh_spmv.argtypes = [vp, dp, dp]
    L.cfdh_solve_step.argtypes = [vp, C.POINTER(Stats)]
    L.cfdh_functional.argtypes = [vp, C.c_int, C.c_int, dp]
    L.cfdh_set_halo.argtypes = [vp, C.c_int, ip, lp, ip, lp, ip]
    L.cfdh_set_global_pressure_space.argtypes = [vp, C.c_int64, C.c_int64, ip, dp, ip, C.c_int64, ip]
    L.cfdh_comm_unique_id.argtypes = [C.c_void_p]
    L.cfdh_comm_init.argtypes = [vp, C.c_void_p, C.c_int, C.c_int]
    L.cfdh_comm_set_callbacks.argtypes = [vp, ALLREDUCE_FN, EXCHANGE_FN, C.c_void_p, C.c_int, C.c_int]
    L.cfdh_profile_enable.argtypes = [vp, C.c_int]
    L.cfdh_profile_get.argtypes = [vp, C.c_int, dp, lp]
    L.cfdh_profile_reset.argtypes = [vp]
    L.cfdh_info.argtypes = [vp, C.c_int]
    L.cfdh_info.restype = C.c_int64
    _LIB = L
    return L


def _dp(a):
    return None if a is None else a.ctypes.data_as(dp)


def _ip(a):
    return None if a is None else a.ctypes.data_as(ip)


def _lp(a):
    return None if a is None else a.ctypes.data_as(lp)


class CfdhError(RuntimeError):
    pass


def _raise(code, msg):
    # error mapping of the reference: ValueError bad config, RuntimeError init/convergence
    # (/root/reference/main.py:56-82, stabilized_schur.py:332-334)
    if code == -1:
        raise ValueError(msg)
    raise CfdhError(msg)


class Context:
    """Thin owner of a cfdh_ctx: arrays in, arrays out, exceptions for error codes."""

    def __init__(self, x, cells, facet_cells, facet_local, facet_marker, nv_owned=None, device=0, etype=0):
        """etype 0: P1 triangles / tetrahedra (cfdh_create); 1 P2 triangles, 2 Q1 quadrilaterals, 3 P1 triangles through the
        generic kernels (cfdh_create_elem: `x` are node coordinates, `cells` list nloc nodes)."""
        L = lib()
        self.L = L
        x = np.asarray(x, dtype=np.float64)
        cells = np.asarray(cells)
        self.etype = int(etype)
        # generic elements: the geometric dimension follows from the coordinates (P2: 6 nodes -> triangles, 10 -> tetrahedra;
        # Q1: 4 -> quadrilaterals, 8 -> hexahedra; P1 through the generic kernels: 3 / 4)
        self.dim = (3 if x.shape[1] >= 3 and cells.shape[1] in (8, 10) or (self.etype == 3 and cells.shape[1] == 4) else 2) if self.etype \
            else cells.shape[1] - 1  # triangles -> 2, tetrahedra -> 3
        if self.dim not in (2, 3) or x.shape[1] < self.dim:
            raise ValueError("cells must be triangles [nc,3] or tetrahedra [nc,4] with matching coordinates")
        nodes = {(1, 2): 6, (2, 2): 4, (3, 2): 3, (1, 3): 10, (2, 3): 8, (3, 3): 4}
        if self.etype and cells.shape[1] != nodes[(self.etype, self.dim)]:
            raise ValueError("element type %d needs %d nodes per cell for gdim %d" % (self.etype, nodes[(self.etype, self.dim)], self.dim))
        self.x = np.ascontiguousarray(x[:, : self.dim]).copy()
        self.cells = np.ascontiguousarray(cells, dtype=np.int32)
        self.nv = len(self.x)
        self.nvo = self.nv if nv_owned is None else int(nv_owned)
        fc = np.ascontiguousarray(facet_cells, dtype=np.int32)
        fl = np.ascontiguousarray(facet_local, dtype=np.int32)
        fm = np.ascontiguousarray(facet_marker, dtype=np.int32)
        h = C.c_void_p()
        if self.etype:
            if self.nvo != self.nv:   # one part of a partitioned run (owned nodes first, ghosts after)
                rc = L.cfdh_create_elem_part(C.byref(h), int(device), self.dim, self.etype, self.nv, self.nvo, len(self.cells), _ip(self.cells),
                                             _dp(self.x), len(fc), _ip(fc), _ip(fl), _ip(fm))
            else:
                rc = L.cfdh_create_elem(C.byref(h), int(device), self.dim, self.etype, self.nv, len(self.cells), _ip(self.cells), _dp(self.x),
                                        len(fc), _ip(fc), _ip(fl), _ip(fm))
        else:
            rc = L.cfdh_create(C.byref(h), int(device), self.dim, self.nv, self.nvo, len(self.cells), _ip(self.cells), _dp(self.x),
                               len(fc), _ip(fc), _ip(fl), _ip(fm))
        if rc != 0:
            _raise(rc, "cfdh_create failed: " + L.cfdh_last_error(None).decode())
        self.h = h
        self._cb = None

    def close(self):
        if getattr(self, "h", None):
            self.L.cfdh_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            _raise(rc, self.L.cfdh_last_error(self.h).decode())

    def set_params(self, dt, rho, mu, mu_facet=None, f=(0.0, 0.0)):
        ff = np.zeros(3)
        f = np.atleast_1d(np.asarray(f, dtype=np.float64))[:3]
        ff[: len(f)] = f
        self._chk(self.L.cfdh_set_params(self.h, dt, rho, mu, mu if mu_facet is None else mu_facet, _dp(ff)))

    def default_options(self):
        o = Options()
        self.L.cfdh_default_options(C.byref(o))
        return o

    def set_options(self, o):
        self._chk(self.L.cfdh_set_options(self.h, C.byref(o)))

    def clear_dirichlet(self):
        self._chk(self.L.cfdh_clear_dirichlet(self.h))

    def add_dirichlet(self, field, nodes, values):
        nodes = np.ascontiguousarray(nodes, dtype=np.int32)
        values = np.ascontiguousarray(values, dtype=np.float64)
        self._chk(self.L.cfdh_add_dirichlet(self.h, int(field), len(nodes), _ip(nodes), _dp(values)))

    def update_dirichlet(self, field, nodes, values):
        """New values for dofs that are already constrained (no object added, diagonal counts unchanged)."""
        nodes = np.ascontiguousarray(nodes, dtype=np.int32)
        values = np.ascontiguousarray(values, dtype=np.float64)
        self._chk(self.L.cfdh_update_dirichlet(self.h, int(field), len(nodes), _ip(nodes), _dp(values)))

    def set_state(self, u_prev=None, p_prev=None, u=None, p=None):
        a = [None if v is None else np.ascontiguousarray(v, dtype=np.float64).reshape(-1) for v in (u_prev, p_prev, u, p)]
        for v, n in zip(a, (self.dim * self.nv, self.nv, self.dim * self.nv, self.nv)):
            if v is not None and v.size != n:
                raise ValueError("state array has wrong size")
        self._chk(self.L.cfdh_set_state(self.h, _dp(a[0]), _dp(a[1]), _dp(a[2]), _dp(a[3])))

    def get_solution(self, u=None, p=None):
        u = np.empty(self.dim * self.nv) if u is None else u
        p = np.empty(self.nv) if p is None else p
        self._chk(self.L.cfdh_get_solution(self.h, _dp(u), _dp(p)))
        return u, p

    def get_previous(self, u=None, p=None):
        u = np.empty(self.dim * self.nv) if u is None else u
        p = np.empty(self.nv) if p is None else p
        self._chk(self.L.cfdh_get_previous(self.h, _dp(u), _dp(p)))
        return u, p

    def get_residual(self):
        ru, rp = np.zeros(self.dim * self.nv), np.zeros(self.nv)
        self._chk(self.L.cfdh_get_residual(self.h, _dp(ru), _dp(rp)))
        return ru, rp

    def advance(self):
        self._chk(self.L.cfdh_advance(self.h))

    def advance_field(self, field):
        self._chk(self.L.cfdh_advance_field(self.h, int(field)))

    def set_time_scheme(self, theta, a0, a1, a2):
        self._chk(self.L.cfdh_set_time_scheme(self.h, float(theta), float(a0), float(a1), float(a2)))

    def set_facet_markers(self, markers):
        """Markers of the exterior facets in the order given to the constructor (Solver.setup's facet_tags)."""
        m = np.ascontiguousarray(markers, dtype=np.int32)
        self._chk(self.L.cfdh_set_facet_markers(self.h, len(m), _ip(m)))

    def set_boundary_terms(self, ds_terms=True, backflow_marker=-1, beta=0.0):
        self._chk(self.L.cfdh_set_boundary_terms(self.h, int(bool(ds_terms)), int(backflow_marker), float(beta)))

    def wall_shear_stress(self, download=True):
        if not download:
            self._chk(self.L.cfdh_wall_shear_stress(self.h, None))
            return None
        out = np.zeros(self.dim * self.nv)
        self._chk(self.L.cfdh_wall_shear_stress(self.h, _dp(out)))
        return out

    def set_previous2(self, u_prev2):
        u_prev2 = np.ascontiguousarray(u_prev2, dtype=np.float64).reshape(-1)
        assert u_prev2.size == self.dim * self.nv
        self._chk(self.L.cfdh_set_previous2(self.h, _dp(u_prev2)))

    def get_previous2(self):
        u = np.empty(self.dim * self.nv)
        self._chk(self.L.cfdh_get_previous2(self.h, _dp(u)))
        return u

    def shift_history(self):
        self._chk(self.L.cfdh_shift_history(self.h))

    def assemble(self, want_jacobian=True):
        self._chk(self.L.cfdh_assemble(self.h, int(want_jacobian)))

    def get_csr(self):
        import scipy.sparse as sp
        nnz = C.c_int64()
        self._chk(self.L.cfdh_get_csr(self.h, C.byref(nnz), None, None, None))
        n1 = self.dim + 1
        rowptr = np.empty(n1 * self.nvo + 1, dtype=np.int32)
        col = np.empty(nnz.value, dtype=np.int32)
        val = np.empty(nnz.value)
        self._chk(self.L.cfdh_get_csr(self.h, C.byref(nnz), _ip(rowptr), _ip(col), _dp(val)))
        return sp.csr_matrix((val, col, rowptr), shape=(n1 * self.nvo, n1 * self.nv))

    def spmv(self, xvec):
        xvec = np.ascontiguousarray(xvec, dtype=np.float64)
        y = np.empty((self.dim + 1) * self.nvo)
        self._chk(self.L.cfdh_spmv(self.h, _dp(xvec), _dp(y)))
        return y

    def solve_step(self):
        st = Stats()
        rc = self.L.cfdh_solve_step(self.h, C.byref(st))
        if rc != 0:
            msg = self.L.cfdh_last_error(self.h).decode()
            if rc == -4:
                # the reference raises RuntimeError(f"Did not converge, reason: {reason}.") (stabilized_schur.py:332-334)
                raise RuntimeError("Did not converge, reason: %d. (%s)" % (st.reason, msg))
            _raise(rc, msg)
        return st

    def functional(self, kind, marker=0):
        out = C.c_double()
        self._chk(self.L.cfdh_functional(self.h, int(kind), int(marker), C.byref(out)))
        return out.value

    def set_halo(self, nbr_rank, send_ptr, send_idx, recv_ptr, recv_idx):
        a = np.ascontiguousarray(nbr_rank, dtype=np.int32)
        sp_ = np.ascontiguousarray(send_ptr, dtype=np.int64)
        si = np.ascontiguousarray(send_idx, dtype=np.int32)
        rp = np.ascontiguousarray(recv_ptr, dtype=np.int64)
        ri = np.ascontiguousarray(recv_idx, dtype=np.int32)
        self._chk(self.L.cfdh_set_halo(self.h, len(a), _ip(a), _lp(sp_), _ip(si), _lp(rp), _ip(ri)))

    def set_global_pressure_space(self, x_global, cells_global, owned_global, pbc_nodes_global):
        xg = np.ascontiguousarray(x_global, dtype=np.float64)[:, : self.dim].copy()
        cg = np.ascontiguousarray(cells_global, dtype=np.int32)
        og = np.ascontiguousarray(owned_global, dtype=np.int32)
        pb = np.ascontiguousarray(pbc_nodes_global, dtype=np.int32)
        self._chk(self.L.cfdh_set_global_pressure_space(self.h, len(xg), len(cg), _ip(cg), _dp(xg), _ip(og), len(pb), _ip(pb)))

    def comm_init_rccl(self, uid_bytes, rank, nranks):
        buf = C.create_string_buffer(bytes(uid_bytes), 128)
        self._chk(self.L.cfdh_comm_init(self.h, buf, int(rank), int(nranks)))

    def comm_set_callbacks(self, allreduce, exchange, rank, nranks):
        self._cb = (ALLREDUCE_FN(allreduce), EXCHANGE_FN(exchange))
        self._chk(self.L.cfdh_comm_set_callbacks(self.h, self._cb[0], self._cb[1], None, int(rank), int(nranks)))

    def profile_enable(self, on=True):
        self._chk(self.L.cfdh_profile_enable(self.h, int(on)))

    def profile_reset(self):
        self._chk(self.L.cfdh_profile_reset(self.h))

    def profile_get(self, kind):
        ms, n = C.c_double(), C.c_int64()
        self._chk(self.L.cfdh_profile_get(self.h, int(kind), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def info(self, what):
        return int(self.L.cfdh_info(self.h, int(what)))


def rccl_unique_id():
    buf = C.create_string_buffer(128)
    rc = lib().cfdh_comm_unique_id(buf)
    if rc != 0:
        raise CfdhError("cfdh_comm_unique_id failed: " + lib().cfdh_last_error(None).decode())
    return bytes(buf.raw)
