"""Drop-in `Solver` for the reference's `--solver stabilized_schur`
(/root/reference/src/solvers/stabilized_schur.py:40-334): SUPG/PSPG/LSIC
stabilised P1/P1 Navier-Stokes, Newton linearisation, Schur-complement
preconditioned FGMRES -- executed by hand-written gfx950 kernels in
libcfdh.so (include/cfdh.h) instead of FEniCSx/PETSc.

Same constructor, `setup(bcu, bcp, facet_tags=None, tags=None)`, `solveStep()`,
state Functions and error behaviour (RuntimeError "Did not converge, reason: r.").
Fields live in HBM between steps; the host arrays behind `*.x.array` are
synchronised lazily (download on read, upload when the caller may have
written).  Extra, optional keyword arguments tune the solver:
`newton_rtol`, `krylov_rtol`, `device`, `verbose`, `quiet`, `options` (dict of
cfdh_options fields) and `comm` (a parallel.PartComm: the mesh is then
partitioned by elements over the ranks, one GPU each, as DOLFINx partitions it
over MPI ranks in the reference).
"""
from __future__ import annotations

from typing import Callable

import numpy as np

from .. import _lib
from ..boundaryCondition import BoundaryCondition
from ..solverBase import SolverBase


class Solver(SolverBase):
    MAX_ITER = 20

    def __init__(self, mesh, dt: float, rho: float, mu: float, f: list,
                 initial_velocity: Callable[[np.ndarray], np.ndarray] = None, **kwargs):
        super().__init__(mesh, dt, rho, mu, f)
        # Lagrange degree of both spaces: 1 as in stabilized_schur.py:55-57 (on quadrilateral cells that is Q1); the backflow
        # variant passes its p_grade (stabilized_schur_backflow.py:84-87)
        degree = int(kwargs.pop("_degree", 1))
        super().initVelocitySpace("Lagrange", mesh.topology.cell_name(), degree, shape=(mesh.geometry.dim,))
        super().initPressureSpace("Lagrange", mesh.topology.cell_name(), degree)
        dm = self._dm
        etype = int(getattr(dm, "etype", 0))
        if kwargs.get("generic_kernels") and etype == 0 and mesh.geometry.dim == 2:
            etype = 3  # P1 triangles through the quadrature kernels of the P2 / Q1 path (cross-check)
        if initial_velocity:
            self.u_prev.interpolate(initial_velocity)
        self._mu_float = float(mu)  # raw python float of the ds term (stabilized_schur.py:79)
        self._verbose = int(kwargs.get("verbose", 0))
        self._quiet = bool(kwargs.get("quiet", False))
        self._comm = kwargs.get("comm", None)
        device = int(kwargs.get("device", 0))
        self._part = None
        if etype != 0 and self._comm is not None and self._comm.size > 1:
            # P2 / Q1 in a partitioned run (round 4, gdim 2 and 3): the NODE mesh is partitioned exactly like a vertex mesh -- owned nodes,
            # every cell touching one, the remaining nodes of those cells as ghosts (cfdh_create_elem_part)
            from ..mesh import PartCommView
            mesh.comm = PartCommView(self._comm)
            part = self._comm.make_part(dm)
            self._part = part
            self.ctx = _lib.Context(part.x, part.cells, part.facet_cells, part.facet_local, part.facet_marker, nv_owned=part.nvo, device=device,
                                    etype=etype)
            self._comm.attach(self.ctx)
        elif etype != 0:
            self.ctx = _lib.Context(dm.x, dm.cells, dm.facet_cells, dm.facet_local, dm.facet_marker, device=device, etype=etype)
        elif self._comm is not None and self._comm.size > 1:
            from ..mesh import PartCommView
            mesh.comm = PartCommView(self._comm)  # rank-0 guards of the harness (printing, file output) see the real rank
            part = self._comm.make_part(mesh)
            self._part = part
            self.ctx = _lib.Context(part.x, part.cells, part.facet_cells, part.facet_local, part.facet_marker,
                                    nv_owned=part.nvo, device=device)
            self._comm.attach(self.ctx)
        else:
            self.ctx = _lib.Context(mesh.x, mesh.cells, mesh.facet_cells, mesh.facet_local, mesh.facet_marker,
                                    device=device)
        ff = np.atleast_1d(np.asarray(f, dtype=np.float64))
        self.ctx.set_params(float(dt), float(rho), float(mu), mu_facet=self._mu_float, f=ff)
        self.options = self.ctx.default_options()
        if "newton_rtol" in kwargs:
            self.options.snes_rtol = float(kwargs["newton_rtol"])
        if "krylov_rtol" in kwargs:
            self.options.ksp_rtol = float(kwargs["krylov_rtol"])
        for k, v in dict(kwargs.get("options", {})).items():
            setattr(self.options, k, v)
        self.options.verbose = self._verbose
        self.ctx.set_options(self.options)
        self.last_stats = None
        self._bcs = []
        self._bc_cache = None
        self._bc_nodes = None
        self._bc_owned = None
        # lazy host/device synchronisation of the state Functions
        self._dev_newer = {"sol": False, "res": False, "wss": False}
        self._prev_host_dirty = True
        self._prev_dev_newer = False
        self.transfers = {"downloads": 0, "uploads": 0}  # whole-field host <-> device copies since construction
        self._u_sol.x._pre_access = self._sync_solution
        self._p_sol.x._pre_access = self._sync_solution
        self.u_residual.x._pre_access = self._sync_residual
        self.p_residual.x._pre_access = self._sync_residual
        self._u_prev.x._pre_access = self._sync_previous
        self._p_prev.x._pre_access = self._sync_previous
        self._u_prev.x._post_access = self._mark_prev_dirty
        self._p_prev.x._post_access = self._mark_prev_dirty
        # `u_prev.x.array[:] = u_sol.x.array[:]` (scenario.py:306-307) stays on the device
        self._u_prev.x._assign_hook = lambda src: self._assign_previous(0, src)
        self._p_prev.x._assign_hook = lambda src: self._assign_previous(1, src)

    # -- global <-> local (identity on one GPU) ------------------------------------
    def _loc_u(self, a):
        return a if self._part is None else np.ascontiguousarray(a.reshape(-1, self.mesh.geometry.dim)[self._part.l2g]).reshape(-1)

    def _loc_p(self, a):
        return a if self._part is None else np.ascontiguousarray(a[self._part.l2g])

    def _store(self, dst_u, dst_p, lu, lp):
        if self._part is None:
            dst_u[:] = lu
            dst_p[:] = lp
        else:
            nvg = self._dm.num_vertices  # nodes of the function space (= mesh vertices for P1 / Q1)
            dst_u[:] = self._comm.allgather_owned(lu, self.mesh.geometry.dim, nvg)
            dst_p[:] = self._comm.allgather_owned(lp, 1, nvg)

    # -- lazy sync -----------------------------------------------------------
    def _sync_solution(self):
        if self._dev_newer["sol"]:
            self._dev_newer["sol"] = False
            self.transfers["downloads"] += 1
            lu, lp = self.ctx.get_solution()
            self._store(self._u_sol.x._array, self._p_sol.x._array, lu, lp)

    def _sync_residual(self):
        if self._dev_newer["res"]:
            self._dev_newer["res"] = False
            ru, rp = self.ctx.get_residual()
            self._store(self.u_residual.x._array, self.p_residual.x._array, ru, rp)

    def _sync_wss(self):
        if self._dev_newer["wss"]:
            self._dev_newer["wss"] = False
            w = self.ctx.wall_shear_stress(download=True)
            a = self.shear_stress.x._array
            a[:] = w if self._part is None else self._comm.allgather_owned(w, self.mesh.geometry.dim, self._dm.num_vertices)

    def assemble_wss(self):
        """solverBase.py:185-195 on the device (cfdh_wall_shear_stress); the host array behind
        `shear_stress.x.array` is filled when it is read."""
        if not hasattr(self, "shear_stress"):
            return
        self.shear_stress.x._pre_access = self._sync_wss
        self.ctx.wall_shear_stress(download=False)
        self._dev_newer["wss"] = True

    def _sync_previous(self):
        if self._prev_dev_newer:
            self._prev_dev_newer = False
            self.transfers["downloads"] += 1
            lu, lp = self.ctx.get_previous()
            self._store(self._u_prev.x._array, self._p_prev.x._array, lu, lp)

    def _mark_prev_dirty(self):
        self._prev_host_dirty = True

    def _assign_previous(self, field, src):
        """Whole-field assignment `u_prev <- u_sol` / `p_prev <- p_sol` as a device-to-device copy.  Possible when the
        source is this solver's solution field and the device holds its current values; anything else goes through
        the host as before (returns False)."""
        if src is not (self._u_sol.x if field == 0 else self._p_sol.x) or not self._dev_newer["sol"]:
            return False
        if self._prev_host_dirty and not self._prev_dev_newer:
            # host writes to u_prev / p_prev not uploaded yet: bring the device up to date before overwriting one field
            self.ctx.set_state(u_prev=self._loc_u(self._u_prev.x._array), p_prev=self._loc_p(self._p_prev.x._array))
        self._prev_host_dirty = False
        self.ctx.advance_field(field)
        self._prev_dev_newer = True  # the host copies of u_prev / p_prev are stale now
        return True

    # -- reference API ---------------------------------------------------------
    def setup(self, bcu: list[BoundaryCondition], bcp: list[BoundaryCondition], facet_tags=None, tags=None) -> None:
        self.bcu_d = [bc.getBC(self.V) for bc in bcu]
        self.bcp_d = [bc.getBC(self.Q) for bc in bcp]
        self._bcs = [(0, bc) for bc in self.bcu_d] + [(1, bc) for bc in self.bcp_d]
        self._bc_cache = None
        self._bc_nodes = None
        self._bc_owned = None
        if facet_tags is not None:
            # the reference hands the facet tags over here, not at construction (scenario.py:146-149):
            # drag/lift by marker and the backflow marker follow them
            mk = np.zeros(self.mesh.num_facets, dtype=np.int32)
            mk[np.asarray(facet_tags.indices, dtype=np.int64)] = np.asarray(facet_tags.values, dtype=np.int32)
            self.ctx.set_facet_markers(mk if self._part is None else mk[self._part.facet_ids])
        self._upload_bcs()
        if self._part is not None:
            # one known-answer exchange + reduction through the communicator before trusting it (multi-GPU
            # RCCL cannot be rehearsed on the one-GPU development box); host-staged exchange as the safety net
            if not getattr(self, "_comm_checked", False):
                self._comm_checked = True
                bad = self._comm.selfcheck(self.ctx, self._dm)
                if bad and self._comm.backend == "rccl":
                    self._comm.fall_back_to_host(self.ctx, bad)
                    bad = self._comm.selfcheck(self.ctx, self._dm)
                if bad:
                    raise RuntimeError("communicator self-check failed: " + bad)
            # the pressure part of the preconditioner is solved globally (replicated) on every rank
            pnodes = np.unique(np.concatenate([bc.dofs for bc in self.bcp_d])) if self.bcp_d else np.zeros(0, np.int32)
            if not getattr(self, "_ds_terms", True):
                # do-nothing boundary: the vertices of the exterior facets that are not no-slip/inflow facets are
                # the Dirichlet set of the preconditioner's pressure Laplacian (as build_cc_host does per rank)
                fixed = np.zeros(self._dm.num_vertices, dtype=bool)
                for bc in self.bcu_d:
                    fixed[bc.dofs] = True
                fv = self._dm.facet_vertices
                open_f = ~np.all(fixed[fv], axis=1)
                pnodes = np.unique(np.concatenate([pnodes, fv[open_f].ravel()])).astype(np.int32)
            self.ctx.set_global_pressure_space(self._dm.x, self._dm.cells, self._part.owned_global, pnodes)
        # x_n = (u_prev, p_prev): initial guess of the first step (stabilized_schur.py:216-223)
        self._sync_previous()
        up, pp = self._loc_u(self._u_prev.x._array), self._loc_p(self._p_prev.x._array)
        self.ctx.set_state(u_prev=up, p_prev=pp, u=up, p=pp)
        self._prev_host_dirty = False

    def _bc_nodes_values(self, field, bc):
        g = bc.g.x._array
        nodes = bc.dofs
        vals = g.reshape(-1, self.mesh.geometry.dim)[nodes] if field == 0 else g[nodes]
        if self._part is not None:
            loc = self._part.g2l[nodes]
            keep = loc >= 0
            nodes, vals = loc[keep].astype(np.int32), vals[keep]
        return nodes, np.array(vals, copy=True)

    def _upload_bcs(self):
        items = [self._bc_nodes_values(fld, bc) for fld, bc in self._bcs]
        vals = [v for _, v in items]
        if self._bc_cache is not None and len(vals) == len(self._bc_cache) and all(
                np.array_equal(a, b) for a, b in zip(vals, self._bc_cache)):
            return
        nodes_now = [n for n, _ in items]
        same_sets = (self._bc_cache is not None and self._bc_nodes is not None and len(nodes_now) == len(self._bc_nodes)
                     and all(a is b or np.array_equal(a, b) for a, b in zip(nodes_now, self._bc_nodes)))
        if same_sets:
            # time-dependent data on unchanged dof sets (pulsatile inlet): only the objects whose values changed are re-sent,
            # and of those only the dofs they determine (a later object holding the same dof keeps its value)
            for k, ((fld, _), (nodes, v)) in enumerate(zip(self._bcs, items)):
                if not np.array_equal(v, self._bc_cache[k]):
                    own = self._bc_owned[k]
                    self.ctx.update_dirichlet(fld, nodes[own], v[own])
            self._bc_cache = vals
            return
        self.ctx.clear_dirichlet()
        for (fld, _), (nodes, v) in zip(self._bcs, items):
            self.ctx.add_dirichlet(fld, nodes, v)
        self._bc_cache = vals
        self._bc_nodes = nodes_now
        # dofs whose final value each object determines: not held by a LATER object of the same field
        self._bc_owned = []
        for k, ((fld, _), (nodes, _v)) in enumerate(zip(self._bcs, items)):
            later = [items[j][0] for j in range(k + 1, len(items)) if self._bcs[j][0] == fld]
            own = np.ones(len(nodes), dtype=bool) if not later else ~np.isin(nodes, np.concatenate(later))
            self._bc_owned.append(own)

    def solveStep(self):
        for _, bc in self._bcs:
            bc.update()  # stabilized_schur.py:170
        self._upload_bcs()
        if self._prev_host_dirty and not self._prev_dev_newer:
            self.transfers["uploads"] += 1
            self.ctx.set_state(u_prev=self._loc_u(self._u_prev.x._array), p_prev=self._loc_p(self._p_prev.x._array))
        self._prev_host_dirty = False
        st = self.ctx.solve_step()  # raises RuntimeError("Did not converge, reason: r.") like :332-334
        self.last_stats = st
        self._dev_newer["sol"] = True
        self._dev_newer["res"] = True
        if not self._quiet and self.mesh.comm.rank == 0:
            print(f"Solver converged in {st.newton_its} nonlinear iterations"
                  f" (with total number of {st.krylov_its} linear iterations)")

    # -- device-resident extras ---------------------------------------------------
    def advance(self):
        """u_prev <- u_sol, p_prev <- p_sol without leaving HBM (the copy of
        /root/reference/src/scenario.py:306-307)."""
        self.ctx.advance()
        self._prev_dev_newer = True
        self._prev_host_dirty = False

    def functional(self, kind, marker=0):
        """Global value (the library reduces over the ranks)."""
        return self.ctx.functional(kind, marker)
