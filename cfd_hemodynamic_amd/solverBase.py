"""`SolverBase` with the reference's plugin surface
(/root/reference/src/solverBase.py:25-195): Constants dt/rho/mu/f, spaces V/Q,
state Functions u_sol/p_sol/u_prev/p_prev/u_residual/p_residual, the stress
helpers and `assemble_wss`."""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Callable

import numpy as np

from .boundaryCondition import BoundaryCondition
from .fem import Constant, Function, FunctionSpace


class SolverBase(ABC):
    @abstractmethod
    def __init__(self, mesh, dt: float, rho: float, mu: float, f: list,
                 initial_velocity: Callable[[np.ndarray], np.ndarray] = None):
        self.mesh = mesh
        self.dt = Constant(mesh, dt)
        self.rho = Constant(mesh, rho)
        self.mu = Constant(mesh, mu)
        self.f = Constant(mesh, f)
        self._u_sol = self._p_sol = self._u_prev = self._p_prev = None
        self._V = self._Q = None

    def _need(self, attr: str, initialiser: str):
        value = getattr(self, attr)
        if value is None:
            raise AssertionError(f"{type(self).__name__}.{attr.lstrip('_')} does not exist before {initialiser}() was called")
        return value

    u_sol = property(lambda self: self._need("_u_sol", "initVelocitySpace"))
    u_prev = property(lambda self: self._need("_u_prev", "initVelocitySpace"))
    V = property(lambda self: self._need("_V", "initVelocitySpace"))
    p_sol = property(lambda self: self._need("_p_sol", "initPressureSpace"))
    p_prev = property(lambda self: self._need("_p_prev", "initPressureSpace"))
    Q = property(lambda self: self._need("_Q", "initPressureSpace"))

    @abstractmethod
    def setup(self, bcu: list[BoundaryCondition], bcp: list[BoundaryCondition]) -> None:
        pass

    @abstractmethod
    def solveStep(self) -> None:
        pass

    def _dof_mesh(self, degree):
        """The node set of Lagrange elements of `degree` on the mesh: the mesh itself for P1 / Q1, vertices + edge midpoints
        for P2 triangles (`p_grade = 2`, stabilized_schur_backflow.py:84-87)."""
        from .elements import dof_mesh
        dm = dof_mesh(self.mesh, degree)
        if getattr(self, "_dm", None) not in (None, dm):
            raise ValueError("velocity and pressure must use the same Lagrange degree (equal-order elements)")
        self._dm = dm
        return dm

    def initVelocitySpace(self, family, cell, degree, shape=None) -> None:
        if str(family) not in ("Lagrange", "CG", "P"):
            raise ValueError("only Lagrange spaces are implemented")
        dm = self._dof_mesh(degree)
        self._V = FunctionSpace(dm, self.mesh.geometry.dim if shape is None else int(shape[0]))
        self._u_sol = Function(self.V, name="velocity")
        self._u_prev = Function(self.V)
        self.u_residual = Function(self.V, name="u_residual")

    def initPressureSpace(self, family, cell, degree, shape=None) -> None:
        if str(family) not in ("Lagrange", "CG", "P"):
            raise ValueError("only Lagrange spaces are implemented")
        self._Q = FunctionSpace(self._dof_mesh(degree), 1)
        self._p_sol = Function(self.Q, name="pressure")
        self._p_prev = Function(self.Q)
        self.p_residual = Function(self.Q, name="p_residual")

    def initStressForm(self):
        """Allocates the stress fields the Scenario writes
        (/root/reference/src/solverBase.py:144-173)."""
        dm = getattr(self, "_dm", None) or self.mesh
        self.normal_stress = Function(FunctionSpace(dm, 1), name="normal_stress")
        self.shear_stress = Function(FunctionSpace(dm, self.mesh.geometry.dim), name="shear_stress")

    def assemble_wss(self):
        """Wall shear stress  (1/|e|) oint w . (T - (T.n) n),  T = -sigma(u,p) n
        (/root/reference/src/solverBase.py:163-195), assembled on the host from
        the current solution.  Post-processing, outside the timed hot path."""
        if not hasattr(self, "shear_stress"):
            return
        mesh = self.mesh
        if mesh.geometry.dim != 2 or getattr(getattr(self, "_dm", mesh), "etype", 0) != 0:
            raise NotImplementedError("host restatement of the wall shear stress: P1 triangles only (the device path covers the other elements)")
        u = self.u_sol.x.array.reshape(-1, 2)
        mu = float(self.mu.value)
        fc, fl, fv = mesh.facet_cells, mesh.facet_local, mesh.facet_vertices
        cells = mesh.cells[fc]
        X = mesh.x[cells]
        det = (X[:, 1, 0] - X[:, 0, 0]) * (X[:, 2, 1] - X[:, 0, 1]) - (X[:, 1, 1] - X[:, 0, 1]) * (X[:, 2, 0] - X[:, 0, 0])
        g = np.empty((len(fc), 3, 2))
        g[:, 0, 0] = (X[:, 1, 1] - X[:, 2, 1]) / det
        g[:, 0, 1] = (X[:, 2, 0] - X[:, 1, 0]) / det
        g[:, 1, 0] = (X[:, 2, 1] - X[:, 0, 1]) / det
        g[:, 1, 1] = (X[:, 0, 0] - X[:, 2, 0]) / det
        g[:, 2, 0] = (X[:, 0, 1] - X[:, 1, 1]) / det
        g[:, 2, 1] = (X[:, 1, 0] - X[:, 0, 0]) / det
        idx = np.arange(len(fc))
        gf = g[idx, fl]
        n = -gf / np.linalg.norm(gf, axis=1)[:, None]
        G = np.einsum("cai,caj->cij", g, u[cells])  # d_i u_j
        E = 0.5 * (G + np.transpose(G, (0, 2, 1)))
        # tangential part of T = -(2 mu E - p I) n : the pressure part is purely normal
        T = -2.0 * mu * np.einsum("cij,cj->ci", E, n)
        Tt = T - np.einsum("ci,ci->c", T, n)[:, None] * n
        out = self.shear_stress.x.array.reshape(-1, 2)
        out[:] = 0.0
        # (1/|e|) oint lambda_a Tt = Tt / 2 per facet vertex
        np.add.at(out, fv[:, 0], 0.5 * Tt)
        np.add.at(out, fv[:, 1], 0.5 * Tt)

    @staticmethod
    def _simplex_gradients(mesh):
        """grad(lambda_a) of every cell of a P1 simplex mesh, [nc, d + 1, d]."""
        X = np.asarray(mesh.x)[np.asarray(mesh.cells)]
        d = X.shape[2]
        if X.shape[1] != d + 1:
            raise NotImplementedError("epsilon / sigma: cell-wise values exist for P1 simplices (constant gradients) only")
        M = X[:, 1:, :] - X[:, :1, :]                     # rows x_a - x_0
        Minv = np.linalg.inv(M)                           # columns = grad(lambda_a), a >= 1
        g = np.empty((len(X), d + 1, d))
        g[:, 1:, :] = np.transpose(Minv, (0, 2, 1))
        g[:, 0, :] = -g[:, 1:, :].sum(axis=1)
        return g

    @staticmethod
    def epsilon(u):
        """`sym(nabla_grad(u))` of solverBase.py:176-178, evaluated: the reference returns the UFL expression, this mirror has no
        form language, so it returns what the expression IS on a P1 field -- one symmetric d x d matrix per cell, [nc, d, d]
        (`nabla_grad(u)[i, j] = d_i u_j`)."""
        mesh = u.function_space.mesh
        g = SolverBase._simplex_gradients(mesh)
        uv = np.asarray(u.x.array, dtype=float).reshape(-1, u.function_space.bs)[np.asarray(mesh.cells)]
        G = np.einsum("cai,caj->cij", g, uv)
        return 0.5 * (G + np.transpose(G, (0, 2, 1)))

    @staticmethod
    def sigma(u, p, mu):
        """`2 mu sym(nabla_grad(u)) - p I` of solverBase.py:180-182 per cell, [nc, d, d]; p enters with its cell mean (the value of
        a P1 field at the centroid)."""
        E = SolverBase.epsilon(u)
        mesh = u.function_space.mesh
        pc = np.asarray(p.x.array, dtype=float)[np.asarray(mesh.cells)].mean(axis=1)
        d = E.shape[1]
        return 2.0 * float(mu) * E - pc[:, None, None] * np.eye(d)[None, :, :]
