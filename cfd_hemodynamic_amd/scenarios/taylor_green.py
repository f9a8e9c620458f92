"""Decaying Taylor-Green vortex on the unit square: the 2-D counterpart of the reference's analytic
scenario (/root/reference/src/scenarios/taylor_green.py:12-139 solves the 3-D Ethier-Steinman field on a
unit cube; this package is 2-D), with the same structure: the exact velocity AND
pressure are imposed as time-dependent Dirichlet data on the whole boundary through the Functions the
BoundaryConditions wrap (:22-27,65-72), `initial_velocity` is the exact field at t = 0, and
`exact_velocity` switches on the harness's error log (scenario.py:231-262).

    u = (-cos(kx) sin(ky), sin(kx) cos(ky)) exp(-2 nu k^2 t),   p = -rho/4 (cos 2kx + cos 2ky) exp(-4 nu k^2 t)

is an exact solution of the incompressible Navier-Stokes equations with f = 0 (k = 2 pi here)."""
from __future__ import annotations

import numpy as np

from ..boundaryCondition import BoundaryCondition
from ..fem import Function
from ..mesh import create_unit_square, locate_entities_boundary
from ..scenario import Scenario


class TaylorGreenSimulation(Scenario):
    K = 2.0 * np.pi

    def __init__(self, solver_name, dt, T, f: tuple[float, float] = (0, 0), *, rho=1, mu=1 / 50, nx=32, **solver_kwargs):
        self._mesh = None
        self._bcu = None
        self._bcp = None
        self._boundary_facets = None
        self.nx = int(nx)
        self.quiet = bool(solver_kwargs.get("quiet", False))
        self._rho, self._mu = float(rho), float(mu)
        super().__init__(solver_name, "taylor_green", rho, mu, dt, T, f, **solver_kwargs)
        self._u_bc = Function(self.solver.V)
        self._p_bc = Function(self.solver.Q)
        self._u_bc.interpolate(self.exact_velocity(0))
        self._p_bc.interpolate(self.exact_pressure(0))
        self.setup()

    @property
    def mesh(self):
        if not self._mesh:
            self._mesh = create_unit_square(self.nx, self.nx)
            self._boundary_facets = locate_entities_boundary(self._mesh, 1, lambda x: np.ones(x.shape[1], bool))
        return self._mesh

    @property
    def bcu(self):
        if not self._bcu:
            bc = BoundaryCondition(self._u_bc)
            bc.initTopological(1, self._boundary_facets)
            self._bcu = [bc]
        return self._bcu

    @property
    def bcp(self):
        if not self._bcp:
            bc = BoundaryCondition(self._p_bc)
            bc.initTopological(1, self._boundary_facets)
            self._bcp = [bc]
        return self._bcp

    def initial_velocity(self, x):
        return self.exact_velocity(0)(x)

    def solve(self, output_folder=None, afterStepCallback=None, **kw):
        def update_boundary_conditions(t):
            self._u_bc.interpolate(self.exact_velocity(t))
            self._p_bc.interpolate(self.exact_pressure(t))
            if afterStepCallback:
                afterStepCallback(t)

        return super().solve(output_folder, update_boundary_conditions, **kw)

    def exact_velocity(self, t):
        k, nu = self.K, self._mu / self._rho

        def velocity(x):
            d = np.exp(-2.0 * nu * k * k * t)
            return np.vstack((-np.cos(k * x[0]) * np.sin(k * x[1]) * d, np.sin(k * x[0]) * np.cos(k * x[1]) * d))

        return velocity

    def exact_pressure(self, t):
        k, nu, rho = self.K, self._mu / self._rho, self._rho

        def pressure(x):
            return -0.25 * rho * (np.cos(2 * k * x[0]) + np.cos(2 * k * x[1])) * np.exp(-4.0 * nu * k * k * t)

        return pressure
