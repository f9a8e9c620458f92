"""Lid-driven cavity (/root/reference/src/scenarios/lid_driven2D.py:14-75):
unit square, "right" diagonals, no-slip walls, lid u=(1,0) on the open top
edge, no pressure condition (singular pressure)."""
from __future__ import annotations

import numpy as np

from ..boundaryCondition import BoundaryCondition
from ..fem import Function
from ..mesh import create_unit_square, locate_entities_boundary
from ..scenario import Scenario


class LidDriven2DSimulation(Scenario):
    def __init__(self, solver_name, dt, T, f: tuple[float, float] = (0, 0), *, rho=1, mu=1, nx=50, **solver_kwargs):
        self._mesh = None
        self._bcu = None
        self._bcp = None
        self.Re = str(int(1 / mu))
        self.nx = int(nx)
        self.quiet = bool(solver_kwargs.get("quiet", False))
        super().__init__(solver_name, "lid_driven2D", rho, mu, dt, T, f, **solver_kwargs)
        self.setup()

    @property
    def mesh(self):
        if not self._mesh:
            self._mesh = create_unit_square(self.nx, self.nx)
        return self._mesh

    def _velocity_condition(self, where, value):
        """A BoundaryCondition holding the constant velocity `value` on the boundary facets selected by `where`."""
        g = Function(self.solver.V)
        g.x.array.reshape(-1, 2)[:] = value
        facet_dim = self.mesh.topology.dim - 1
        bc = BoundaryCondition(g)
        bc.initTopological(facet_dim, locate_entities_boundary(self.mesh, facet_dim, where))
        return bc

    @property
    def bcu(self):
        # order matters where the sets touch: the lid excludes the two top corners, which stay no-slip (lid_driven2D.py:33-52)
        if not self._bcu:
            self._bcu = [self._velocity_condition(self.walls, (0.0, 0.0)), self._velocity_condition(self.lid, (1.0, 0.0))]
        return self._bcu

    @property
    def bcp(self):
        if not self._bcp:
            self._bcp = []
        return self._bcp

    def initial_velocity(self, x):
        return np.zeros((self.mesh.geometry.dim, x.shape[1]))

    @staticmethod
    def lid(x):
        return np.isclose(x[1], 1.0) & (x[0] > 1e-10) & (x[0] < 1.0 - 1e-10)

    @staticmethod
    def walls(x):
        return np.logical_or.reduce((np.isclose(x[0], 0), np.isclose(x[0], 1), np.isclose(x[1], 0)))

    def centerline_u(self, ys):
        """u_x(0.5, y): the quantity of the Ghia tables the reference ships
        (src/benchmark_data/lid_driven2D/plot_u_y_Ghia*.csv)."""
        nx = self.nx
        u = self.solver.u_sol.x.array.reshape(-1, 2)[:, 0].reshape(nx + 1, nx + 1)  # [j (y), i (x)]
        xs = np.linspace(0, 1, nx + 1)
        col = np.array([np.interp(0.5, xs, u[j]) for j in range(nx + 1)])
        return np.interp(ys, xs, col)
