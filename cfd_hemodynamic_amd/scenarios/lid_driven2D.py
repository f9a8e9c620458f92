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

    @property
    def bcu(self):
        if not self._bcu:
            u_noslip = Function(self.solver.V)
            u_noslip.x.array[:] = 0
            fdim = self.mesh.topology.dim - 1
            bc_noslip = BoundaryCondition(u_noslip)
            bc_noslip.initTopological(fdim, locate_entities_boundary(self.mesh, fdim, self.walls))
            u_lid = Function(self.solver.V)
            u_lid.interpolate(lambda x: np.vstack((np.ones(x.shape[1]), np.zeros(x.shape[1]))))
            bc_lid = BoundaryCondition(u_lid)
            bc_lid.initTopological(fdim, locate_entities_boundary(self.mesh, fdim, self.lid))
            self._bcu = [bc_noslip, bc_lid]
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
