"""Pressure-driven 2-D channel flow on a QUADRILATERAL rectangle mesh -- boundary data and constants of
/root/reference/src/scenarios/unit_square_pipe.py:

  * mesh: `create_rectangle([[0, 0], [L, H]], [NX, NY], cell_type=CellType.quadrilateral)` with L = 80, H = 1.5 (mm),
    NX = 587, NY = 11 (:43-52, :101-105) -> 7 056 nodes, 21 168 Q1/Q1 DOF;
  * facets with x = 0 -> inlet (1), x = L -> outlet (2), y = 0 | H -> wall (3) (:109-132);
  * velocity: no-slip on the walls only (:137-148); pressure: Dirichlet `p_inlet` on the inlet and `p_outlet` on the
    outlet (:150-171); zero initial velocity; blood in mm-g-s units, rho = 1.06e-3, mu = 3.5e-3 (:65-66).

With `--solver stabilized_schur` the spaces are ("Lagrange", "quadrilateral", 1): Q1/Q1 -- SURVEY.md section 8f-4.
`nx`, `ny` override the resolution (tests).
"""
from __future__ import annotations

import numpy as np

from ..boundaryCondition import BoundaryCondition
from ..elements import create_rectangle
from ..fem import Function
from ..mesh import locate_entities_boundary, meshtags
from ..scenario import Scenario

_L = 80.0
_H = 1.5
_NX = 587
_NY = 11


class UnitSquarePipeSimulation(Scenario):
    inlet_marker = 1
    outlet_marker = 2
    wall_marker = 3

    def __init__(self, solver_name, dt, T, f: tuple = (0.0, 0.0), *, rho: float = 1.06e-3, mu: float = 3.5e-3, p_inlet: float,
                 p_outlet: float, early_stop_tolerance: float = 1e-5, nx: int = _NX, ny: int = _NY, L: float = _L, H: float = _H,
                 **solver_kwargs):
        self.p_inlet, self.p_outlet = float(p_inlet), float(p_outlet)
        self.nx, self.ny, self.L, self.H = int(nx), int(ny), float(L), float(H)
        self._mesh = self._ft = self._bcu = self._bcp = None
        self.quiet = bool(solver_kwargs.get("quiet", False))
        # the reference hands p_inlet / p_outlet to the solver constructor too (:85-86); plugins without these keywords drop them
        super().__init__(solver_name, "unit_square_pipe", rho, mu, dt, T, list(f), early_stop_tolerance=early_stop_tolerance,
                         p_inlet=self.p_inlet, p_outlet=self.p_outlet, **solver_kwargs)
        self.setup()

    @property
    def mesh(self):
        if self._mesh is None:
            m = create_rectangle((0.0, 0.0), (self.L, self.H), (self.nx, self.ny), cell_type="quadrilateral")
            fdim = 1
            inlet = locate_entities_boundary(m, fdim, lambda x: np.isclose(x[0], 0.0))
            outlet = locate_entities_boundary(m, fdim, lambda x: np.isclose(x[0], self.L))
            wall = locate_entities_boundary(m, fdim, lambda x: np.isclose(x[1], 0.0) | np.isclose(x[1], self.H))
            idx = np.concatenate([inlet, outlet, wall])
            val = np.concatenate([np.full(len(inlet), self.inlet_marker), np.full(len(outlet), self.outlet_marker),
                                  np.full(len(wall), self.wall_marker)]).astype(np.int32)
            order = np.argsort(idx)
            self._ft = meshtags(m, fdim, idx[order], val[order])
            self._mesh = m
        return self._mesh

    @property
    def bcu(self):
        if self._bcu is None:
            bc = BoundaryCondition(Function(self.solver.V))  # zero
            bc.initTopological(1, self._ft.find(self.wall_marker))
            self._bcu = [bc]
        return self._bcu

    @property
    def bcp(self):
        if self._bcp is None:
            out = []
            for value, marker in ((self.p_inlet, self.inlet_marker), (self.p_outlet, self.outlet_marker)):
                g = Function(self.solver.Q)
                g.x.array[:] = value
                bc = BoundaryCondition(g)
                bc.initTopological(1, self._ft.find(marker))
                out.append(bc)
            self._bcp = out
        return self._bcp

    def initial_velocity(self, x):
        return np.zeros((2, x.shape[1]))
