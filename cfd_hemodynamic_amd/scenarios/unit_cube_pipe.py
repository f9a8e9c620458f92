"""Pressure-driven 3-D pipe flow on a HEXAHEDRAL box mesh -- boundary data and constants of
/root/reference/src/scenarios/unit_cube_pipe.py:

  * mesh: `create_box([[0, 0, 0], [L, W, H]], [NX, NY, NZ], cell_type=CellType.hexahedron)` with L = 80, W = H = 1.5 (mm),
    NX = 213, NY = NZ = 4 (:44-54, :103-109) -> 214 x 5 x 5 = 5 350 nodes, 21 400 Q1/Q1 DOF;
  * facets with x = 0 -> inlet (1), x = L -> outlet (2), y = 0 | W or z = 0 | H -> wall (3) (:111-139);
  * velocity: no-slip on the walls only (:143-155); pressure: Dirichlet `p_inlet` on the inlet and `p_outlet` on the outlet
    (:157-177); zero initial velocity; blood in mm-g-s units, rho = 1.06e-3, mu = 3.5e-3 (:70-71).

With `--solver stabilized_schur` the spaces are ("Lagrange", "hexahedron", 1): Q1/Q1 on hexahedra -- SURVEY.md section 8f-4.
`nx`, `ny`, `nz` override the resolution (tests, bench).  `cell_type="tetrahedron"` splits every brick into six Kuhn tetrahedra
(the duct as a workload for the tetrahedral element types; the reference builds hexahedra only).
"""
from __future__ import annotations

import numpy as np

from ..boundaryCondition import BoundaryCondition
from ..elements import create_box
from ..fem import Function
from ..mesh import locate_entities_boundary, meshtags
from ..scenario import Scenario

_L = 80.0
_W = 1.5
_H = 1.5
_NX = 213
_NY = 4
_NZ = 4


class UnitCubePipeSimulation(Scenario):
    inlet_marker = 1
    outlet_marker = 2
    wall_marker = 3

    def __init__(self, solver_name, dt, T, f: tuple = (0.0, 0.0, 0.0), *, rho: float = 1.06e-3, mu: float = 3.5e-3, p_inlet: float,
                 p_outlet: float, early_stop_tolerance: float = 1e-5, nx: int = _NX, ny: int = _NY, nz: int = _NZ, L: float = _L,
                 W: float = _W, H: float = _H, cell_type: str = "hexahedron", **solver_kwargs):
        self.cell_type = str(cell_type).split(".")[-1]
        self.p_inlet, self.p_outlet = float(p_inlet), float(p_outlet)
        self.nx, self.ny, self.nz, self.L, self.W, self.H = int(nx), int(ny), int(nz), float(L), float(W), float(H)
        self._mesh = self._ft = self._bcu = self._bcp = None
        self.quiet = bool(solver_kwargs.get("quiet", False))
        # the reference hands p_inlet / p_outlet to the solver constructor too (:93-94); plugins without these keywords drop them
        super().__init__(solver_name, "unit_cube_pipe", rho, mu, dt, T, list(f), early_stop_tolerance=early_stop_tolerance,
                         p_inlet=self.p_inlet, p_outlet=self.p_outlet, **solver_kwargs)
        self.setup()

    @property
    def mesh(self):
        if self._mesh is None:
            if self.cell_type == "tetrahedron":
                from ..mesh3d import Mesh3D, _voxel_tets
                xs, ys, zs = (np.linspace(0.0, hi, n + 1) for hi, n in ((self.L, self.nx), (self.W, self.ny), (self.H, self.nz)))
                cells, pts = _voxel_tets(np.ones((self.nx, self.ny, self.nz), bool), xs, ys, zs)
                m = Mesh3D(cells, pts, name="unit_cube_pipe")
            else:
                m = create_box((0.0, 0.0, 0.0), (self.L, self.W, self.H), (self.nx, self.ny, self.nz), cell_type="hexahedron")
            fdim = 2
            inlet = locate_entities_boundary(m, fdim, lambda x: np.isclose(x[0], 0.0))
            outlet = locate_entities_boundary(m, fdim, lambda x: np.isclose(x[0], self.L))
            wall = locate_entities_boundary(m, fdim, lambda x: np.isclose(x[1], 0.0) | np.isclose(x[1], self.W) | np.isclose(x[2], 0.0) | np.isclose(x[2], self.H))
            if self.cell_type == "tetrahedron":
                # the reference's marker (:127-137) is vertex-wise; on triangles a corner facet of the inlet / outlet plane has all its
                # vertices on SOME wall plane without lying in one: keep those with the inlet / outlet
                wall = np.setdiff1d(wall, np.concatenate([inlet, outlet]))
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
            bc.initTopological(2, self._ft.find(self.wall_marker))
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
                bc.initTopological(2, self._ft.find(marker))
                out.append(bc)
            self._bcp = out
        return self._bcp

    def initial_velocity(self, x):
        return np.zeros((3, x.shape[1]))
