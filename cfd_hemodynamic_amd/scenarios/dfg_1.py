"""DFG 2D-1 benchmark (Re=20) with the reference's boundary data and
post-processing (/root/reference/src/scenarios/dfg_1.py:17-255); the gmsh
geometry is replaced by the deterministic block generator
`mesh.create_dfg_channel(m)` (`m`: cells across the channel height; m=18 is
about the reference's coarse mesh, m=200 gives ~1M DOF)."""
from __future__ import annotations

import numpy as np

from ..boundaryCondition import BoundaryCondition
from ..fem import Function
from ..mesh import create_dfg_channel
from ..scenario import Scenario


class DFG1Benchmark(Scenario):
    fluid_marker = 1
    inlet_marker = 2
    outlet_marker = 3
    wall_marker = 4
    obstacle_marker = 5

    def __init__(self, solver_name, dt, T, f: tuple[float, float] = (0, 0), *, rho=1, mu=1 / 1000, m=18, mesh_file=None,
                 **solver_kwargs):
        self._mesh = None
        self._ft = None
        self._bcu = None
        self._bcp = None
        self.mu = mu
        self.rho = rho
        self.m = int(m)
        # a gmsh file with the reference's markers (inlet 2, outlet 3, walls 4, obstacle 5) takes the place of the
        # XDMF the reference loads when present (dfg_1.py:42-48); otherwise the block-structured generator
        self.mesh_file = mesh_file
        self.quiet = bool(solver_kwargs.get("quiet", False))
        super().__init__(solver_name, "dfg_1", rho, mu, dt, T, f, **solver_kwargs)
        self.setup()

    @property
    def mesh(self):
        if not self._mesh:
            if self.mesh_file and str(self.mesh_file).lower().endswith(".xdmf"):
                from ..xdmf import read_xdmf  # dfg_1.py:43-48: read_mesh(name="Grid"), read_meshtags(name="Facet markers")
                self._mesh, self._ft = read_xdmf(self.mesh_file, "Grid", "Facet markers")
            elif self.mesh_file:
                from ..meshio import read_msh
                self._mesh, self._ft = read_msh(self.mesh_file)
            else:
                self._mesh, self._ft = create_dfg_channel(self.m)
        return self._mesh

    @property
    def bcu(self):
        if not self._bcu:
            fdim = self.mesh.topology.dim - 1
            u_inlet = Function(self.solver.V)
            u_inlet.interpolate(self.inlet_velocity)
            bcu_inflow = BoundaryCondition(u_inlet)
            bcu_inflow.initTopological(fdim, self._ft.find(self.inlet_marker))
            u_nonslip = Function(self.solver.V)
            u_nonslip.x.array[:] = 0
            bcu_walls = BoundaryCondition(u_nonslip)
            bcu_walls.initTopological(fdim, self._ft.find(self.wall_marker))
            bcu_obstacle = BoundaryCondition(u_nonslip)
            bcu_obstacle.initTopological(fdim, self._ft.find(self.obstacle_marker))
            self._bcu = [bcu_inflow, bcu_obstacle, bcu_walls]
        return self._bcu

    @property
    def bcp(self):
        if not self._bcp:
            fdim = self.mesh.topology.dim - 1
            pr = Function(self.solver.Q)
            pr.x.array[:] = 0
            bc_outflow = BoundaryCondition(pr)
            bc_outflow.initTopological(fdim, self._ft.find(self.outlet_marker))
            self._bcp = [bc_outflow]
        return self._bcp

    def initial_velocity(self, x):
        return np.zeros((self.mesh.geometry.dim, x.shape[1]))

    @staticmethod
    def inlet_velocity(x):
        values = np.zeros((2, x.shape[1]))
        values[0] = 4 * 0.3 * x[1] * (0.41 - x[1]) / (0.41**2)
        return values

    def drag_lift(self):
        """500*F_D, 500*F_L over the obstacle (dfg_1.py:183-211), evaluated on the device."""
        FD = self.solver.functional(0, self.obstacle_marker)
        FL = self.solver.functional(1, self.obstacle_marker)
        return 500 * FD, 500 * FL

    def pressure_difference(self):
        """p(0.15,0.2) - p(0.25,0.2) by point evaluation (dfg_1.py:213-253)."""
        v = self.mesh.eval_p1(self.solver.p_sol.x.array, [(0.15, 0.2), (0.25, 0.2)])
        if np.isnan(v).any():
            return None
        return v[0] - v[1]

    def solve(self, output_folder=None, afterStepCallback=None, **kw):
        out_path = super().solve(output_folder, afterStepCallback, **kw)
        self.drag, self.lift = self.drag_lift()
        self.p_diff = self.pressure_difference()
        if self.mesh.comm.rank == 0:
            if not self.quiet:
                print(f"Drag: {self.drag}")
                print(f"Lift: {self.lift}")
                print(f"Pressure difference: {self.p_diff}")
            if out_path:
                with open(f"{out_path}/drag_lift.txt", "w") as f:
                    f.write(f"Drag: {self.drag}\n")
                    f.write(f"Lift: {self.lift}\n")
                if self.p_diff is not None:
                    with open(f"{out_path}/pressure_diff.txt", "w") as f:
                        f.write(f"Pressure difference: {self.p_diff}\n")
        return out_path
