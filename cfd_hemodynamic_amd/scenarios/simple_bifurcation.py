"""3-D bifurcation with the boundary data and non-dimensionalisation of
/root/reference/src/scenarios/simple_bifurcation.py (BASELINE config 5, 3-D variant):

  * Re = rho_real U_c L_c / mu_real with rho_real 1055, mu_real 3.5e-3, U_c 0.01, L_c = (100 / r_mesh_in) / 1e6 (:20-26,45-48),
    rho = 1, mu = 1 / Re;
  * walls (tag 11) no-slip, THEN the inlet (tag 8) `u_y = v_inlet (1 - (r / r_mesh_in)^2)`, r^2 = x^2 + z^2 (:77-99,123-133);
  * outlets (tags 9, 10): p = p_outlet / (rho_real U_c^2) (:101-121).

The reference reads `simple_bifurcation.msh` (a gmsh file that is not part of the repository); here the vessel is
generated (`mesh3d.create_bifurcation`: voxel tetrahedra of an implicit Y-shaped tube, inlet disc of radius r_mesh_in in
the plane y = 0).  On the staircase rim of that disc the wall and the inlet condition share vertices at r < r_mesh_in; the
inlet source is zeroed there so that both DirichletBC objects hold the same value (on a body-fitted mesh the rim lies at
r = r_mesh_in, where the profile vanishes by itself).

Tolerance caveat (inherited, not changed): the solver removes the pressure mean from the initial guess of every step
(stabilized_schur.py:319) although p is prescribed at the outlets, so each step starts with an O(|p|) misfit in the
outlet rows.  On this millimetre-sized mesh the momentum and continuity rows are ~1e-9 of that misfit, so `snes_rtol`
relative to it is met before those rows are solved; from ~1e6 DOF on the steps after the first then end after one Krylov
iteration with a meaningless field.  `options={"remove_p_mean": 0}` skips the removal (the solve is then relative to the
PDE rows alone); bench.py --config c5b does so and says so in its `config`."""
from __future__ import annotations

import numpy as np

from ..boundaryCondition import BoundaryCondition
from ..fem import Function
from ..mesh3d import create_bifurcation
from ..scenario import Scenario


class MicrovasculatureSimulation(Scenario):
    fluid_tag = 7
    inlet_tag = 8
    outlet1_tag = 9
    outlet2_tag = 10
    wall_tag = 11

    rho_real = 1055.0
    mu_real = 3.5e-3
    r_mesh_in = 0.003918604
    r_mesh_out2 = 0.000922768
    L_c = (100 / r_mesh_in) / 1e6
    U_c = 0.01

    def __init__(self, solver_name, dt, T, f: tuple[float, float, float] = (0, 0, 0), v_inlet=1.5, p_outlet1=0, p_outlet2=0, *,
                 rho=None, mu=None, res=4.0e-4, mesh_file=None, **solver_kwargs):
        self._mesh = None
        self._ft = None
        self._bcu = None
        self._bcp = None
        self.res = float(res)
        self.mesh_file = mesh_file  # a gmsh .msh (the reference's meshes/simple_bifurcation.msh, :71-75) or an .xdmf pair
        self.Re = self.rho_real * self.U_c * self.L_c / self.mu_real
        p_c = self.rho_real * self.U_c ** 2
        self.v_inlet = float(v_inlet)
        self.p_outlet1_adim = float(p_outlet1) / p_c
        self.p_outlet2_adim = float(p_outlet2) / p_c
        self.quiet = bool(solver_kwargs.get("quiet", False))
        if not self.quiet:
            print(f"MicrovasculatureSimulation (Simple Bifurcation): Reynolds = {self.Re}")
        super().__init__(solver_name, "simple_bifurcation", 1.0, 1.0 / self.Re, dt, T, f, **solver_kwargs)
        self.setup()

    # tags of the Scenario surface (scenario.py:137-144)
    inlet_marker = inlet_tag
    outlet_marker = outlet1_tag
    wall_marker = wall_tag

    @property
    def mesh(self):
        if self._mesh is None:
            if self.mesh_file and str(self.mesh_file).lower().endswith(".xdmf"):
                from ..xdmf import read_xdmf
                self._mesh, self._ft = read_xdmf(self.mesh_file, "mesh", "mesh_tags")
            elif self.mesh_file:
                from ..meshio import read_msh
                self._mesh, self._ft = read_msh(self.mesh_file)
            else:
                self._mesh, self._ft = create_bifurcation(self.res, r_in=self.r_mesh_in)
        return self._mesh

    @staticmethod
    def inlet_velocity(v_max, r_max):
        def velocity(x):
            values = np.zeros((3, x.shape[1]))
            r = (x[0] ** 2 + x[2] ** 2) ** 0.5
            values[1] = v_max * (1 - (r / r_max) ** 2)
            return values
        return velocity

    @property
    def bcu(self):
        if not self._bcu:
            fdim = 2
            walls = self._ft.find(self.wall_tag)
            bc_w = BoundaryCondition(Function(self.solver.V))
            bc_w.initTopological(fdim, walls)
            u_in = Function(self.solver.V)
            u_in.interpolate(self.inlet_velocity(self.v_inlet, self.r_mesh_in))
            rim = np.unique(self.mesh.facet_vertices[walls])
            u_in.x.array.reshape(-1, 3)[rim] = 0.0
            bc_in = BoundaryCondition(u_in)
            bc_in.initTopological(fdim, self._ft.find(self.inlet_tag))
            self._bcu = [bc_w, bc_in]
        return self._bcu

    @property
    def bcp(self):
        if not self._bcp:
            out = []
            for tag, val in ((self.outlet1_tag, self.p_outlet1_adim), (self.outlet2_tag, self.p_outlet2_adim)):
                pf = Function(self.solver.Q)
                pf.x.array[:] = val
                bc = BoundaryCondition(pf)
                bc.initTopological(2, self._ft.find(tag))
                out.append(bc)
            self._bcp = out
        return self._bcp

    def initial_velocity(self, x):
        return np.zeros((3, x.shape[1]))

    def flow_rates(self):
        """Volume flux through inlet (entering) and the two outlets (leaving), evaluated on the device."""
        s = self.solver
        return -s.functional(7, self.inlet_tag), s.functional(7, self.outlet1_tag), s.functional(7, self.outlet2_tag)
