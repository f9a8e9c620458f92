"""2-D stenosed channel with the boundary data of
/root/reference/src/scenarios/stenosis.py:124-156 (walls no-slip, parabolic
inlet `v_max (1 - ((y-R_in)/R_in)^2)` only when `v_max` is given) on the
structured y-profile mesh of `mesh.create_stenosis_channel`.
Units mm, g, s: rho = 1.06e-3 g/mm^3, mu = 3.5e-3 (stenosis.py:27-31).

Pressure: the reference passes NO pressure condition to `stabilized_schur` for this
scenario (its p_inlet/p_outlet kwargs are swallowed).  With the
form of stabilized_schur.py:79 (`+ p n.v ds` on every exterior facet) the constant
pressure is then in the null space of every Jacobian even with an open outlet, and the
Newton systems with a prescribed inflow are inconsistent (FGMRES stagnates at ~1e-2; the
CPU oracle shows the same).  `outlet_pressure=0.0` (default here) therefore fixes p at the
outlet like the other open-boundary scenarios (dfg_1.py:79-91); `outlet_pressure=None`
reproduces the reference literally."""
from __future__ import annotations

import os

import numpy as np

from ..boundaryCondition import BoundaryCondition
from ..fem import Function
from ..mesh import create_stenosis_channel
from ..scenario import Scenario


class StenosisSimulation(Scenario):
    fluid_marker = 1
    inlet_marker = 2
    outlet_marker = 3
    wall_marker = 4

    # stenosis.py:27-31
    stenosis_grades = {
        "mild": {"severity": 0.25, "slope": 0.3},
        "moderate": {"severity": 0.50, "slope": 0.3},
        "severe": {"severity": 0.75, "slope": 0.3},
    }

    def __init__(self, solver_name, dt, T, f: tuple[float, float] = (0, 0), grade="severe", *, rho=1.06e-3, mu=3.5e-3,
                 ny=None, res=0.15, L=138.0, R_in=1.57, R_out=1.2, x_sten=30.0, x_position_stenosis=None, severity=None,
                 slope=None, tension=0.5, v_max=None, outlet_pressure=0.0, **solver_kwargs):
        """Geometry defaults of stenosis.py:60-69 (L 138, R_in 1.57, R_out 1.2, stenosis at x = 30, res 0.15,
        severity 0.567, slope 0.4).  The reference writes those defaults into its mesh options BEFORE the grade
        table is consulted (:70-77: `if k not in self.mesh_options`), so `grade` never takes effect there: every
        grade builds the 0.567 / 0.4 throat unless severity / slope are passed explicitly.  That effective behaviour
        is reproduced (the table is kept as data, as in the reference).  `ny` (cells across the inlet) replaces `res`
        as the resolution parameter of the structured mesh: ny = round(2 R_in / res) when not given."""
        self._mesh = None
        self._ft = None
        self._bcu = None
        self._bcp = None
        self.grade = grade  # accepted and, like in the reference, without effect on the geometry (see above)
        self.severity = 0.567 if severity is None else severity
        self.slope = 0.4 if slope is None else slope
        self.tension = tension
        self.ny = int(ny) if ny is not None else max(4, int(round(2.0 * R_in / res)))
        self.L, self.R_in, self.R_out = L, R_in, R_out
        self.x_sten = x_sten if x_position_stenosis is None else x_position_stenosis
        self.v_max = v_max
        self.outlet_pressure = outlet_pressure
        self.quiet = bool(solver_kwargs.get("quiet", False))
        if v_max is not None:
            solver_kwargs["v_max"] = float(v_max)  # stenosis.py:92-94 (required by the backflow solver)
        super().__init__(solver_name, "stenosis", rho, mu, dt, T, f, **solver_kwargs)
        self.setup()

    @property
    def mesh(self):
        if not self._mesh:
            self._mesh, self._ft = create_stenosis_channel(self.ny, self.L, self.R_in, self.R_out, self.x_sten, self.severity, self.slope, self.tension)
        return self._mesh

    def inlet_profile(self, x):
        v = np.zeros((2, x.shape[1]))
        v[0] = self.v_max * (1.0 - ((x[1] - self.R_in) / self.R_in) ** 2)
        return v

    @property
    def bcu(self):
        if not self._bcu:
            fdim = 1
            u0 = Function(self.solver.V)
            bc_w = BoundaryCondition(u0)
            bc_w.initTopological(fdim, self._ft.find(self.wall_marker))
            self._bcu = [bc_w]
            if self.v_max is not None:
                ui = Function(self.solver.V)
                ui.interpolate(self.inlet_profile)
                bc_i = BoundaryCondition(ui)
                bc_i.initTopological(fdim, self._ft.find(self.inlet_marker))
                self._bcu = [bc_w, bc_i]  # walls first, inlet appended (stenosis.py:135,155)
        return self._bcu

    @property
    def bcp(self):
        if self._bcp is None:
            self._bcp = []
            if self.outlet_pressure is not None:
                pr = Function(self.solver.Q)
                pr.x.array[:] = float(self.outlet_pressure)
                bc_o = BoundaryCondition(pr)
                bc_o.initTopological(1, self._ft.find(self.outlet_marker))
                self._bcp = [bc_o]
        return self._bcp

    def solve(self, output_folder=None, afterStepCallback=None, **kw):
        result = super().solve(output_folder, afterStepCallback, **kw)
        self._compute_ffr(output_folder)
        return result

    def _compute_ffr(self, output_folder):
        """FFR = p_distal / p_proximal at the channel centreline y = R_in (stenosis.py:163-211)."""
        p = self.mesh.eval_p1(self.solver.p_sol.x.array, [(0.0, self.R_in), (self.L, self.R_in)])
        p = np.where(np.isnan(p), 0.0, p)
        self.p_proximal, self.p_distal = float(p[0]), float(p[1])
        self.ffr = self.p_distal / self.p_proximal if abs(self.p_proximal) > 1e-12 else float("nan")
        if self.mesh.comm.rank == 0:
            txt = "\n".join([f"p_proximal (inlet center):  {self.p_proximal:.6f}",
                             f"p_distal   (outlet center): {self.p_distal:.6f}",
                             f"FFR = p_distal / p_proximal: {self.ffr:.6f}"])
            if not self.quiet:
                print(f"\n[FFR] {txt}", flush=True)
            if output_folder:
                with open(os.path.join(output_folder, "ffr.txt"), "w") as f:
                    f.write(txt + "\n")

    def initial_velocity(self, x):
        """Zero without `v_max`; with it the flow-rate-conserving parabola of stenosis.py:219-259: at every x the
        profile `v_loc (1 - (r / R_loc)^2)` with v_loc R_loc = v_max R_in, R_loc from the linear taper and a cosine
        approximation of the narrowing (the reference's own approximation of its Bezier wall)."""
        if self.v_max is None:
            return np.zeros((2, x.shape[1]))
        R_taper = self.R_in + (self.R_out - self.R_in) * (x[0] / self.L)
        r_mid = self.R_in + (self.R_out - self.R_in) * (self.x_sten / self.L)
        h_sten = self.severity * r_mid
        dist_x = h_sten / self.slope if self.slope > 0 else self.L / 4
        dist_x = max(dist_x, self.L * 0.05)
        dist_x = min(dist_x, min(self.x_sten, self.L - self.x_sten) * 0.95)
        dx = np.abs(x[0] - self.x_sten)
        bump = np.where(dx < dist_x, h_sten * 0.5 * (1.0 + np.cos(np.pi * dx / dist_x)), 0.0)
        R_loc = np.maximum(R_taper - bump, 1e-6)
        v = np.zeros((2, x.shape[1]))
        v[0] = np.maximum(float(self.v_max) * self.R_in / R_loc * (1.0 - ((x[1] - self.R_in) / R_loc) ** 2), 0.0)
        return v
