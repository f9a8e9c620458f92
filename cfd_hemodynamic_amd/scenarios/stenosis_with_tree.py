"""Stenosed artery with a vascular tree at its outlet -- boundary data and geometry parameters of
/root/reference/src/scenarios/stenosis_with_tree.py (BASELINE config 5):

  * inlet (marker 2, x = 0): parabola `4 v_max y (H - y) / H^2`, v_max = `inlet_max_velocity` (:114-121, :518-527);
  * walls (marker 4): no-slip, listed after the inlet (:123-128);
  * outlets (marker 3, end caps of the terminal branches): p = 0 (:131-141);
  * rho = 1.0, mu = 3.3e-6, defaults L = 0.03, H = 0.003, res = 1e-4, x_position_stenosis = 0.01 (:77-94);
    grades mild / moderate / severe (:59-63, "severe" also lengthens the artery to L = 0.05).

Geometry: `mesh.create_stenosis_tree` (the reference's outline; the tree from the Murray-law generator instead of
the external VascuSynth binary; own mesher instead of gmsh).  Extra here: `n_generations` and the other tree
parameters, and a pulsatile inlet `v_max (1 + pulse_amplitude sin(2 pi t / pulse_period))` -- the reference has no
time-dependent inlet (SURVEY.md headline facts); it travels through the reference's own mechanism, the Function
behind the inlet BoundaryCondition re-read by `bc.update()` before every residual (stabilized_schur.py:170), set
for the time level of the step about to be solved.
"""
from __future__ import annotations

import numpy as np

from ..boundaryCondition import BoundaryCondition
from ..fem import Function
from ..mesh import create_stenosis_tree
from ..scenario import Scenario

_MESH_KEYS = ("L", "H", "res", "x_position_stenosis", "severity", "slope", "tension", "n_generations", "gamma",
              "bifurcation_angle", "length_ratio", "asymmetry", "coupling_slope")


class StenosisWithTreeSimulation(Scenario):
    fluid_marker = 1
    inlet_marker = 2
    outlet_marker = 3
    wall_marker = 4

    stenosis_grades = {
        "mild": {"severity": 0.25, "slope": 0.3},
        "moderate": {"severity": 0.50, "slope": 0.5},
        "severe": {"severity": 0.75, "slope": 0.3, "L": 0.05},
    }

    def __init__(self, solver_name, dt, T, f: tuple[float, float] = (0, 0), grade="severe", inlet_max_velocity=1.5, *,
                 rho=1.0, mu=3.3e-6, pulse_amplitude=0.0, pulse_period=1.0, ramp_time=0.0, **kwargs):
        self._mesh = None
        self._ft = None
        self._bcu = None
        self._bcp = None
        self.inlet_max_velocity = float(inlet_max_velocity)
        self.pulse_amplitude, self.pulse_period = float(pulse_amplitude), float(pulse_period)
        self.ramp_time = float(ramp_time)  # > 0: smooth start (1 - cos(pi t / ramp_time)) / 2 instead of an impulsive one
        opts = {"L": 0.03, "H": 0.003, "res": 0.0001, "x_position_stenosis": 0.01}
        opts.update(self.stenosis_grades.get(grade, self.stenosis_grades["severe"]))
        opts.update({k: kwargs.pop(k) for k in list(kwargs) if k in _MESH_KEYS})  # explicit options win over the grade
        self.mesh_options = opts
        self.quiet = bool(kwargs.get("quiet", False))
        super().__init__(solver_name, "stenosis_with_tree", rho, mu, dt, T, f, **kwargs)
        self.setup()

    @property
    def mesh(self):
        if self._mesh is None:
            o = dict(self.mesh_options)
            o["x_sten"] = o.pop("x_position_stenosis")
            self._mesh, self._ft = create_stenosis_tree(o.pop("res"), **o)
        return self._mesh

    def inlet_factor(self, t):
        f = 1.0 + self.pulse_amplitude * np.sin(2.0 * np.pi * t / self.pulse_period)
        if self.ramp_time > 0.0 and t < self.ramp_time:
            f *= 0.5 * (1.0 - np.cos(np.pi * t / self.ramp_time))
        return f

    @staticmethod
    def inlet_velocity(v_max, y_max):
        def velocity(x):
            values = np.zeros((2, x.shape[1]))
            values[0] = 4.0 * v_max * x[1] * (y_max - x[1]) / (y_max ** 2)
            return values
        return velocity

    @property
    def bcu(self):
        if not self._bcu:
            H = self.mesh_options["H"]
            self._inlet_base = Function(self.solver.V)
            self._inlet_base.interpolate(self.inlet_velocity(self.inlet_max_velocity, H))
            self._u_inlet = Function(self.solver.V)
            self._u_inlet.x.array[:] = self._inlet_base.x.array * self.inlet_factor(self.dt)
            bc_in = BoundaryCondition(self._u_inlet)
            bc_in.initTopological(1, self._ft.find(self.inlet_marker))
            bc_w = BoundaryCondition(Function(self.solver.V))
            bc_w.initTopological(1, self._ft.find(self.wall_marker))
            self._bcu = [bc_in, bc_w]
        return self._bcu

    @property
    def bcp(self):
        if not self._bcp:
            bc_o = BoundaryCondition(Function(self.solver.Q))
            bc_o.initTopological(1, self._ft.find(self.outlet_marker))
            self._bcp = [bc_o]
        return self._bcp

    def initial_velocity(self, x):
        return np.zeros((2, x.shape[1]))

    def set_inlet_time(self, t):
        """Inlet data for the step that ends at time t."""
        if self.pulse_amplitude != 0.0 or self.ramp_time > 0.0:
            # only the entries of the inlet profile change (the reference re-interpolates the whole field; at 8 M DOF that is
            # 130 MB of host traffic per step, 5.8 ms of a 29 ms step)
            if getattr(self, "_inlet_idx", None) is None:
                # the dofs the inlet condition reads: both velocity components of the vertices of the inlet facets
                base = np.asarray(self._inlet_base.x.array)
                nodes = np.unique(np.asarray(self.mesh.facet_vertices)[np.asarray(self._ft.find(self.inlet_marker), dtype=np.int64)])
                self._inlet_idx = (2 * nodes[:, None] + np.arange(2)[None, :]).ravel()
                self._inlet_vals = base[self._inlet_idx].copy()
            arr = self._u_inlet.x.array
            arr[self._inlet_idx] = self._inlet_vals * self.inlet_factor(t)

    def solve(self, output_folder=None, afterStepCallback=None, **kw):
        def after(t):
            self.set_inlet_time(t + self.dt)  # read by bc.update() at the start of the next step
            if afterStepCallback:
                afterStepCallback(t)
        return super().solve(output_folder, after, **kw)

    def outlet_flow_rates(self):
        """Volume flux through every terminal cap (host post-processing): sum over its facets of |e| (u_a + u_b)/2 . n."""
        m = self.mesh
        u = np.asarray(self.solver.u_sol.x.array).reshape(-1, 2)
        fv = m.facet_vertices
        mid = m.facet_midpoints()
        out = []
        for p0, p1 in m.outlet_caps:
            e = p1 - p0
            ln = np.linalg.norm(e)
            w = mid - p0
            s = (w @ e) / (ln * ln)
            dist = np.abs(w[:, 0] * e[1] - w[:, 1] * e[0]) / ln
            sel = np.nonzero((m.facet_marker == self.outlet_marker) & (dist < 1e-6 * ln + 0.31 * self.mesh_options["res"]) & (s > -0.05) & (s < 1.05))[0]
            t = m.x[fv[sel, 1]] - m.x[fv[sel, 0]]
            n = np.stack([t[:, 1], -t[:, 0]], 1)  # |e| * unit normal (sign fixed below)
            q = (0.5 * (u[fv[sel, 0]] + u[fv[sel, 1]]) * n).sum(1)
            cen = m.x[m.cells[m.facet_cells[sel]]].mean(axis=1)
            sign = np.sign(((mid[sel] - cen) * n).sum(1))  # outward
            out.append(float((q * sign).sum()))
        return np.array(out)
