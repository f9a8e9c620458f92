"""Drop-in `Solver` for the reference's `--solver stabilized_schur_bdf2`
(/root/reference/src/solvers/stabilized_schur_bdf2.py:39-326): the same SUPG/PSPG/LSIC P1/P1
Newton/Schur-FGMRES step as `stabilized_schur`, but fully implicit in space (u_mid = u_sol,
:79-92) with the time term (a0 u + a1 u_prev + a2 u_prev2)/dt (:82-86,105-110): BDF1 on the
first step, BDF2 afterwards (:298-305).

Runs on the same gfx950 kernels: `cfdh_set_time_scheme(theta=1, a0, a1, a2)` selects the
scheme, `u_prev2` lives in HBM and is shifted on the device (`cfdh_shift_history`, the copy of
:324) at the end of every `solveStep()`.

Public attributes follow the reference: `u_prev2` (Function on V), `step_count`,
`bdf_a0/bdf_a1/bdf_a2` (objects with a `.value`, as the reference's Constants).
"""
from __future__ import annotations

from typing import Callable

import numpy as np

from ..fem import Constant, Function
from .stabilized_schur import Solver as _MidpointSolver


class Solver(_MidpointSolver):
    MAX_ITER = 20

    def __init__(self, mesh, dt: float, rho: float, mu: float, f: list,
                 initial_velocity: Callable[[np.ndarray], np.ndarray] = None, **kwargs):
        super().__init__(mesh, dt, rho, mu, f, initial_velocity, **kwargs)
        self._u_prev2 = Function(self.V)  # u at time n-1 (:72)
        self.bdf_a0 = Constant(mesh, 1.0)
        self.bdf_a1 = Constant(mesh, -1.0)
        self.bdf_a2 = Constant(mesh, 0.0)
        self.step_count = 0
        self._prev2_host_dirty = True
        self._prev2_dev_newer = False
        self._u_prev2.x._pre_access = self._sync_previous2
        self._u_prev2.x._post_access = self._mark_prev2_dirty
        self.ctx.set_time_scheme(1.0, 1.0, -1.0, 0.0)

    @property
    def u_prev2(self):
        return self._u_prev2

    def _sync_previous2(self):
        if self._prev2_dev_newer:
            self._prev2_dev_newer = False
            lu = self.ctx.get_previous2()
            if self._part is None:
                self._u_prev2.x._array[:] = lu
            else:
                self._u_prev2.x._array[:] = self._comm.allgather_owned(lu, 2, self.mesh.num_vertices)

    def _mark_prev2_dirty(self):
        self._prev2_host_dirty = True

    def solveStep(self):
        # BDF1 for the first step, BDF2 thereafter (:298-305)
        if self.step_count == 0:
            self.bdf_a0.value, self.bdf_a1.value, self.bdf_a2.value = 1.0, -1.0, 0.0
        else:
            self.bdf_a0.value, self.bdf_a1.value, self.bdf_a2.value = 1.5, -2.0, 0.5
        self.ctx.set_time_scheme(1.0, float(self.bdf_a0.value), float(self.bdf_a1.value), float(self.bdf_a2.value))
        if self._prev2_host_dirty and not self._prev2_dev_newer:
            self.ctx.set_previous2(self._loc_u(self._u_prev2.x._array))
        self._prev2_host_dirty = False
        # u_prev must be on the device before the shift below reads it
        super().solveStep()
        # u_prev2 <- u_prev for the next step (:323-325)
        self.ctx.shift_history()
        self._prev2_dev_newer = True
        self.step_count += 1
