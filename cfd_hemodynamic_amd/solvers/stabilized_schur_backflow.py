"""Drop-in `Solver` for the reference's `--solver stabilized_schur_backflow`
(/root/reference/src/solvers/stabilized_schur_backflow.py:50-350): the `stabilized_schur` step with

* a do-nothing outlet -- the `dot(p n, v) ds - dot(mu grad(u_mid) n, v) ds` pair of the base
  solver is not imposed (:107),
* backflow stabilisation `-beta rho (u_prev.n)_- (u_mid . v) ds_out` on the facets tagged
  `tags["outlet"]` (:158-176; Moghadam et al. 2011, eq. 10),
* no pressure Dirichlet condition: `bcp` is ignored (:193-195).

Same gfx950 kernels; `cfdh_set_boundary_terms(ds_terms=0, outlet marker, beta)` selects the
facet terms.  Constructor as in the reference: `v_max` is required (ValueError otherwise, :66-70),
`p_grade` 1 or 2 (P2/P2 on the node set of `elements.NodeMesh`, quadrature kernels of csrc/cfdh_gen.hip), `beta_backflow` defaults to 0.2.
"""
from __future__ import annotations

from typing import Callable

import numpy as np

from ..boundaryCondition import BoundaryCondition
from .stabilized_schur import Solver as _MidpointSolver


class Solver(_MidpointSolver):
    MAX_ITER = 20

    def __init__(self, mesh, dt: float, rho: float, mu: float, f: list,
                 initial_velocity: Callable[[np.ndarray], np.ndarray] = None,
                 v_max: float = None, p_grade: int = 1, beta_backflow: float = 0.2, **kwargs):
        if v_max is None:
            raise ValueError("v_max is required for stabilized_schur_backflow. Pass it via CLI: --v_max <value>")
        if int(p_grade) not in (1, 2):
            raise NotImplementedError("p_grade=%r: P1/P1 and P2/P2 run on the gfx950 kernels" % (p_grade,))
        self.p_grade = int(p_grade)
        kwargs["_degree"] = self.p_grade
        self.v_max = float(v_max)
        self.beta_backflow = float(beta_backflow)
        for k in ("p_inlet", "p_outlet", "beta_nitsche", "R_resistance", "initial_ffr"):
            kwargs.pop(k, None)  # scenario-level keywords of the other stenosis solvers (stenosis.py:84-99)
        super().__init__(mesh, dt, rho, mu, f, initial_velocity, **kwargs)
        if mesh.comm.rank == 0 and not self._quiet:
            print(f"[Solver] p_grade={p_grade}, v_max={self.v_max:.4f}, beta_backflow={self.beta_backflow:.2f}", flush=True)

    def setup(self, bcu: list[BoundaryCondition], bcp: list[BoundaryCondition], facet_tags=None, tags=None) -> None:
        if tags is None or "outlet" not in tags:
            raise KeyError("outlet")  # the reference indexes tags["outlet"] (:162)
        self._ds_terms = False
        self.ctx.set_boundary_terms(ds_terms=False, backflow_marker=int(tags["outlet"]), beta=self.beta_backflow)
        super().setup(bcu, [], facet_tags, tags)  # self.bcp_d = [] (:195)
