#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/ with the NumPy twin
(direct sparse solves, oracle/np_twin.py).  The reference itself cannot run here
(SURVEY.md 8c), so these pin the build's own restatement: inputs (mesh, Dirichlet
objects, random state) and expected outputs (residual, Jacobian, two converged
time steps, drag/lift, L2 norms).

    python tools/gen_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from util import dfg_case, lid_case, make_twin  # noqa: E402


def gen(name, case, seed):
    pb = make_twin(case)
    nv = case.nv
    rng = np.random.default_rng(seed)
    xv = 0.1 * rng.standard_normal(3 * nv)
    un = 0.1 * rng.standard_normal((nv, 2))
    F, J = pb.assemble(xv, un)
    J = J.tocsr()
    J.sort_indices()
    # two time steps from rest, Newton to round-off with a direct solver
    x = np.zeros(3 * nv)
    u_prev = np.zeros((nv, 2))
    sols = []
    for _ in range(2):
        x[2 * nv:] -= x[2 * nv:].mean()
        x, hist = pb.newton(x, u_prev, rtol=1e-13, atol=1e-13)
        u_prev = x[: 2 * nv].reshape(-1, 2).copy()
        sols.append(x.copy())
    m = case.mesh
    out = dict(
        x=m.x, cells=m.cells, facet_cells=m.facet_cells, facet_local=m.facet_local, facet_marker=m.facet_marker,
        dt=case.dt, rho=case.rho, mu=case.mu, f=np.asarray(case.f, dtype=float),
        nbc=len(case.bcs), state=xv, u_prev=un, F=F, J_data=J.data, J_indices=J.indices, J_indptr=J.indptr,
        step1=sols[0], step2=sols[1], l2=np.asarray(pb.l2_norms(sols[1])),
    )
    for k, (field, nodes, vals) in enumerate(case.bcs):
        out["bc%d_field" % k] = field
        out["bc%d_nodes" % k] = nodes
        out["bc%d_vals" % k] = vals
    if "ft" in case.markers:
        obst = case.markers["ft"].find(5)
        out["obstacle_facets"] = obst
        out["drag_lift"] = np.asarray(pb.drag_lift(sols[1], obst, case.mu))
    path = os.path.join(ROOT, "tests", "golden", name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "nv", nv, "size %.0f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
    gen("dfg_m6", dfg_case(6), 1)
    gen("lid_n8", lid_case(8), 2)
