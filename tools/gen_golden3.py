#!/usr/bin/env python3
"""Generate tests/golden/cube3d.npz: golden vectors for the three 3-D element types (P1 and P2 tetrahedra, Q1 hexahedra) with the
NumPy twin oracle/np_twin_gen3.py in its OWN arithmetic (einsum element tensors, no C routine) -- inputs (node mesh of a small sheared
box, Dirichlet objects, a random state that violates them, history) and expected outputs (residual, every Jacobian entry, L2 norms).
They pin the element formulas AND the quadrature tables the three implementations share (include/cfdh_quad_tet.h: a change of the
table shows up here).  The reference itself cannot run in this image: these pin the build's own restatement.

    python tools/gen_golden3.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from gen3_util import facet_node_set3, node_mesh3, problem3  # noqa: E402
from oracle import np_twin_nd as TN  # noqa: E402


def case(kind, seed):
    rng = np.random.default_rng(seed)
    m = node_mesh3(kind, 2, distort=0.05)
    nv = m.num_vertices
    prm = TN.Params(0.02, 1.3, 0.04, (0.2, -0.1, 0.3), theta=1.0, a0=1.5, a1=-2.0, a2=0.5)
    pb = problem3(kind, m, prm)
    bnd = facet_node_set3(m, np.arange(m.num_facets))[::2]
    vals = rng.standard_normal((len(bnd), 3))
    pn = facet_node_set3(m, np.arange(m.num_facets))[1::5]
    pb.add_bc_u(bnd, vals)
    pb.add_bc_p(pn, 0.5 * np.ones(len(pn)))
    xv = 0.3 * rng.standard_normal(4 * nv)
    un, un2 = 0.3 * rng.standard_normal((nv, 3)), 0.3 * rng.standard_normal((nv, 3))
    F, J = pb.assemble(xv, un, un2=un2)
    J = J.tocsr()
    J.sort_indices()
    return {"x": m.x, "cells": m.cells, "facet_cells": m.facet_cells, "facet_local": m.facet_local,
            "params": np.array([prm.dt, prm.rho, prm.mu, prm.mu_facet, prm.theta, prm.a0, prm.a1, prm.a2]), "f": np.asarray(prm.f, dtype=float),
            "bcu_nodes": bnd, "bcu_vals": vals, "bcp_nodes": pn, "state": xv, "u_prev": un, "u_prev2": un2,
            "F": F, "J_data": J.data, "J_indices": J.indices, "J_indptr": J.indptr, "l2": np.asarray(pb.l2_norms(xv))}


if __name__ == "__main__":
    out = {}
    for kind, seed in (("P1", 11), ("P2", 12), ("Q1", 13)):
        for k, v in case(kind, seed).items():
            out["%s_%s" % (kind, k)] = v
    path = os.path.join(ROOT, "tests", "golden", "cube3d.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "size %.0f KB" % (os.path.getsize(path) / 1024))
