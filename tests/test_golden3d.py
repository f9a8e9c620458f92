"""Committed golden vectors of the three 3-D element types (tests/golden/cube3d.npz, made by tools/gen_golden3.py with the NumPy
twin in its own arithmetic): the C oracle (CPU) and the HIP kernels (GPU) reproduce residual, every Jacobian entry and the L2 norms.
The vectors also pin the quadrature tables the three implementations share (include/cfdh_quad_tet.h, cfdh_quad_gl.h)."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from gen3_util import ETYPE3, LIB_ETYPE3
from oracle import np_twin_gen3 as G3, np_twin_nd as TN, orcg3

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "cube3d.npz"))


def _case(kind):
    g = {k[len(kind) + 1:]: GOLD[k] for k in GOLD.files if k.startswith(kind + "_")}
    dt, rho, mu, muf, theta, a0, a1, a2 = g["params"]
    prm = TN.Params(dt, rho, mu, tuple(g["f"]), theta=theta, a0=a0, a1=a1, a2=a2)
    assert prm.mu_facet == muf
    n = len(g["state"])
    J = sp.csr_matrix((g["J_data"], g["J_indices"], g["J_indptr"]), shape=(n, n))
    return g, prm, J


@pytest.mark.parametrize("kind", ["P1", "P2", "Q1"])
def test_c_oracle_reproduces_the_3d_golden_vectors(kind, monkeypatch):
    g, prm, J = _case(kind)
    monkeypatch.setattr(G3, "element_tensors", orcg3.element_tensors)  # the twin's assembly over the C element routine
    pb = G3.Problem(ETYPE3[kind], g["x"], g["cells"], g["facet_cells"], g["facet_local"], prm)
    pb.add_bc_u(g["bcu_nodes"], g["bcu_vals"])
    pb.add_bc_p(g["bcp_nodes"], 0.5 * np.ones(len(g["bcp_nodes"])))
    F, Jc = pb.assemble(g["state"], g["u_prev"], un2=g["u_prev2"])
    assert np.abs(F - g["F"]).max() <= 1e-13 * np.abs(g["F"]).max()
    assert abs(Jc - J).max() <= 1e-13 * abs(J).max()
    assert np.allclose(pb.l2_norms(g["state"]), g["l2"], rtol=1e-13)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["P1", "P2", "Q1"])
def test_kernels_reproduce_the_3d_golden_vectors(kind):
    from cfd_hemodynamic_amd import _lib
    g, prm, J = _case(kind)
    nv = len(g["x"])
    fm = np.zeros(len(g["facet_cells"]), dtype=np.int32)
    ctx = _lib.Context(g["x"], g["cells"], g["facet_cells"], g["facet_local"], fm, etype=LIB_ETYPE3[kind])
    ctx.set_params(prm.dt, prm.rho, prm.mu, mu_facet=prm.mu_facet, f=prm.f)
    ctx.set_time_scheme(prm.theta, prm.a0, prm.a1, prm.a2)
    ctx.add_dirichlet(0, g["bcu_nodes"], g["bcu_vals"])
    ctx.add_dirichlet(1, g["bcp_nodes"], 0.5 * np.ones(len(g["bcp_nodes"])))
    xv = g["state"]
    ctx.set_state(u_prev=g["u_prev"].ravel(), p_prev=np.zeros(nv), u=xv[: 3 * nv], p=xv[3 * nv:])
    ctx.set_previous2(g["u_prev2"].ravel())
    ctx.assemble(True)
    Fg = np.concatenate(ctx.get_residual())
    assert np.abs(Fg - g["F"]).max() <= 1e-12 * np.abs(g["F"]).max()
    assert abs(ctx.get_csr() - J).max() <= 1e-12 * abs(J).max()
    assert abs(ctx.functional(2) - g["l2"][0]) <= 1e-12 * g["l2"][0] and abs(ctx.functional(3) - g["l2"][1]) <= 1e-12 * g["l2"][1]
    ctx.close()


@pytest.mark.gpu
def test_closed_form_tetrahedral_kernels_reproduce_the_p1_golden_vectors():
    """The same P1 vectors through the closed-form 3-D kernels (csrc/cfdh3_kernels.hip: tau-moments on the 171-point rule)."""
    from cfd_hemodynamic_amd import _lib
    g, prm, J = _case("P1")
    nv = len(g["x"])
    fm = np.zeros(len(g["facet_cells"]), dtype=np.int32)
    ctx = _lib.Context(g["x"], g["cells"], g["facet_cells"], g["facet_local"], fm)
    ctx.set_params(prm.dt, prm.rho, prm.mu, mu_facet=prm.mu_facet, f=prm.f)
    ctx.set_time_scheme(prm.theta, prm.a0, prm.a1, prm.a2)
    ctx.add_dirichlet(0, g["bcu_nodes"], g["bcu_vals"])
    ctx.add_dirichlet(1, g["bcp_nodes"], 0.5 * np.ones(len(g["bcp_nodes"])))
    xv = g["state"]
    ctx.set_state(u_prev=g["u_prev"].ravel(), p_prev=np.zeros(nv), u=xv[: 3 * nv], p=xv[3 * nv:])
    ctx.set_previous2(g["u_prev2"].ravel())
    ctx.assemble(True)
    Fg = np.concatenate(ctx.get_residual())
    assert np.abs(Fg - g["F"]).max() <= 1e-12 * np.abs(g["F"]).max()
    assert abs(ctx.get_csr() - J).max() <= 1e-12 * abs(J).max()
    ctx.close()
