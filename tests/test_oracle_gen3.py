"""Pins of oracle/np_twin_gen3.py (hexahedra Q1/Q1, P2/P2 tetrahedra; SURVEY.md section 8f-4, 3-D half) -- CPU only:
P1 tetrahedra through the quadrature twin == the closed-form nd twin; Jacobian == central difference of the residual;
patch tests (uniform flow, hydrostatic pressure); exact L2 norms and fluxes."""
import numpy as np
import pytest

from gen3_util import facet_node_set3, node_mesh3, problem3
from oracle import np_twin_gen3 as G3, np_twin_nd as TN

VARIANTS = [dict(), dict(theta=1.0, a0=1.5, a1=-2.0, a2=0.5), dict(ds_terms=False, beta_backflow=0.3)]


@pytest.mark.parametrize("kw", VARIANTS[:2])
def test_p1_tetrahedra_by_quadrature_equal_the_closed_form_twin(kw):
    rng = np.random.default_rng(1)
    m = node_mesh3("P1", 2, distort=0.1)
    nv = m.num_vertices
    prm = TN.Params(0.02, 1.3, 0.04, (0.2, -0.1, 0.3), **kw)
    u, un, un2 = (0.3 * rng.standard_normal((nv, 3)) for _ in range(3))
    p = rng.standard_normal(nv)
    ff = np.zeros(m.num_cells, dtype=np.uint16)
    np.bitwise_or.at(ff, m.facet_cells, (1 << m.facet_local).astype(np.uint16))
    Fg, Jg = G3.element_tensors(G3.P1_TET, m.x, m.cells.astype(np.int64), u, un, p, prm, ff, un2=un2)
    Fn, Jn = TN.element_tensors(m.x, m.cells.astype(np.int64), u, un, p, prm, ff.astype(np.uint8), un2=un2)
    assert np.abs(Fg - Fn).max() <= 1e-12 * np.abs(Fn).max()
    assert np.abs(Jg - Jn).max() <= 1e-12 * np.abs(Jn).max()


@pytest.mark.parametrize("kind", ["P2", "Q1"])
@pytest.mark.parametrize("kw", VARIANTS)
def test_jacobian_is_the_derivative_of_the_residual(kind, kw):
    rng = np.random.default_rng(2)
    m = node_mesh3(kind, 1, distort=0.1)
    nv = m.num_vertices
    prm = TN.Params(0.05, 1.3, 0.04, (0.2, -0.1, 0.3), **kw)
    pb = problem3(kind, m, prm)
    if kw.get("beta_backflow"):
        pb.set_boundary_terms(False, np.arange(0, m.num_facets, 2), 0.3)
    x0, un, un2 = 0.3 * rng.standard_normal(4 * nv), 0.3 * rng.standard_normal((nv, 3)), 0.3 * rng.standard_normal((nv, 3))
    _, J = pb.assemble(x0, un, apply_bc=False, un2=un2)
    eps = 1e-6
    for k in rng.choice(4 * nv, size=6, replace=False):
        e = np.zeros(4 * nv)
        e[k] = eps
        Fp, _ = pb.assemble(x0 + e, un, want_jac=False, apply_bc=False, un2=un2)
        Fm, _ = pb.assemble(x0 - e, un, want_jac=False, apply_bc=False, un2=un2)
        fd = (Fp - Fm) / (2 * eps)
        col = np.asarray(J[:, k].todense()).ravel()
        assert np.abs(fd - col).max() <= 2e-7 * max(np.abs(col).max(), 1.0)


@pytest.mark.parametrize("kind", ["P2", "Q1"])
def test_patch_tests_and_exact_integrals(kind):
    m = node_mesh3(kind, 2, distort=0.1)
    nv = m.num_vertices
    # uniform flow, steady, no force: every term vanishes (ds pair off: the pressure term of the ds pair is a boundary integral)
    prm = TN.Params(0.05, 1.3, 0.04, (0.0, 0.0, 0.0), ds_terms=False)
    pb = problem3(kind, m, prm)
    u0 = np.tile([0.3, -0.2, 0.5], (nv, 1))
    x = np.concatenate([u0.ravel(), np.zeros(nv)])
    F, _ = pb.assemble(x, u0, want_jac=False, apply_bc=False)
    assert np.abs(F).max() <= 1e-13
    # hydrostatic pressure p = rho f.x balances the force at rest in the interior rows (Galerkin + PSPG/SUPG residual R = 0)
    f = np.array([0.2, -0.1, 0.3])
    prm = TN.Params(0.05, 1.3, 0.04, tuple(f), ds_terms=False)
    pb = problem3(kind, m, prm)
    x = np.concatenate([np.zeros(3 * nv), 1.3 * (m.x @ f)])
    F, _ = pb.assemble(x, np.zeros((nv, 3)), want_jac=False, apply_bc=False)
    interior = np.setdiff1d(np.arange(nv), facet_node_set3(m, np.arange(m.num_facets)))
    if len(interior):
        rows = np.concatenate([3 * interior, 3 * interior + 1, 3 * interior + 2])
        assert np.abs(F[rows]).max() <= 1e-12
    # L2 norms of a trilinear / quadratic field and the flux of a linear field are integrated exactly
    vol = m.cell_volumes().sum()
    xq = np.concatenate([np.tile([1.0, 2.0, -2.0], nv), np.full(nv, 3.0)])
    nu_, np_ = pb.l2_norms(xq)
    assert abs(nu_ - 3.0 * np.sqrt(vol)) <= 1e-12 * nu_ and abs(np_ - 3.0 * np.sqrt(vol)) <= 1e-12 * np_
    ulin = np.stack([m.x[:, 0], 2.0 * m.x[:, 1], -0.5 * m.x[:, 2]], axis=1)       # div u = 2.5
    q = pb.flux(np.concatenate([ulin.ravel(), np.zeros(nv)]), np.arange(m.num_facets))
    assert abs(q - 2.5 * vol) <= 1e-12 * abs(q)


@pytest.mark.parametrize("kind", ["P1", "P2", "Q1"])
@pytest.mark.parametrize("kw", VARIANTS)
def test_c_restatement_equals_the_twin(kind, kw):
    """oracle/cfdh_oracle_gen3.c (scalar loops) against the einsum twin: element residuals and Jacobians, all three variants."""
    from gen3_util import ETYPE3
    from oracle import orcg3
    rng = np.random.default_rng(7)
    m = node_mesh3(kind, 2, distort=0.1)
    nv = m.num_vertices
    prm = TN.Params(0.02, 1.3, 0.04, (0.2, -0.1, 0.3), **kw)
    u, un, un2 = (0.3 * rng.standard_normal((nv, 3)) for _ in range(3))
    p = rng.standard_normal(nv)
    ff = np.zeros(m.num_cells, dtype=np.uint16)
    np.bitwise_or.at(ff, m.facet_cells, (1 << m.facet_local).astype(np.uint16))
    if kw.get("beta_backflow"):
        k = np.arange(0, m.num_facets, 2)
        np.bitwise_or.at(ff, m.facet_cells[k], (256 << m.facet_local[k]).astype(np.uint16))
    args = (ETYPE3[kind], m.x, m.cells.astype(np.int64), u, un, p, prm, ff)
    Ft, Jt = G3.element_tensors(*args, un2=un2)
    Fc, Jc = orcg3.element_tensors(*args, un2=un2)
    assert np.abs(Fc - Ft).max() <= 1e-13 * np.abs(Ft).max()
    assert np.abs(Jc - Jt).max() <= 1e-13 * np.abs(Jt).max()
    Fc0, J0 = orcg3.element_tensors(*args, want_jac=False, un2=un2)
    assert J0 is None and np.array_equal(Fc0, Fc)
