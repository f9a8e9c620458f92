"""Pins of the P2 / Q1 checker (oracle/np_twin_gen.py, oracle/cfdh_oracle_gen.c) -- SURVEY.md section 8f-4.  The reference
holds no vector for this path (parity unpinned); what holds the restatement in place:
  * on P1 triangles the quadrature-based twin equals the closed-form twin (np_twin.py) to round-off -- residual, Jacobian,
    ds pair, BDF2 coefficients, backflow term, Dirichlet handling;
  * the Jacobian is the derivative of the residual (central differences) for P2 and Q1;
  * patch tests: uniform flow, hydrostatic pressure, exact L2 norms, Poiseuille flow (exact for P2);
  * the C restatement equals the twin to 1e-13."""
import numpy as np
import pytest

from gen_util import ETYPE, facet_node_set, node_mesh, problem
from oracle import np_twin as T, np_twin_gen as G, orcg

VARIANTS = [dict(), dict(theta=1.0, a0=1.5, a1=-2.0, a2=0.5), dict(ds_terms=False, beta_backflow=0.3)]


def _state(nv, rng, uniform=False):
    un = np.tile([0.7, -0.4], (nv, 1)) + 0.05 * rng.standard_normal((nv, 2)) if uniform else 0.3 * rng.standard_normal((nv, 2))
    return 0.3 * rng.standard_normal(3 * nv), un, 0.3 * rng.standard_normal((nv, 2))


@pytest.mark.parametrize("kw", VARIANTS)
def test_generic_twin_equals_closed_form_twin_on_p1(kw):
    rng = np.random.default_rng(0)
    m = node_mesh("P1", 5, distort=0.05)
    nv = m.num_vertices
    prm = T.Params(0.02, 1.3, 0.04, (0.2, -0.1), **kw)
    p0 = T.Problem(m.x, m.cells, m.facet_cells, m.facet_local, prm)
    p1 = problem("P1", m, prm)
    if kw.get("beta_backflow"):
        out = np.arange(0, m.num_facets, 3)
        p0.set_boundary_terms(False, out, 0.3)
        p1.set_boundary_terms(False, out, 0.3)
    xv, un, un2 = _state(nv, rng, uniform=bool(kw.get("beta_backflow")))  # no sign change of u_n.n: both facet rules are exact
    bnd = facet_node_set(m, np.arange(m.num_facets))[::2]
    vals = rng.standard_normal((len(bnd), 2))
    for pb in (p0, p1):
        pb.add_bc_u(bnd, vals)
        pb.add_bc_u(bnd[:3], vals[:3])  # a dof held by two objects: diagonal 2
    F0, J0 = p0.assemble(xv, un, un2=un2)
    F1, J1 = p1.assemble(xv, un, un2=un2)
    assert np.abs(F0 - F1).max() <= 1e-13 * np.abs(F0).max()
    assert abs(J0 - J1).max() <= 1e-13 * abs(J0).max()
    assert np.allclose(p0.l2_norms(xv), p1.l2_norms(xv), rtol=1e-13)


@pytest.mark.parametrize("kind", ["P2", "Q1"])
@pytest.mark.parametrize("kw", VARIANTS)
def test_jacobian_is_the_derivative_of_the_residual(kind, kw):
    rng = np.random.default_rng(1)
    m = node_mesh(kind, 3, distort=0.1)
    nv = m.num_vertices
    prm = T.Params(0.02, 1.3, 0.04, (0.2, -0.1), **kw)
    pb = problem(kind, m, prm)
    if kw.get("beta_backflow"):
        pb.set_boundary_terms(False, np.arange(0, m.num_facets, 2), 0.3)
    xv, un, un2 = _state(nv, rng)
    _, J = pb.assemble(xv, un, apply_bc=False, un2=un2)
    d, eps = rng.standard_normal(3 * nv), 1e-6
    Fp, _ = pb.assemble(xv + eps * d, un, want_jac=False, apply_bc=False, un2=un2)
    Fm, _ = pb.assemble(xv - eps * d, un, want_jac=False, apply_bc=False, un2=un2)
    fd = (Fp - Fm) / (2 * eps)
    assert np.abs(J @ d - fd).max() <= 2e-8 * np.abs(fd).max()


@pytest.mark.parametrize("kind", ["P2", "Q1"])
def test_patch_tests(kind):
    m = node_mesh(kind, 4, distort=0.1)
    nv = m.num_vertices
    bn = facet_node_set(m, np.arange(m.num_facets))
    inter = np.setdiff1d(np.arange(nv), bn)
    # uniform flow, constant pressure, no force: the strong residual vanishes, so do SUPG / PSPG; interior rows are zero
    pb = problem(kind, m, T.Params(0.02, 1.3, 0.04, (0.0, 0.0)))
    u0 = np.tile([0.7, -0.4], (nv, 1))
    F, _ = pb.assemble(np.concatenate([u0.ravel(), 2.0 * np.ones(nv)]), u0, want_jac=False, apply_bc=False)
    assert np.abs(F[: 2 * nv].reshape(-1, 2)[inter]).max() < 1e-13 and np.abs(F[2 * nv:][inter]).max() < 1e-13
    # hydrostatic: u = 0, grad p = rho f -- every row vanishes (the ds pair supplies the boundary term)
    f = np.array([0.3, -0.2])
    pb = problem(kind, m, T.Params(0.02, 1.3, 0.04, f))
    F, _ = pb.assemble(np.concatenate([np.zeros(2 * nv), 1.3 * (m.x @ f)]), np.zeros((nv, 2)), want_jac=False, apply_bc=False)
    assert np.abs(F).max() < 1e-13
    # L2 norm of u = (x, y) on the (sheared) domain against a fine Riemann sum is too crude: use the exact mass instead
    nu, npr = pb.l2_norms(np.concatenate([np.ones(2 * nv), np.ones(nv)]))
    area = 1.0 if kind == "P2" else 0.8
    assert abs(nu - np.sqrt(2 * area)) < 1e-12 and abs(npr - np.sqrt(area)) < 1e-12


def test_poiseuille_is_reproduced_exactly_by_p2():
    """Steady plane Poiseuille flow u = (4 y (1 - y), 0), p = -8 mu x (unit_square.py:100-104) lies in the P2 space: with that
    state as u and u_prev, Dirichlet velocity on the whole boundary, every residual row vanishes (convection u.grad u = 0,
    viscous part of the strong residual cancels grad p) -- P1 only approximates it."""
    m = node_mesh("P2", 4)
    nv = m.num_vertices
    mu = 0.05
    pb = problem("P2", m, T.Params(0.02, 1.3, mu, (0.0, 0.0)))
    u = np.stack([4 * m.x[:, 1] * (1 - m.x[:, 1]), 0 * m.x[:, 0]], axis=1)
    bn = facet_node_set(m, np.arange(m.num_facets))
    pb.add_bc_u(bn, u[bn])
    x = np.concatenate([u.ravel(), -8.0 * mu * m.x[:, 0]])
    F, _ = pb.assemble(x, u)
    assert np.abs(F).max() < 1e-12
    xs, hist = pb.newton(np.zeros(3 * nv), u, rtol=1e-13)
    d = xs - x
    d[2 * nv:] -= d[2 * nv:].mean()
    assert np.abs(d).max() < 1e-9, hist


@pytest.mark.parametrize("kind", ["P1", "P2", "Q1"])
@pytest.mark.parametrize("kw", VARIANTS)
def test_c_oracle_equals_twin(kind, kw):
    rng = np.random.default_rng(2)
    m = node_mesh(kind, 3, distort=0.1)
    nv = m.num_vertices
    prm = T.Params(0.02, 1.3, 0.04, (0.2, -0.1), **kw)
    pb = problem(kind, m, prm)
    if kw.get("beta_backflow"):
        pb.set_boundary_terms(False, np.arange(0, m.num_facets, 2), 0.3)
    u, un, un2 = 0.3 * rng.standard_normal((nv, 2)), 0.3 * rng.standard_normal((nv, 2)), 0.3 * rng.standard_normal((nv, 2))
    p = rng.standard_normal(nv)
    Fe, Je = G.element_tensors(ETYPE[kind], m.x, m.cells, u, un, p, prm, pb.facet_flags, True, un2)
    Fc, Jc = orcg.element_tensors(ETYPE[kind], m.x, m.cells, u, un, p, prm, pb.facet_flags, True, un2)
    assert np.abs(Fe - Fc).max() <= 1e-13 * np.abs(Fe).max() and np.abs(Je - Jc).max() <= 1e-13 * np.abs(Je).max()
    Fc2, none = orcg.element_tensors(ETYPE[kind], m.x, m.cells, u, un, p, prm, pb.facet_flags, False, un2)
    assert none is None and np.array_equal(Fc, Fc2)
