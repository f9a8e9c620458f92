"""P2/P2 triangles and Q1/Q1 quadrilaterals on the GPU (csrc/cfdh_gen.hip) against the checker (oracle/np_twin_gen.py with the
C element routine of oracle/cfdh_oracle_gen.c) -- SURVEY.md section 8f-4, in the pattern of tests/test_time_scheme.py:
assembly (residual, every Jacobian entry, Dirichlet rows / columns / lifting, ds pair, BDF2 coefficients, backflow term),
SpMV, functionals, time steps, and the two plugin routes: `stabilized_schur_backflow` with `p_grade=2`
(stabilized_schur_backflow.py:84-87) and `unit_square_pipe` on quadrilateral cells (unit_square_pipe.py:101-105)."""
import numpy as np
import pytest

from cfd_hemodynamic_amd import _lib
from gen_util import ETYPE, LIB_ETYPE, facet_node_set, node_mesh, problem, stenosis_nodes
from oracle import np_twin as T, np_twin_gen as G, orcg

pytestmark = pytest.mark.gpu
VARIANTS = [dict(), dict(theta=1.0, a0=1.5, a1=-2.0, a2=0.5), dict(ds_terms=False, beta_backflow=0.3)]


@pytest.fixture(autouse=True)
def _c_element_routine(monkeypatch):
    monkeypatch.setattr(G, "element_tensors", orcg.element_tensors)  # the twin assembles with the C restatement


def _ctx(kind, m, prm, markers=None):
    fm = np.zeros(m.num_facets, dtype=np.int32) if markers is None else markers
    ctx = _lib.Context(m.x, m.cells, m.facet_cells, m.facet_local, fm, etype=LIB_ETYPE[kind])
    ctx.set_params(prm.dt, prm.rho, prm.mu, mu_facet=prm.mu_facet, f=prm.f)
    ctx.set_time_scheme(prm.theta, prm.a0, prm.a1, prm.a2)
    return ctx


@pytest.mark.parametrize("kind", ["P1", "P2", "Q1"])
@pytest.mark.parametrize("kw", VARIANTS)
def test_assembly_matches_the_oracle(kind, kw):
    rng = np.random.default_rng(3)
    m = node_mesh(kind, 12, distort=0.05)
    nv = m.num_vertices
    prm = T.Params(0.02, 1.3, 0.04, (0.2, -0.1), **kw)
    pb = problem(kind, m, prm)
    markers = np.zeros(m.num_facets, dtype=np.int32)
    ctx = _ctx(kind, m, prm, markers)
    if kw.get("beta_backflow"):
        out = np.arange(0, m.num_facets, 2)
        markers[out] = 7
        pb.set_boundary_terms(False, out, 0.3)
        ctx.set_facet_markers(markers)
        ctx.set_boundary_terms(False, 7, 0.3)
    bnd = facet_node_set(m, np.arange(m.num_facets))[::2]
    vals = rng.standard_normal((len(bnd), 2))
    pn = facet_node_set(m, np.arange(m.num_facets))[1::5]
    for target in (pb, ctx):
        (target.add_bc_u if target is pb else lambda n, v: target.add_dirichlet(0, n, v))(bnd, vals)
        (target.add_bc_u if target is pb else lambda n, v: target.add_dirichlet(0, n, v))(bnd[:4], vals[:4])   # held twice: diagonal 2
        (target.add_bc_p if target is pb else lambda n, v: target.add_dirichlet(1, n, v))(pn, 0.5 * np.ones(len(pn)))
    xv, un, un2 = 0.3 * rng.standard_normal(3 * nv), 0.3 * rng.standard_normal((nv, 2)), 0.3 * rng.standard_normal((nv, 2))
    F, J = pb.assemble(xv, un, un2=un2)
    ctx.set_state(u_prev=un.ravel(), p_prev=np.zeros(nv), u=xv[: 2 * nv], p=xv[2 * nv:])
    ctx.set_previous2(un2.ravel())
    ctx.assemble(True)
    Fg = np.concatenate(ctx.get_residual())
    Jg = ctx.get_csr()
    assert np.abs(Fg - F).max() <= 1e-12 * np.abs(F).max()
    assert abs(Jg - J).max() <= 1e-12 * abs(J).max()
    y = rng.standard_normal(3 * nv)
    assert np.abs(ctx.spmv(y) - J @ y).max() <= 1e-12 * np.abs(J @ y).max()
    # residual-only pass (line-search trial points): the same residual
    ctx.assemble(False)
    assert np.abs(np.concatenate(ctx.get_residual()) - F).max() <= 1e-12 * np.abs(F).max()
    # functionals on the iterate
    nu_, np_ = pb.l2_norms(xv)
    assert abs(ctx.functional(2) - nu_) <= 1e-12 * nu_ and abs(ctx.functional(3) - np_) <= 1e-12 * np_
    if kw.get("beta_backflow"):
        q = pb.flux(xv, np.nonzero(markers == 7)[0])
        assert abs(ctx.functional(7, 7) - q) <= 1e-12 * max(abs(q), 1.0)
    ctx.close()


def test_generic_kernels_agree_with_the_closed_form_p1_kernels():
    """P1 triangles through the quadrature kernels of the P2 / Q1 path vs the production P1 kernels: same residual, same CSR."""
    rng = np.random.default_rng(4)
    m = node_mesh("P1", 20, distort=0.05)
    nv = m.num_vertices
    res = []
    for et in (0, 3):
        ctx = _lib.Context(m.x, m.cells, m.facet_cells, m.facet_local, np.zeros(m.num_facets, np.int32), etype=et)
        ctx.set_params(0.02, 1.3, 0.04, f=(0.2, -0.1))
        bnd = facet_node_set(m, np.arange(m.num_facets))
        rng2 = np.random.default_rng(5)
        ctx.add_dirichlet(0, bnd, rng2.standard_normal((len(bnd), 2)))
        st = np.random.default_rng(6)
        ctx.set_state(u_prev=0.3 * st.standard_normal(2 * nv), p_prev=np.zeros(nv), u=0.3 * st.standard_normal(2 * nv), p=st.standard_normal(nv))
        ctx.assemble(True)
        res.append((np.concatenate(ctx.get_residual()), ctx.get_csr(), ctx.info(28)))
        ctx.close()
    (F0, J0, e0), (F1, J1, e1) = res
    assert (e0, e1) == (0, 0)  # both are P1 contexts; the second ran the generic kernels
    assert np.abs(F0 - F1).max() <= 1e-12 * np.abs(F0).max() and abs(J0 - J1).max() <= 1e-12 * abs(J0).max()


@pytest.mark.parametrize("kind", ["P2", "Q1"])
def test_time_steps_match_the_twin(kind):
    """Three steps of a driven cavity-like problem (lid on top, no-slip elsewhere, singular pressure) and of a channel with
    p = 0 outlet: device Newton + FGMRES vs the twin's Newton with a direct solve."""
    m = node_mesh(kind, 8)
    nv = m.num_vertices
    prm = T.Params(0.02, 1.0, 0.02, (0.0, 0.0))
    top = np.nonzero(np.isclose(m.facet_midpoints()[:, 1], m.x[:, 1].max()))[0]
    left = np.nonzero(np.isclose(m.facet_midpoints()[:, 0], 0.0))[0]
    right = np.nonzero(np.isclose(m.facet_midpoints()[:, 0], m.x[:, 0].max()))[0]
    bottom = np.nonzero(np.isclose(m.facet_midpoints()[:, 1], 0.0))[0]
    for case in ("cavity", "channel"):
        pb = problem(kind, m, prm)
        ctx = _ctx(kind, m, prm)
        if case == "cavity":
            walls = facet_node_set(m, np.concatenate([left, right, bottom]))
            lid = np.setdiff1d(facet_node_set(m, top), walls)
            sets = [(0, walls, np.zeros((len(walls), 2))), (0, lid, np.tile([1.0, 0.0], (len(lid), 1)))]
        else:
            walls = facet_node_set(m, np.concatenate([top, bottom]))
            inl = np.setdiff1d(facet_node_set(m, left), walls)
            y = m.x[inl, 1] / m.x[:, 1].max()
            outn = facet_node_set(m, right)
            sets = [(0, walls, np.zeros((len(walls), 2))), (0, inl, np.stack([4 * y * (1 - y), 0 * y], 1)), (1, outn, np.zeros(len(outn)))]
        for fld, nodes, vals in sets:
            (pb.add_bc_u if fld == 0 else pb.add_bc_p)(nodes, vals)
            ctx.add_dirichlet(fld, nodes, vals)
        o = ctx.default_options()
        # (the singular cavity system cannot be driven below ~1e-9 relative: a rounding-level part of the right-hand side lies
        # outside the range of the Jacobian, as for the P1 cavity -- DESIGN.md section 6)
        o.snes_rtol, o.snes_stol, o.ksp_rtol = (1e-10, 0.0, 1e-8) if case == "cavity" else (1e-12, 0.0, 1e-10)
        tol_u, tol_p = (1e-7, 1e-6) if case == "cavity" else (1e-8, 1e-7)
        ctx.set_options(o)
        z2, z1 = np.zeros(2 * nv), np.zeros(nv)
        ctx.set_state(u_prev=z2, p_prev=z1, u=z2, p=z1)
        x, un = np.zeros(3 * nv), np.zeros((nv, 2))
        for step in range(3):
            st = ctx.solve_step()
            assert st.reason > 0
            u, p = ctx.get_solution()
            ctx.advance()
            if case == "cavity":
                x[2 * nv:] -= x[2 * nv:].mean()
            x, _ = pb.newton(x, un)
            un = x[: 2 * nv].reshape(-1, 2).copy()
            pt = x[2 * nv:] - (x[2 * nv:].mean() if case == "cavity" else 0.0)
            pg = p - (p.mean() if case == "cavity" else 0.0)
            assert np.abs(u - x[: 2 * nv]).max() <= tol_u * np.abs(x[: 2 * nv]).max(), (kind, case, step)
            assert np.abs(pg - pt).max() <= tol_p * np.abs(pt).max(), (kind, case, step)
        ctx.close()


def test_backflow_plugin_with_p_grade_2_on_the_stenosis():
    """`--simulation stenosis --solver stabilized_schur_backflow --p_grade 2` (stabilized_schur_backflow.py:63,84-87): P2/P2 on the
    stenosed channel, do-nothing outlet with backflow stabilisation; the scenario loop against the twin."""
    from cfd_hemodynamic_amd.scenarios.stenosis import StenosisSimulation
    kw = dict(ny=6, L=12.0, x_sten=5.0, v_max=60.0, quiet=True, beta_backflow=0.2, p_grade=2,
              options=dict(snes_rtol=1e-12, snes_stol=0.0, ksp_rtol=1e-10))
    sc = StenosisSimulation("stabilized_schur_backflow", 0.01, 0.025, **kw)
    dm = sc.solver.V.mesh
    nv = dm.num_vertices
    assert sc.solver.p_grade == 2 and sc.solver.ctx.info(28) == 1 and nv > sc.mesh.num_vertices
    assert sc.solver.V.dofmap.index_map.size_global == nv and sc.solver.V.dofmap.index_map_bs == 2
    u0 = np.array(sc.solver.u_prev.x.array, dtype=float)
    sc.solve(None)
    assert sc.num_steps == 3
    m2, ft = stenosis_nodes("P2", 6, 12.0, 5.0)
    assert np.array_equal(m2.x, dm.x)
    prm = T.Params(0.01, 1.06e-3, 3.5e-3, (0.0, 0.0), ds_terms=False, beta_backflow=0.2)
    pb = problem("P2", m2, prm)
    pb.set_boundary_terms(False, ft.find(3), 0.2)
    wn = facet_node_set(m2, ft.find(4))
    inn = facet_node_set(m2, ft.find(2))
    pb.add_bc_u(wn, np.zeros((len(wn), 2)))
    y = m2.x[inn, 1]
    pb.add_bc_u(inn, np.stack([60.0 * (1.0 - ((y - 1.57) / 1.57) ** 2), 0 * y], 1))
    x = np.concatenate([u0, np.zeros(nv)])
    un = u0.reshape(-1, 2).copy()
    for _ in range(3):
        x, _ = pb.newton(x, un)
        un = x[: 2 * nv].reshape(-1, 2).copy()
    xg = np.concatenate([np.asarray(sc.solver.u_sol.x.array), np.asarray(sc.solver.p_sol.x.array)])
    assert np.abs(xg - x).max() <= 1e-8 * np.abs(x).max()
    nu_, np_ = pb.l2_norms(x)
    assert abs(sc.norm_v - nu_) <= 1e-9 * nu_ and abs(sc.norm_p - np_) <= 1e-8 * np_


def test_unit_square_pipe_on_quadrilaterals(tmp_path):
    """unit_square_pipe.py: pressure-driven channel on quadrilateral cells with `--solver stabilized_schur` (Q1/Q1).  A short
    channel here; the scenario's constants and the full-size mesh shape are checked without running it."""
    from cfd_hemodynamic_amd.io import read_vtu
    from cfd_hemodynamic_amd.scenarios.unit_square_pipe import UnitSquarePipeSimulation
    sc = UnitSquarePipeSimulation("stabilized_schur", 0.01, 0.035, p_inlet=7.47, p_outlet=0.0, nx=40, ny=6, L=6.0, quiet=True,
                                  options=dict(snes_rtol=1e-12, snes_stol=0.0, ksp_rtol=1e-10))
    m = sc.mesh
    assert m.topology.cell_name() == "quadrilateral" and sc.solver.ctx.info(28) == 2 and sc.solver.ctx.info(29) == 4
    assert (len(sc._ft.find(1)), len(sc._ft.find(2)), len(sc._ft.find(3))) == (6, 6, 80)
    out = tmp_path / "run"
    sc.solve(str(out))
    nv = m.num_vertices
    prm = T.Params(0.01, 1.06e-3, 3.5e-3, (0.0, 0.0))
    pb = problem("Q1", m, prm)
    wn = facet_node_set(m, sc._ft.find(3))
    pb.add_bc_u(wn, np.zeros((len(wn), 2)))
    for mk, val in ((1, 7.47), (2, 0.0)):
        n_ = facet_node_set(m, sc._ft.find(mk))
        pb.add_bc_p(n_, val * np.ones(len(n_)))
    x, un = np.zeros(3 * nv), np.zeros((nv, 2))
    for _ in range(sc.num_steps):
        x, _ = pb.newton(x, un)
        un = x[: 2 * nv].reshape(-1, 2).copy()
    xg = np.concatenate([np.asarray(sc.solver.u_sol.x.array), np.asarray(sc.solver.p_sol.x.array)])
    assert sc.num_steps == 4 and np.abs(xg - x).max() <= 1e-8 * np.abs(x).max()
    # flow goes down the pressure gradient; the VTU series holds quadrilateral cells
    assert sc.solver.functional(7, 2) > 0 > sc.solver.functional(7, 1)
    last = read_vtu(str(out / ("v_%06d.vtu" % sc.num_steps)))
    assert np.allclose(last["v"][:, :2].ravel(), np.asarray(sc.solver.u_sol.x.array), rtol=0, atol=1e-14)
    # the reference's constants: 587 x 11 cells of 80 x 1.5 mm -> 7 056 nodes, 21 168 DOF
    from cfd_hemodynamic_amd.scenarios import unit_square_pipe as usp
    full = usp.create_rectangle((0.0, 0.0), (usp._L, usp._H), (usp._NX, usp._NY))
    assert (usp._L, usp._H, usp._NX, usp._NY) == (80.0, 1.5, 587, 11) and full.num_vertices == 7056


@pytest.mark.parametrize("kind", ["P2", "Q1"])
def test_generic_assembly_is_bitwise_reproducible(kind):
    """Round 4: the element blocks are staged by destination and summed in a fixed order (no atomics), so two passes -- and two
    contexts -- give identical bits in the residual and in every CSR value."""
    rng = np.random.default_rng(11)
    m = node_mesh(kind, 40, distort=0.05)
    nv = m.num_vertices
    prm = T.Params(0.02, 1.3, 0.04, (0.2, -0.1))
    bnd = facet_node_set(m, np.arange(m.num_facets))[::2]
    vals = rng.standard_normal((len(bnd), 2))
    xv, un = 0.3 * rng.standard_normal(3 * nv), 0.3 * rng.standard_normal((nv, 2))
    out = []
    for _ in range(2):
        ctx = _ctx(kind, m, prm)
        ctx.add_dirichlet(0, bnd, vals)
        ctx.set_state(u_prev=un.ravel(), p_prev=np.zeros(nv), u=xv[: 2 * nv], p=xv[2 * nv:])
        for _rep in range(2):
            ctx.assemble(True)
            out.append((np.concatenate(ctx.get_residual()), ctx.get_csr().data.copy()))
        ctx.close()
    for F, A in out[1:]:
        assert np.array_equal(F, out[0][0]) and np.array_equal(A, out[0][1])


@pytest.mark.parametrize("kind,n", [("P2", 92), ("Q1", 184)])
def test_time_steps_match_the_twin_at_size(kind, n):
    """At-size step parity for the SURVEY 8f-4 elements (missing 6): >= 100 k DOF, two steps of the channel
    problem (parabolic inlet, no-slip walls, p = 0 outlet) from rest, device Newton + FGMRES + Cahouet-Chabard/AMG at tight
    tolerances against the twin's Newton with a DIRECT sparse solve (C element routine)."""
    m = node_mesh(kind, n)
    nv = m.num_vertices
    assert 3 * nv >= 100000
    prm = T.Params(0.02, 1.0, 0.02, (0.0, 0.0))
    mid = m.facet_midpoints()
    top, bottom = np.nonzero(np.isclose(mid[:, 1], m.x[:, 1].max()))[0], np.nonzero(np.isclose(mid[:, 1], 0.0))[0]
    left, right = np.nonzero(np.isclose(mid[:, 0], 0.0))[0], np.nonzero(np.isclose(mid[:, 0], m.x[:, 0].max()))[0]
    walls = facet_node_set(m, np.concatenate([top, bottom]))
    inl = np.setdiff1d(facet_node_set(m, left), walls)
    y = m.x[inl, 1] / m.x[:, 1].max()
    outn = facet_node_set(m, right)
    pb = problem(kind, m, prm)
    ctx = _ctx(kind, m, prm)
    for fld, nodes, vals in [(0, walls, np.zeros((len(walls), 2))), (0, inl, np.stack([4 * y * (1 - y), 0 * y], 1)), (1, outn, np.zeros(len(outn)))]:
        (pb.add_bc_u if fld == 0 else pb.add_bc_p)(nodes, vals)
        ctx.add_dirichlet(fld, nodes, vals)
    o = ctx.default_options()
    o.snes_rtol, o.snes_stol, o.ksp_rtol = 1e-12, 0.0, 1e-10
    ctx.set_options(o)
    z2, z1 = np.zeros(2 * nv), np.zeros(nv)
    ctx.set_state(u_prev=z2, p_prev=z1, u=z2, p=z1)
    x, un = np.zeros(3 * nv), np.zeros((nv, 2))
    for step in range(2):
        st = ctx.solve_step()
        assert st.reason > 0
        u, p = ctx.get_solution()
        ctx.advance()
        x, _ = pb.newton(x, un)
        un = x[: 2 * nv].reshape(-1, 2).copy()
        assert np.abs(u - x[: 2 * nv]).max() <= 1e-8 * np.abs(x[: 2 * nv]).max(), (kind, step)
        assert np.abs(p - x[2 * nv:]).max() <= 1e-7 * np.abs(x[2 * nv:]).max(), (kind, step)
    nu_, np_ = pb.l2_norms(x)
    assert abs(ctx.functional(2) - nu_) <= 1e-9 * nu_ and abs(ctx.functional(3) - np_) <= 1e-8 * np_
    ctx.close()
