"""Q1/Q1 hexahedra and P2/P2 tetrahedra on the GPU (csrc/cfdh_gen3.hip) against the checker (oracle/np_twin_gen3.py with the C
element routine of oracle/cfdh_oracle_gen3.c) -- SURVEY.md section 8f-4, 3-D half, in the pattern of tests/test_gpu_gen.py:
assembly (residual, every Jacobian entry, Dirichlet rows / columns / lifting, ds pair, BDF2 coefficients, backflow term), SpMV,
functionals, bitwise reproducibility, time steps, and the plugin routes: `unit_cube_pipe` on hexahedral cells
(unit_cube_pipe.py:103-109) and `stabilized_schur_backflow` with `p_grade=2` on tetrahedra (stabilized_schur_backflow.py:84-87)."""
import numpy as np
import pytest

from cfd_hemodynamic_amd import _lib
from gen3_util import ETYPE3, LIB_ETYPE3, facet_node_set3, node_mesh3, problem3
from oracle import np_twin_gen3 as G3, np_twin_nd as TN, orcg3

pytestmark = pytest.mark.gpu
VARIANTS = [dict(), dict(theta=1.0, a0=1.5, a1=-2.0, a2=0.5), dict(ds_terms=False, beta_backflow=0.3)]


@pytest.fixture(autouse=True)
def _c_element_routine(monkeypatch):
    monkeypatch.setattr(G3, "element_tensors", orcg3.element_tensors)  # the twin assembles with the C restatement


def _ctx(kind, m, prm, markers=None):
    fm = np.zeros(m.num_facets, dtype=np.int32) if markers is None else markers
    ctx = _lib.Context(m.x, m.cells, m.facet_cells, m.facet_local, fm, etype=LIB_ETYPE3[kind])
    assert ctx.dim == 3 and ctx.info(26) == 3
    ctx.set_params(prm.dt, prm.rho, prm.mu, mu_facet=prm.mu_facet, f=prm.f)
    ctx.set_time_scheme(prm.theta, prm.a0, prm.a1, prm.a2)
    return ctx


@pytest.mark.parametrize("kind", ["P1", "P2", "Q1"])
@pytest.mark.parametrize("kw", VARIANTS)
def test_assembly_matches_the_oracle(kind, kw):
    rng = np.random.default_rng(3)
    m = node_mesh3(kind, 3, distort=0.05)
    nv = m.num_vertices
    prm = TN.Params(0.02, 1.3, 0.04, (0.2, -0.1, 0.3), **kw)
    pb = problem3(kind, m, prm)
    markers = np.zeros(m.num_facets, dtype=np.int32)
    ctx = _ctx(kind, m, prm, markers)
    if kw.get("beta_backflow"):
        out = np.arange(0, m.num_facets, 2)
        markers[out] = 7
        pb.set_boundary_terms(False, out, 0.3)
        ctx.set_facet_markers(markers)
        ctx.set_boundary_terms(False, 7, 0.3)
    bnd = facet_node_set3(m, np.arange(m.num_facets))[::2]
    vals = rng.standard_normal((len(bnd), 3))
    pn = facet_node_set3(m, np.arange(m.num_facets))[1::5]
    for target in (pb, ctx):
        (target.add_bc_u if target is pb else lambda n, v: target.add_dirichlet(0, n, v))(bnd, vals)
        (target.add_bc_u if target is pb else lambda n, v: target.add_dirichlet(0, n, v))(bnd[:4], vals[:4])   # held twice: diagonal 2
        (target.add_bc_p if target is pb else lambda n, v: target.add_dirichlet(1, n, v))(pn, 0.5 * np.ones(len(pn)))
    xv, un, un2 = 0.3 * rng.standard_normal(4 * nv), 0.3 * rng.standard_normal((nv, 3)), 0.3 * rng.standard_normal((nv, 3))
    F, J = pb.assemble(xv, un, un2=un2)
    ctx.set_state(u_prev=un.ravel(), p_prev=np.zeros(nv), u=xv[: 3 * nv], p=xv[3 * nv:])
    ctx.set_previous2(un2.ravel())
    ctx.assemble(True)
    Fg = np.concatenate(ctx.get_residual())
    Jg = ctx.get_csr()
    assert np.abs(Fg - F).max() <= 1e-12 * np.abs(F).max()
    assert abs(Jg - J).max() <= 1e-12 * abs(J).max()
    y = rng.standard_normal(4 * nv)
    assert np.abs(ctx.spmv(y) - J @ y).max() <= 1e-12 * np.abs(J @ y).max()
    ctx.assemble(False)  # residual-only pass (line-search trial points): the same residual
    assert np.abs(np.concatenate(ctx.get_residual()) - F).max() <= 1e-12 * np.abs(F).max()
    nu_, np_ = pb.l2_norms(xv)
    assert abs(ctx.functional(2) - nu_) <= 1e-12 * nu_ and abs(ctx.functional(3) - np_) <= 1e-12 * np_
    if kw.get("beta_backflow"):
        q = pb.flux(xv, np.nonzero(markers == 7)[0])
        assert abs(ctx.functional(7, 7) - q) <= 1e-12 * max(abs(q), 1.0)
    ctx.close()


def test_generic_kernels_agree_with_the_closed_form_tetrahedral_kernels():
    """P1 tetrahedra through the quadrature kernels vs the production closed-form kernels: same residual, same CSR values."""
    rng = np.random.default_rng(4)
    m = node_mesh3("P1", 4, distort=0.05)
    nv = m.num_vertices
    res = []
    for et in (0, 3):
        ctx = _lib.Context(m.x, m.cells, m.facet_cells, m.facet_local, np.zeros(m.num_facets, np.int32), etype=et)
        ctx.set_params(0.02, 1.3, 0.04, f=(0.2, -0.1, 0.3))
        bnd = facet_node_set3(m, np.arange(m.num_facets))
        ctx.add_dirichlet(0, bnd, np.random.default_rng(5).standard_normal((len(bnd), 3)))
        st = np.random.default_rng(6)
        ctx.set_state(u_prev=0.3 * st.standard_normal(3 * nv), p_prev=np.zeros(nv), u=0.3 * st.standard_normal(3 * nv), p=st.standard_normal(nv))
        ctx.assemble(True)
        res.append((np.concatenate(ctx.get_residual()), ctx.get_csr()))
        ctx.close()
    (F0, J0), (F1, J1) = res
    assert np.abs(F0 - F1).max() <= 1e-12 * np.abs(F0).max() and abs(J0 - J1).max() <= 1e-12 * abs(J0).max()


@pytest.mark.parametrize("kind", ["P2", "Q1"])
def test_generic_3d_assembly_is_bitwise_reproducible(kind):
    rng = np.random.default_rng(11)
    m = node_mesh3(kind, 5, distort=0.05)
    nv = m.num_vertices
    prm = TN.Params(0.02, 1.3, 0.04, (0.2, -0.1, 0.3))
    bnd = facet_node_set3(m, np.arange(m.num_facets))[::2]
    vals = rng.standard_normal((len(bnd), 3))
    xv, un = 0.3 * rng.standard_normal(4 * nv), 0.3 * rng.standard_normal((nv, 3))
    out = []
    for _ in range(2):
        ctx = _ctx(kind, m, prm)
        ctx.add_dirichlet(0, bnd, vals)
        ctx.set_state(u_prev=un.ravel(), p_prev=np.zeros(nv), u=xv[: 3 * nv], p=xv[3 * nv:])
        for _rep in range(2):
            ctx.assemble(True)
            out.append((np.concatenate(ctx.get_residual()), ctx.get_csr().data.copy()))
        ctx.close()
    for F, A in out[1:]:
        assert np.array_equal(F, out[0][0]) and np.array_equal(A, out[0][1])


@pytest.mark.parametrize("kind,n", [("P2", 3), ("Q1", 5)])
def test_time_steps_match_the_twin(kind, n):
    """Two steps of a duct flow (parabolic-like inlet on x = 0, no-slip side walls, p = 0 on x = max) and of a lid-driven box
    (singular pressure): device Newton + FGMRES + Cahouet-Chabard/AMG vs the twin's Newton with a direct solve."""
    m = node_mesh3(kind, n)
    nv = m.num_vertices
    prm = TN.Params(0.02, 1.0, 0.02, (0.0, 0.0, 0.0))
    mid = m.facet_midpoints()
    hi = m.x.max(axis=0)
    on = lambda d, v: np.nonzero(np.isclose(mid[:, d], v))[0]  # noqa: E731
    for case in ("duct", "lid"):
        pb = problem3(kind, m, prm)
        ctx = _ctx(kind, m, prm)
        if case == "duct":
            walls = facet_node_set3(m, np.concatenate([on(1, 0.0), on(1, hi[1]), on(2, 0.0), on(2, hi[2])]))
            inl = np.setdiff1d(facet_node_set3(m, on(0, 0.0)), walls)
            y, z = m.x[inl, 1] / hi[1], m.x[inl, 2] / hi[2]
            outn = facet_node_set3(m, on(0, hi[0]))
            sets = [(0, walls, np.zeros((len(walls), 3))), (0, inl, np.stack([16 * y * (1 - y) * z * (1 - z), 0 * y, 0 * y], 1)), (1, outn, np.zeros(len(outn)))]
        else:
            walls = facet_node_set3(m, np.concatenate([on(0, 0.0), on(0, hi[0]), on(1, 0.0), on(1, hi[1]), on(2, 0.0)]))
            lid = np.setdiff1d(facet_node_set3(m, on(2, hi[2])), walls)
            sets = [(0, walls, np.zeros((len(walls), 3))), (0, lid, np.tile([1.0, 0.0, 0.0], (len(lid), 1)))]
        for fld, nodes, vals in sets:
            (pb.add_bc_u if fld == 0 else pb.add_bc_p)(nodes, vals)
            ctx.add_dirichlet(fld, nodes, vals)
        o = ctx.default_options()
        o.snes_rtol, o.snes_stol, o.ksp_rtol = (1e-10, 0.0, 1e-8) if case == "lid" else (1e-12, 0.0, 1e-10)
        tol_u, tol_p = (1e-7, 1e-6) if case == "lid" else (1e-8, 1e-7)
        ctx.set_options(o)
        z3, z1 = np.zeros(3 * nv), np.zeros(nv)
        ctx.set_state(u_prev=z3, p_prev=z1, u=z3, p=z1)
        x, un = np.zeros(4 * nv), np.zeros((nv, 3))
        for step in range(2):
            st = ctx.solve_step()
            assert st.reason > 0
            u, p = ctx.get_solution()
            ctx.advance()
            if case == "lid":
                x[3 * nv:] -= x[3 * nv:].mean()
            x, _ = pb.newton(x, un)
            un = x[: 3 * nv].reshape(-1, 3).copy()
            pt = x[3 * nv:] - (x[3 * nv:].mean() if case == "lid" else 0.0)
            pg = p - (p.mean() if case == "lid" else 0.0)
            assert np.abs(u - x[: 3 * nv]).max() <= tol_u * np.abs(x[: 3 * nv]).max(), (kind, case, step)
            assert np.abs(pg - pt).max() <= tol_p * np.abs(pt).max(), (kind, case, step)
        ctx.close()


def test_unit_cube_pipe_on_hexahedra(tmp_path):
    """unit_cube_pipe.py: pressure-driven duct on hexahedral cells with `--solver stabilized_schur` (Q1/Q1).  A short duct here;
    the scenario's constants and the full-size mesh shape are checked without running it."""
    from cfd_hemodynamic_amd.io import read_vtu
    from cfd_hemodynamic_amd.scenarios.unit_cube_pipe import UnitCubePipeSimulation
    sc = UnitCubePipeSimulation("stabilized_schur", 0.01, 0.025, p_inlet=8.85, p_outlet=0.0, nx=12, ny=3, nz=3, L=6.0, quiet=True,
                                options=dict(snes_rtol=1e-12, snes_stol=0.0, ksp_rtol=1e-10))
    m = sc.mesh
    assert m.topology.cell_name() == "hexahedron" and sc.solver.ctx.info(28) == 2 and sc.solver.ctx.info(29) == 8 and sc.solver.ctx.info(26) == 3
    assert (len(sc._ft.find(1)), len(sc._ft.find(2)), len(sc._ft.find(3))) == (9, 9, 4 * 36)
    out = tmp_path / "run"
    sc.solve(str(out))
    nv = m.num_vertices
    prm = TN.Params(0.01, 1.06e-3, 3.5e-3, (0.0, 0.0, 0.0))
    pb = problem3("Q1", m, prm)
    wn = facet_node_set3(m, sc._ft.find(3))
    pb.add_bc_u(wn, np.zeros((len(wn), 3)))
    for mk, val in ((1, 8.85), (2, 0.0)):
        n_ = facet_node_set3(m, sc._ft.find(mk))
        pb.add_bc_p(n_, val * np.ones(len(n_)))
    x, un = np.zeros(4 * nv), np.zeros((nv, 3))
    for _ in range(sc.num_steps):
        x, _ = pb.newton(x, un)
        un = x[: 3 * nv].reshape(-1, 3).copy()
    xg = np.concatenate([np.asarray(sc.solver.u_sol.x.array), np.asarray(sc.solver.p_sol.x.array)])
    assert sc.num_steps == 3 and np.abs(xg - x).max() <= 1e-8 * np.abs(x).max()
    assert sc.solver.functional(7, 2) > 0 > sc.solver.functional(7, 1)   # flow goes down the pressure gradient
    last = read_vtu(str(out / ("v_%06d.vtu" % sc.num_steps)))
    assert np.allclose(last["v"].ravel(), np.asarray(sc.solver.u_sol.x.array), rtol=0, atol=1e-14)
    # the reference's constants: 213 x 4 x 4 cells of an 80 x 1.5 x 1.5 mm duct -> 5 350 nodes, 21 400 DOF
    from cfd_hemodynamic_amd.scenarios import unit_cube_pipe as ucp
    full = ucp.create_box((0.0, 0.0, 0.0), (ucp._L, ucp._W, ucp._H), (ucp._NX, ucp._NY, ucp._NZ))
    assert (ucp._L, ucp._W, ucp._H, ucp._NX, ucp._NY, ucp._NZ) == (80.0, 1.5, 1.5, 213, 4, 4) and full.num_vertices == 5350


def test_backflow_plugin_with_p_grade_2_on_tetrahedra():
    """`stabilized_schur_backflow` with `p_grade = 2` on a 3-D mesh (stabilized_schur_backflow.py:63,84-87 with the tetrahedral
    meshes of scenario_factory.py:47-49): the plugin class itself -- spaces of degree 2 on tetrahedra, do-nothing outlet with the
    backflow term on tags["outlet"], no pressure condition -- on a duct of Kuhn tetrahedra, two steps against the twin."""
    from cfd_hemodynamic_amd.boundaryCondition import BoundaryCondition
    from cfd_hemodynamic_amd.fem import Function
    from cfd_hemodynamic_amd.mesh import locate_entities_boundary, meshtags
    from cfd_hemodynamic_amd.mesh3d import create_unit_cube
    from cfd_hemodynamic_amd.solvers.stabilized_schur_backflow import Solver
    mesh = create_unit_cube(3)
    inlet = locate_entities_boundary(mesh, 2, lambda x: np.isclose(x[0], 0.0))
    outlet = locate_entities_boundary(mesh, 2, lambda x: np.isclose(x[0], 1.0))
    # plane by plane: a corner triangle of the inlet face has all its vertices on SOME wall plane, but lies in none of them
    wall = np.unique(np.concatenate([locate_entities_boundary(mesh, 2, lambda x, d=d, v=v: np.isclose(x[d], v)) for d in (1, 2) for v in (0.0, 1.0)]))
    idx = np.concatenate([inlet, outlet, wall])
    val = np.concatenate([np.full(len(inlet), 1), np.full(len(outlet), 2), np.full(len(wall), 3)]).astype(np.int32)
    order = np.argsort(idx)
    ft = meshtags(mesh, 2, idx[order], val[order])
    s = Solver(mesh, 0.02, 1.0, 0.02, [0.0, 0.0, 0.0], v_max=1.0, p_grade=2, beta_backflow=0.2, quiet=True,
               options=dict(snes_rtol=1e-12, snes_stol=0.0, ksp_rtol=1e-10))
    dm = s.V.mesh
    nv = dm.num_vertices
    assert s.p_grade == 2 and s.ctx.info(28) == 1 and s.ctx.info(29) == 10 and s.ctx.info(26) == 3 and nv > mesh.num_vertices
    assert s.V.dofmap.index_map.size_global == nv and s.V.dofmap.index_map_bs == 3
    prof = lambda x: np.stack([16.0 * x[1] * (1 - x[1]) * x[2] * (1 - x[2]), 0 * x[0], 0 * x[0]])  # noqa: E731
    u_in = Function(s.V)
    u_in.interpolate(prof)
    bc_w = BoundaryCondition(Function(s.V))
    bc_w.initTopological(2, ft.find(3))
    bc_i = BoundaryCondition(u_in)
    bc_i.initTopological(2, ft.find(1))
    s.setup([bc_w, bc_i], [], ft, {"inlet": 1, "outlet": 2, "wall": 3, "obstacle": -1})
    prm = TN.Params(0.02, 1.0, 0.02, (0.0, 0.0, 0.0), ds_terms=False, beta_backflow=0.2)
    pb = problem3("P2", dm, prm)
    pb.set_boundary_terms(False, ft.find(2), 0.2)
    wn, inn = facet_node_set3(dm, ft.find(3)), facet_node_set3(dm, ft.find(1))
    pb.add_bc_u(wn, np.zeros((len(wn), 3)))
    pb.add_bc_u(inn, prof(dm.x[inn].T).T)
    x, un = np.zeros(4 * nv), np.zeros((nv, 3))
    for step in range(2):
        s.solveStep()
        assert s.last_stats.reason > 0
        xg = np.concatenate([np.asarray(s.u_sol.x.array), np.asarray(s.p_sol.x.array)])
        s.advance()
        x, _ = pb.newton(x, un)
        un = x[: 3 * nv].reshape(-1, 3).copy()
        assert np.abs(xg[: 3 * nv] - x[: 3 * nv]).max() <= 1e-8 * np.abs(x[: 3 * nv]).max(), step
        assert np.abs(xg[3 * nv:] - x[3 * nv:]).max() <= 1e-7 * np.abs(x[3 * nv:]).max(), step
