"""Tetrahedra (gdim = 3) on the HIP path against the dimension-generic NumPy twin (oracle/np_twin_nd.py, itself pinned
to the 2-D twin for d = 2): assembly (residual, every CSR value, SpMV) at states violating the Dirichlet data, time steps
with the boundary data of /root/reference/src/scenarios/simple_bifurcation.py:77-133 (direct solver on the twin side),
functionals.  fp64; the 3-D assembly accumulates in LDS in no fixed order, hence 1e-12 instead of 1e-13."""
import numpy as np
import pytest

from cfd_hemodynamic_amd import _lib
from cfd_hemodynamic_amd.mesh3d import Mesh3D, create_bifurcation, create_unit_cube
from oracle import np_twin_nd as TN

pytestmark = pytest.mark.gpu


def _cube(n, seed=0):
    m = create_unit_cube(n)
    rng = np.random.default_rng(seed)
    interior = np.abs(m.x - 0.5).max(axis=1) < 0.49
    return Mesh3D(m.cells, m.x + (0.12 / n) * rng.standard_normal(m.x.shape) * interior[:, None]), rng


def _pair(mesh, dt, rho, mu, f, bcs, **scheme):
    prm = TN.Params(dt, rho, mu, f, **scheme)
    pb = TN.Problem(mesh.x, mesh.cells, mesh.facet_cells, mesh.facet_local, prm)
    ctx = _lib.Context(mesh.x, mesh.cells, mesh.facet_cells, mesh.facet_local, mesh.facet_marker)
    assert ctx.dim == 3 and ctx.info(26) == 3
    ctx.set_params(dt, rho, mu, f=f)
    if scheme:
        ctx.set_time_scheme(scheme["theta"], scheme["a0"], scheme["a1"], scheme["a2"])
    for fld, nodes, vals in bcs:
        (pb.add_bc_u if fld == 0 else pb.add_bc_p)(nodes, vals)
        ctx.add_dirichlet(fld, nodes, vals)
    return pb, ctx


@pytest.mark.parametrize("scheme", [{}, dict(theta=1.0, a0=1.5, a1=-2.0, a2=0.5)])
def test_tet_assembly_and_spmv_match_twin(scheme):
    mesh, rng = _cube(4, 3)
    nv = mesh.num_vertices
    bnd = np.unique(mesh.facet_vertices)
    top = bnd[mesh.x[bnd, 2] > 0.99]
    side = bnd[mesh.x[bnd, 0] < 0.01]
    bcs = [(0, side.astype(np.int32), rng.standard_normal((len(side), 3))),
           (0, top.astype(np.int32), rng.standard_normal((len(top), 3))),       # shares an edge with `side`: multiplicity 2
           (1, bnd[mesh.x[bnd, 1] > 0.99].astype(np.int32), rng.standard_normal(int((mesh.x[bnd, 1] > 0.99).sum())))]
    pb, ctx = _pair(mesh, 0.02, 1.3, 0.04, (0.2, -0.1, 0.3), bcs, **scheme)
    xv = 0.3 * rng.standard_normal(4 * nv)
    un, un2 = 0.3 * rng.standard_normal((nv, 3)), 0.3 * rng.standard_normal((nv, 3))
    F, J = pb.assemble(xv, un, un2=un2)
    ctx.set_state(u_prev=un.ravel(), p_prev=np.zeros(nv), u=xv[: 3 * nv], p=xv[3 * nv:])
    if scheme:
        ctx.set_previous2(un2.ravel())
    ctx.assemble(True)
    Fg = np.concatenate(ctx.get_residual())
    assert np.abs(F - Fg).max() <= 1e-12 * np.abs(F).max()
    Jg = ctx.get_csr()
    assert Jg.shape == (4 * nv, 4 * nv) and Jg.nnz == 16 * ctx.info(3)
    assert abs(J - Jg).max() <= 1e-12 * abs(J).max()
    v = rng.standard_normal(4 * nv)
    assert np.abs(ctx.spmv(v) - J @ v).max() <= 1e-12 * np.abs(J @ v).max()
    ctx.assemble(False)  # residual with lifting, Jacobian in registers only
    assert np.abs(np.concatenate(ctx.get_residual()) - Fg).max() <= 1e-12 * np.abs(Fg).max()
    ctx.close()


def _bifurcation_bcs(mesh, ft, v_inlet=1.5):
    wall = np.unique(mesh.facet_vertices[ft.find(11)]).astype(np.int32)
    inl = np.setdiff1d(np.unique(mesh.facet_vertices[ft.find(8)]), wall).astype(np.int32)
    r = np.hypot(mesh.x[inl, 0], mesh.x[inl, 2])
    vin = np.zeros((len(inl), 3))
    vin[:, 1] = v_inlet * (1.0 - (r / mesh.r_in) ** 2)
    bcs = [(0, wall, np.zeros((len(wall), 3))), (0, inl, vin)]
    for tag in (9, 10):
        o = np.unique(mesh.facet_vertices[ft.find(tag)]).astype(np.int32)
        bcs.append((1, o, np.zeros(len(o))))
    return bcs


def test_tet_time_steps_match_twin_on_the_bifurcation():
    """simple_bifurcation.py:20-58,77-133: rho = 1, mu = 1/Re, inlet u_y = v (1 - (r / r_in)^2), no-slip walls, p = 0 at
    both outlets; two steps from rest, both sides converged tightly (twin: Newton with a direct solve)."""
    mesh, ft = create_bifurcation(0.0012)
    nv = mesh.num_vertices
    Re = 1055.0 * 0.01 * ((100 / 0.003918604) / 1e6) / 3.5e-3
    pb, ctx = _pair(mesh, 0.01, 1.0, 1.0 / Re, (0.0, 0.0, 0.0), _bifurcation_bcs(mesh, ft))
    o = ctx.default_options()
    o.snes_rtol, o.snes_stol, o.ksp_rtol = 1e-12, 0.0, 1e-10
    ctx.set_options(o)
    z3, z1 = np.zeros(3 * nv), np.zeros(nv)
    ctx.set_state(u_prev=z3, p_prev=z1, u=z3, p=z1)
    x = np.zeros(4 * nv)
    un = np.zeros((nv, 3))
    for step in range(2):
        st = ctx.solve_step()
        assert st.reason > 0 and st.newton_its <= 8
        xg = np.concatenate(ctx.get_solution())
        ctx.advance()
        x, hist = pb.newton(x, un)
        un = x[: 3 * nv].reshape(-1, 3).copy()
        assert np.linalg.norm(xg[: 3 * nv] - x[: 3 * nv]) <= 1e-8 * np.linalg.norm(x[: 3 * nv]), step
        assert np.linalg.norm(xg[3 * nv:] - x[3 * nv:]) <= 1e-7 * np.linalg.norm(x[3 * nv:]), step
    l2u, l2p = pb.l2_norms(x)
    assert abs(ctx.functional(2) - l2u) <= 1e-8 * l2u and abs(ctx.functional(3) - l2p) <= 1e-7 * l2p
    for tag in (8, 9, 10, 11):
        q = pb.flux(x, ft.find(tag))
        assert abs(ctx.functional(7, tag) - q) <= 1e-7 * abs(pb.flux(x, ft.find(8))) + 1e-18
    u, _ = ctx.get_solution()
    assert ctx.functional(4) == np.abs(u).max()
    # wall shear stress: tangential, concentrated on the walls, zero in the interior
    w = ctx.wall_shear_stress().reshape(-1, 3)
    interior = np.ones(nv, bool)
    interior[np.unique(mesh.facet_vertices)] = False
    assert np.abs(w).max() > 0 and not w[interior].any()
    wt = pb.wall_shear_stress(x).reshape(-1, 3)      # twin restatement of assemble_wss (solverBase.py:163-195)
    assert np.abs(w - wt).max() <= 1e-7 * np.abs(wt).max()
    ctx.close()


def test_simple_bifurcation_scenario_through_the_plugin_surface(tmp_path):
    """MicrovasculatureSimulation (simple_bifurcation.py) end to end on the 3-D path: Scenario.solve with output, the
    literal state copy, flux balance improving with resolution, VTU files with tetrahedra."""
    from cfd_hemodynamic_amd.io import read_vtu
    from cfd_hemodynamic_amd.scenarios.simple_bifurcation import MicrovasculatureSimulation
    sc = MicrovasculatureSimulation("stabilized_schur", 0.01, 0.025, res=8e-4, quiet=True)
    assert abs(sc.Re - 1055.0 * 0.01 * sc.L_c / 3.5e-3) < 1e-12 and float(sc.solver.mu.value) == 1.0 / sc.Re
    assert sc.solver.V.dofmap.index_map_bs == 3 and sc.mesh.topology.dim == 3
    out = sc.solve(str(tmp_path / "run"), device_resident=False)   # the reference's literal loop
    assert sc.num_steps == 3 and sc.solver.transfers["uploads"] <= 2  # initial state only (setup + first step)
    qi, q1, q2 = sc.flow_rates()
    assert qi > 0 and 0.5 * qi < q1 + q2 < 1.05 * qi and abs(q1 - q2) < 0.1 * qi
    u = np.asarray(sc.solver.u_sol.x.array).reshape(-1, 3)
    assert np.abs(u[:, 1]).max() > np.abs(u[:, 0]).max() > 0  # axial flow dominates, the branches deflect it
    v = read_vtu(str(tmp_path / "run" / "v_000003.vtu"))
    assert v["cells"].shape == (sc.mesh.num_cells, 4) and np.allclose(v["v"], u)
    assert abs(sc.norm_v - sc.solver.functional(2)) <= 1e-12 * sc.norm_v
    # finer mesh: better flux balance (PSPG mass defect is first order in h on this staircase mesh)
    fine = MicrovasculatureSimulation("stabilized_schur", 0.01, 0.025, res=4e-4, quiet=True)
    fine.solver.solveStep()
    f_i, f_1, f_2 = fine.flow_rates()
    assert abs((f_1 + f_2) / f_i - 1.0) < abs((q1 + q2) / qi - 1.0)


def test_simple_bifurcation_from_a_gmsh_file(tmp_path):
    """The reference's own route: `gmshio.read_from_msh("meshes/simple_bifurcation.msh", ..., gdim=3)`
    (simple_bifurcation.py:71-75).  The same mesh through a .msh file gives the same step."""
    from cfd_hemodynamic_amd.meshio import write_msh
    from cfd_hemodynamic_amd.scenarios.simple_bifurcation import MicrovasculatureSimulation
    a = MicrovasculatureSimulation("stabilized_schur", 0.01, 0.025, res=8e-4, quiet=True)
    p = str(tmp_path / "simple_bifurcation.msh")
    write_msh(p, a.mesh, a._ft, cell_tag=a.fluid_tag)
    b = MicrovasculatureSimulation("stabilized_schur", 0.01, 0.025, mesh_file=p, quiet=True)
    assert b.mesh.num_cells == a.mesh.num_cells and (b.mesh.cell_tags == a.fluid_tag).all()
    a.solver.solveStep(); b.solver.solveStep()
    ua, ub = np.asarray(a.solver.u_sol.x.array), np.asarray(b.solver.u_sol.x.array)
    assert np.abs(ua - ub).max() <= 1e-9 * np.abs(ua).max()
    assert abs(a.flow_rates()[1] - b.flow_rates()[1]) <= 1e-9 * abs(a.flow_rates()[1])


@pytest.mark.parametrize("res", [4e-4, 2e-4])
def test_tet_assembly_matches_twin_on_the_bifurcation_at_size(res):
    """Workgroup packing at size: res 4e-4 = 138 264 DOF (174 984 tetrahedra), res 2e-4 = 1 025 036 DOF (the c5b bench mesh,
    1.47 M tetrahedra, 62 M scalar CSR values); boundary data of the scenario, state off the Dirichlet values: residual, every
    CSR value and the SpMV against the NumPy twin."""
    mesh, ft = create_bifurcation(res)
    nv = mesh.num_vertices
    pb, ctx = _pair(mesh, 0.01, 1.0, 0.013, (0.0, 0.0, 0.0), _bifurcation_bcs(mesh, ft))
    rng = np.random.default_rng(11)
    xv = 0.3 * rng.standard_normal(4 * nv)
    un = 0.3 * rng.standard_normal((nv, 3))
    F, J = pb.assemble(xv, un)
    ctx.set_state(u_prev=un.ravel(), p_prev=np.zeros(nv), u=xv[: 3 * nv], p=xv[3 * nv:])
    ctx.assemble(True)
    Fg = np.concatenate(ctx.get_residual())
    assert np.abs(F - Fg).max() <= 1e-12 * np.abs(F).max()
    Jg = ctx.get_csr()
    assert Jg.nnz == 16 * ctx.info(3) and abs(J - Jg).max() <= 1e-12 * abs(J).max()
    v = rng.standard_normal(4 * nv)
    assert np.abs(ctx.spmv(v) - J @ v).max() <= 1e-12 * np.abs(J @ v).max()
    ctx.assemble(False)
    assert np.abs(np.concatenate(ctx.get_residual()) - Fg).max() <= 1e-12 * np.abs(Fg).max()
    ctx.close()


def test_bdf2_plugin_on_tetrahedra_matches_twin():
    """`--solver stabilized_schur_bdf2` on the 3-D scenario: BDF1 first step, BDF2 afterwards (stabilized_schur_bdf2.py:
    79-110,298-325), u_prev2 shifted on the device; three steps in lockstep with the twin (theta = 1 and the a0/a1/a2 of
    each step, direct solves)."""
    from cfd_hemodynamic_amd.scenarios.simple_bifurcation import MicrovasculatureSimulation
    # remove_p_mean 0: with the mean removal the Newton tolerance is relative to the outlet-row misfit (see the scenario's
    # docstring) and the momentum rows are solved to ~1e-7 only
    tight = dict(snes_rtol=1e-12, snes_stol=0.0, ksp_rtol=1e-10, remove_p_mean=0)
    sc = MicrovasculatureSimulation("stabilized_schur_bdf2", 0.01, 1.0, res=1.2e-3, quiet=True, options=tight)
    mesh, nv = sc.mesh, sc.mesh.num_vertices
    assert sc.solver.u_prev2.x.array.shape == (3 * nv,)
    bcs = _bifurcation_bcs(mesh, sc._ft)
    x = np.zeros(4 * nv)
    un = np.zeros((nv, 3)); un2 = np.zeros((nv, 3))
    for step in range(3):
        a = (1.0, -1.0, 0.0) if step == 0 else (1.5, -2.0, 0.5)
        prm = TN.Params(0.01, 1.0, 1.0 / sc.Re, (0.0, 0.0, 0.0), theta=1.0, a0=a[0], a1=a[1], a2=a[2])
        pb = TN.Problem(mesh.x, mesh.cells, mesh.facet_cells, mesh.facet_local, prm)
        for fld, nodes, vals in bcs:
            (pb.add_bc_u if fld == 0 else pb.add_bc_p)(nodes, vals)
        x, _ = pb.newton(x, un, un2=un2)
        un2, un = un, x[: 3 * nv].reshape(-1, 3).copy()
        sc.solver.solveStep()
        ug = np.asarray(sc.solver.u_sol.x.array)
        assert np.linalg.norm(ug - x[: 3 * nv]) <= 1e-8 * np.linalg.norm(x[: 3 * nv]), step
        sc.solver.advance()
    assert sc.solver.step_count == 3
    assert np.linalg.norm(np.asarray(sc.solver.u_prev2.x.array) - un2.ravel()) <= 1e-8 * np.linalg.norm(un2)


def test_tet_backflow_term_matches_twin():
    """stabilized_schur_backflow on tetrahedra: no ds pair, backflow stabilisation on the outlet facets (both outlets of the
    bifurcation share marker 9 here), u_prev with reverse flow through them; residual and every CSR value against the twin,
    then three time steps (do-nothing outlets, no pressure condition: singular pressure handled by the null-space logic)."""
    mesh, ft = create_bifurcation(0.0012)
    nv = mesh.num_vertices
    marker = mesh.facet_marker.copy()
    marker[marker == 10] = 9
    out = np.nonzero(marker == 9)[0]
    bcs = [b for b in _bifurcation_bcs(mesh, ft) if b[0] == 0]   # walls + inlet; no pressure condition
    pb, ctx = _pair(mesh, 0.01, 1.0, 0.013, (0.0, 0.0, 0.0), bcs)
    ctx.set_facet_markers(marker)
    ctx.set_boundary_terms(False, 9, 0.5)
    pb.set_boundary_terms(False, out, 0.5)
    rng = np.random.default_rng(5)
    xv = 0.3 * rng.standard_normal(4 * nv)
    un = 0.3 * rng.standard_normal((nv, 3))
    un[:, 1] -= 0.2          # net inflow through the outlets (their normals point along +y)
    F, J = pb.assemble(xv, un)
    pb.set_boundary_terms(False, None, 0.0)
    F0, _ = pb.assemble(xv, un)
    pb.set_boundary_terms(False, out, 0.5)
    assert np.abs(F - F0).max() > 1e-9 * np.abs(F).max()          # the term is active
    ctx.set_state(u_prev=un.ravel(), p_prev=np.zeros(nv), u=xv[: 3 * nv], p=xv[3 * nv:])
    ctx.assemble(True)
    Fg = np.concatenate(ctx.get_residual())
    assert np.abs(F - Fg).max() <= 1e-12 * np.abs(F).max()
    assert abs(J - ctx.get_csr()).max() <= 1e-12 * abs(J).max()
    ctx.assemble(False)
    assert np.abs(np.concatenate(ctx.get_residual()) - Fg).max() <= 1e-12 * np.abs(Fg).max()
    # time steps from rest with the scenario's inlet: both sides converged tightly
    o = ctx.default_options()
    o.snes_rtol, o.snes_stol, o.ksp_rtol = 1e-11, 0.0, 1e-9
    ctx.set_options(o)
    z3, z1 = np.zeros(3 * nv), np.zeros(nv)
    ctx.set_state(u_prev=z3, p_prev=z1, u=z3, p=z1)
    x = np.zeros(4 * nv); unn = np.zeros((nv, 3))
    for step in range(2):
        st = ctx.solve_step()
        assert st.reason > 0
        xg = np.concatenate(ctx.get_solution())
        ctx.advance()
        x, _ = pb.newton(x, unn)
        unn = x[: 3 * nv].reshape(-1, 3).copy()
        assert np.linalg.norm(xg[: 3 * nv] - x[: 3 * nv]) <= 1e-7 * np.linalg.norm(x[: 3 * nv]), step
        pg, pt = xg[3 * nv:], x[3 * nv:]
        assert np.linalg.norm((pg - pg.mean()) - (pt - pt.mean())) <= 1e-6 * np.linalg.norm(pt - pt.mean()), step
    ctx.close()


def test_backflow_plugin_runs_the_bifurcation_scenario():
    """`--simulation simple_bifurcation --solver stabilized_schur_backflow --v_max 1.5`: the plugin class on tetrahedra (pressure
    conditions ignored, do-nothing outlets, backflow term on tags["outlet"] = outlet 1)."""
    from cfd_hemodynamic_amd.scenarios.simple_bifurcation import MicrovasculatureSimulation
    sc = MicrovasculatureSimulation("stabilized_schur_backflow", 0.01, 0.025, res=8e-4, quiet=True, v_max=1.5, beta_backflow=0.2)
    assert sc.solver.bcp_d == [] and sc.solver.ctx.dim == 3
    sc.solve(None)
    qi, q1, q2 = sc.flow_rates()
    assert sc.num_steps == 3 and qi > 0 and 0.5 * qi < q1 + q2 < 1.05 * qi
    assert np.isfinite(sc.norm_p) and sc.norm_v > 0


def test_tet_assembly_matches_the_c_oracle(monkeypatch):
    """Third implementation of the tetrahedral element algebra: oracle/cfdh_oracle3.c (plain C loops) feeds the same
    assembly / Dirichlet logic as the twin; HIP residual, CSR values and SpMV against it at 138 k DOF, with the ds pair
    on the whole boundary and (second pass) the backflow term on the outlets."""
    from oracle import orc3
    monkeypatch.setattr(TN, "element_tensors", orc3.element_tensors)
    mesh, ft = create_bifurcation(4e-4)
    nv = mesh.num_vertices
    marker = mesh.facet_marker.copy()
    marker[marker == 10] = 9
    rng = np.random.default_rng(21)
    for backflow in (False, True):
        bcs = _bifurcation_bcs(mesh, ft)
        if backflow:
            bcs = [b for b in bcs if b[0] == 0]
        pb, ctx = _pair(mesh, 0.01, 1.0, 0.013, (0.0, 0.1, 0.0), bcs)
        if backflow:
            ctx.set_facet_markers(marker)
            ctx.set_boundary_terms(False, 9, 0.4)
            pb.set_boundary_terms(False, np.nonzero(marker == 9)[0], 0.4)
        xv = 0.3 * rng.standard_normal(4 * nv)
        un = 0.3 * rng.standard_normal((nv, 3))
        un[:, 1] -= 0.2
        F, J = pb.assemble(xv, un)
        ctx.set_state(u_prev=un.ravel(), p_prev=np.zeros(nv), u=xv[: 3 * nv], p=xv[3 * nv:])
        ctx.assemble(True)
        Fg = np.concatenate(ctx.get_residual())
        assert np.abs(F - Fg).max() <= 1e-12 * np.abs(F).max(), backflow
        assert abs(J - ctx.get_csr()).max() <= 1e-12 * abs(J).max(), backflow
        v = rng.standard_normal(4 * nv)
        assert np.abs(ctx.spmv(v) - J @ v).max() <= 1e-12 * np.abs(J @ v).max()
        ctx.close()


def test_tet_time_steps_match_the_c_oracle_solver_at_138k_dof():
    """Step parity at size: the bifurcation at res 4e-4 (138 k DOF) against the C oracle's own Newton / FGMRES / Cahouet-Chabard +
    AMG driver on tetrahedra (oracle/cfdh_oracle.c with the element tensors of cfdh_oracle3.c) -- the twin's direct solve stops
    at ~10^4 DOF.  Both sides converged tightly; two steps from rest."""
    from oracle import orc
    mesh, ft = create_bifurcation(4e-4)
    nv = mesh.num_vertices
    assert 4 * nv > 100000
    Re = 1055.0 * 0.01 * ((100 / 0.003918604) / 1e6) / 3.5e-3
    bcs = _bifurcation_bcs(mesh, ft)
    ctx = _lib.Context(mesh.x, mesh.cells, mesh.facet_cells, mesh.facet_local, mesh.facet_marker)
    ctx.set_params(0.01, 1.0, 1.0 / Re, f=(0.0, 0.0, 0.0))
    O = orc.Oracle(mesh.x, mesh.cells, mesh.facet_cells, mesh.facet_local, 0.01, 1.0, 1.0 / Re, (0.0, 0.0, 0.0))
    for fld, nodes, vals in bcs:
        ctx.add_dirichlet(fld, nodes, vals)
        (O.add_bc_u if fld == 0 else O.add_bc_p)(nodes, vals)
    o = ctx.default_options()
    o.snes_rtol, o.snes_stol, o.ksp_rtol, o.remove_p_mean = 1e-11, 0.0, 1e-9, 0
    ctx.set_options(o)
    oo = orc.default_opts(pc_kind=2, snes_rtol=1e-11, snes_stol=0.0, ksp_rtol=1e-9, remove_p_mean=0)
    z3, z1 = np.zeros(3 * nv), np.zeros(nv)
    ctx.set_state(u_prev=z3, p_prev=z1, u=z3, p=z1)
    x = np.zeros(4 * nv)
    for step in range(2):
        st = ctx.solve_step()
        assert st.reason > 0
        xg = np.concatenate(ctx.get_solution())
        ctx.advance()
        O.set_un(x[: 3 * nv].copy())
        x, so = O.solve_step(x, oo)
        assert np.linalg.norm(xg[: 3 * nv] - x[: 3 * nv]) <= 1e-7 * np.linalg.norm(x[: 3 * nv]), step
        assert np.linalg.norm(xg[3 * nv:] - x[3 * nv:]) <= 1e-6 * np.linalg.norm(x[3 * nv:]), step
    ctx.close()


def test_projected_initial_guess_on_tetrahedra(monkeypatch):
    """The projected initial guess of the linear solves (cfdh_options.ksp_guess) on the 4 x 4-block path: the multi-vector SpMV
    of the kept corrections reproduces the true residual of the guess (library-side check under CFDH_GUESS_CHECK), and the
    converged fields do not depend on the guess."""
    from cfd_hemodynamic_amd.scenarios.simple_bifurcation import MicrovasculatureSimulation
    monkeypatch.setenv("CFDH_GUESS_CHECK", "1")
    out = {}
    for guess in (0, 4):
        sc = MicrovasculatureSimulation("stabilized_schur", 0.01, 1.0, res=6e-4, quiet=True,
                                        options=dict(ksp_guess=guess, snes_rtol=1e-12, snes_stol=0.0, ksp_rtol=1e-10))
        its = 0
        for _ in range(7):
            sc.solver.solveStep()
            sc.solver.advance()
            assert sc.solver.last_stats.reason > 0
            its += sc.solver.last_stats.krylov_its
        out[guess] = (np.asarray(sc.solver.u_sol.x.array).copy(), np.asarray(sc.solver.p_sol.x.array).copy(), its, sc.solver.ctx.info(70))
    u0, p0, its0, n0 = out[0]
    u4, p4, its4, n4 = out[4]
    assert n0 == 0 and n4 >= 12 and its4 < its0
    # (3-D: |F0| of a step carries the outlet-row misfit, so snes_rtol is loose in absolute terms -- DESIGN.md, "tolerance trap")
    assert np.linalg.norm(u4 - u0) <= 1e-6 * np.linalg.norm(u0) and np.linalg.norm(p4 - p0) <= 1e-5 * np.linalg.norm(p0)


def test_tet_assembly_is_bitwise_reproducible():
    """Round 4: the rows of an assembly workgroup are dealt to its wavefronts and every LDS accumulator receives its contributions
    from one wavefront in program order, so the tetrahedral assembly gives the same bits in every pass and every context (rounds
    2-3 summed in LDS-arrival order: reproducible to 1e-12 only).  Bifurcation mesh, ~36 k vertices."""
    from cfd_hemodynamic_amd.mesh3d import create_bifurcation
    mesh, ft = create_bifurcation(4e-4)
    nv = mesh.num_vertices
    rng = np.random.default_rng(21)
    xv = 0.3 * rng.standard_normal(4 * nv)
    un = 0.3 * rng.standard_normal((nv, 3))
    bcs = _bifurcation_bcs(mesh, ft)
    out = []
    for _ in range(2):
        _, ctx = _pair(mesh, 0.01, 1.0, 1.0 / 76.9, (0.0, 0.0, 0.0), bcs)
        ctx.set_state(u_prev=un.ravel(), p_prev=np.zeros(nv), u=xv[: 3 * nv], p=xv[3 * nv:])
        for _rep in range(3):
            ctx.assemble(True)
            out.append((np.concatenate(ctx.get_residual()), ctx.get_csr().data.copy()))
        ctx.close()
    for F, A in out[1:]:
        assert np.array_equal(F, out[0][0]) and np.array_equal(A, out[0][1])
