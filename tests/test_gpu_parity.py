"""Parity tests proper: the HIP path through the C-ABI against the CPU oracle on the same seeded
inputs, the committed golden vectors, and size-independent properties at the BASELINE sizes.
Tolerances (fp64): element/assembly/SpMV 1e-13 relative; converged step solution 1e-9;
drag/lift/L2 1e-8 (north_star asks 1e-6)."""
import numpy as np
import pytest

from util import dfg_case, lid_case, load_golden, make_ctx, make_oracle, stenosis_case

pytestmark = pytest.mark.gpu

TIGHT = dict(snes_rtol=1e-12, snes_stol=0.0, ksp_rtol=1e-10)


def _tight(ctx, **kw):
    o = ctx.default_options()
    for k, v in {**TIGHT, **kw}.items():
        setattr(o, k, v)
    ctx.set_options(o)
    return o


def _rand_state(nv, seed):
    rng = np.random.default_rng(seed)
    return 0.1 * rng.standard_normal(3 * nv), 0.1 * rng.standard_normal(2 * nv)


@pytest.mark.parametrize("case_fn,arg", [(dfg_case, 12), (lid_case, 16), (dfg_case, 37)])
def test_assembly_and_spmv_match_oracle(case_fn, arg):
    case = case_fn(arg)
    nv = case.nv
    O, ctx = make_oracle(case), make_ctx(case)
    xv, un = _rand_state(nv, 11)  # does not satisfy the Dirichlet data: lifting is exercised
    O.set_un(un)
    F = O.assemble(xv)
    J = O.csr()
    ctx.set_state(u_prev=un, p_prev=np.zeros(nv), u=xv[: 2 * nv], p=xv[2 * nv:])
    ctx.assemble(True)
    Fg = np.concatenate(ctx.get_residual())
    Jg = ctx.get_csr()
    assert np.abs(F - Fg).max() <= 1e-13 * np.abs(F).max()
    assert abs(J - Jg).max() <= 1e-13 * abs(J).max()
    v = np.random.default_rng(3).standard_normal(3 * nv)
    assert np.abs(ctx.spmv(v) - J @ v).max() <= 1e-13 * np.abs(J @ v).max()
    # residual-only evaluation (lifting still applied) gives the same F (a separately compiled kernel
    # instance: equal to round-off, not bitwise)
    ctx.assemble(False)
    assert np.abs(np.concatenate(ctx.get_residual()) - Fg).max() <= 1e-14 * np.abs(Fg).max()
    ctx.close()


def test_assembly_is_bitwise_reproducible():
    """No atomics: wavefront segmented reduction in a fixed order -> identical bits run to run."""
    case = dfg_case(24)
    nv = case.nv
    xv, un = _rand_state(nv, 5)
    outs = []
    for _ in range(2):
        ctx = make_ctx(case)
        ctx.set_state(u_prev=un, p_prev=np.zeros(nv), u=xv[: 2 * nv], p=xv[2 * nv:])
        ctx.assemble(True)
        outs.append((np.concatenate(ctx.get_residual()), ctx.get_csr().data.copy()))
        ctx.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("name", ["dfg_m6", "lid_n8"])
def test_golden_vectors(name):
    import scipy.sparse as sp
    case, g = load_golden(name)
    nv = case.nv
    ctx = make_ctx(case)
    ctx.set_state(u_prev=g["u_prev"].ravel(), p_prev=np.zeros(nv), u=g["state"][: 2 * nv], p=g["state"][2 * nv:])
    ctx.assemble(True)
    F = np.concatenate(ctx.get_residual())
    assert np.abs(F - g["F"]).max() <= 1e-13 * np.abs(g["F"]).max()
    Jg = sp.csr_matrix((g["J_data"], g["J_indices"], g["J_indptr"]), shape=(3 * nv, 3 * nv))
    assert abs(ctx.get_csr() - Jg).max() <= 1e-13 * abs(Jg).max()
    _tight(ctx)
    z2, z1 = np.zeros(2 * nv), np.zeros(nv)
    ctx.set_state(u_prev=z2, p_prev=z1, u=z2, p=z1)
    for k in (1, 2):
        st = ctx.solve_step()
        assert st.reason > 0
        x = np.concatenate(ctx.get_solution())
        ctx.advance()
        ref = g["step%d" % k]
        assert np.linalg.norm(x - ref) <= 1e-9 * np.linalg.norm(ref)
    assert np.allclose([ctx.functional(2), ctx.functional(3)], g["l2"], rtol=1e-9)
    if "drag_lift" in g:
        assert np.allclose([ctx.functional(0, 5), ctx.functional(1, 5)], g["drag_lift"], rtol=1e-7, atol=1e-12)
    ctx.close()


@pytest.mark.parametrize("case_fn,arg,nsteps", [(dfg_case, 16, 4), (lid_case, 24, 3), (stenosis_case, 12, 3)])
def test_time_steps_match_oracle(case_fn, arg, nsteps):
    """Same mesh, dt, initial state; both sides converged tightly; the oracle runs the reference's
    own preconditioner configuration (ILU-based), the GPU its Chebyshev/AMG one."""
    from oracle import orc
    case = case_fn(arg)
    nv = case.nv
    O, ctx = make_oracle(case), make_ctx(case)
    _tight(ctx)
    z2, z1 = np.zeros(2 * nv), np.zeros(nv)
    ctx.set_state(u_prev=z2, p_prev=z1, u=z2, p=z1)
    x = np.zeros(3 * nv)
    O.set_un(z2)
    opts = orc.default_opts(snes_rtol=1e-12, snes_stol=0.0, ksp_rtol=1e-10, sub_rtol=1e-8)
    for _ in range(nsteps):
        st = ctx.solve_step()
        assert st.reason > 0 and st.newton_its <= 6
        xg = np.concatenate(ctx.get_solution())
        ctx.advance()
        x, _ = O.solve_step(x, opts)
        O.set_un(x[: 2 * nv])
        assert np.linalg.norm(xg - x) <= 1e-9 * np.linalg.norm(x)
    assert abs(ctx.functional(2) - O.functional(x, 2)) <= 1e-10 * O.functional(x, 2)
    assert abs(ctx.functional(3) - O.functional(x, 3)) <= 1e-8 * O.functional(x, 3)
    if "ft" in case.markers:
        obst = case.markers["ft"].find(5)
        for kind in (0, 1):
            a, b = ctx.functional(kind, 5), O.functional(x, kind, obst)
            assert abs(a - b) <= 1e-8 * abs(b) + 1e-14
    # inf-norm functionals of the early-stop test (scenario.py:268-280)
    u, _ = ctx.get_solution()
    up, _ = ctx.get_previous()
    assert ctx.functional(4) == np.abs(u).max() and ctx.functional(6) == np.abs(u - up).max()
    ctx.close()


def test_reference_default_tolerances_and_divergence_error():
    case = dfg_case(16)
    nv = case.nv
    ctx = make_ctx(case)
    z2, z1 = np.zeros(2 * nv), np.zeros(nv)
    ctx.set_state(u_prev=z2, p_prev=z1, u=z2, p=z1)
    st = ctx.solve_step()  # PETSc defaults
    assert st.reason in (3, 4) and st.newton_its <= 5 and st.krylov_its > 0
    # RuntimeError exactly like stabilized_schur.py:332-334 when Newton cannot converge
    o = ctx.default_options()
    o.snes_max_it, o.snes_rtol, o.snes_stol = 1, 1e-15, 0.0
    ctx.set_options(o)
    ctx.set_state(u_prev=z2, p_prev=z1, u=z2, p=z1)
    with pytest.raises(RuntimeError, match=r"Did not converge, reason: -5"):
        ctx.solve_step()
    ctx.close()


def test_facet_markers_can_be_set_after_create():
    """The reference hands facet_tags to Solver.setup(), after the spaces exist (scenario.py:146-149): a context
    created with untagged facets must give drag/lift by marker once cfdh_set_facet_markers delivered the tags."""
    from cfd_hemodynamic_amd import _lib
    case = dfg_case(10)
    m, nv = case.mesh, case.nv
    a = make_ctx(case)  # markers given at creation
    z2, z1 = np.zeros(2 * nv), np.zeros(nv)
    a.set_state(u_prev=z2, p_prev=z1, u=z2, p=z1)
    a.solve_step()
    u, p = a.get_solution()
    b = _lib.Context(m.x, m.cells, m.facet_cells, m.facet_local, np.zeros_like(m.facet_marker))
    b.set_params(case.dt, case.rho, case.mu, f=case.f)
    b.set_state(u_prev=z2, p_prev=z1, u=u, p=p)
    assert b.functional(0, 5) == 0.0 and b.functional(1, 5) == 0.0  # no facet carries marker 5 yet
    with pytest.raises(ValueError, match="markers"):
        b.set_facet_markers(m.facet_marker[:-1])
    b.set_facet_markers(m.facet_marker)
    assert b.functional(0, 5) == a.functional(0, 5) != 0.0
    assert b.functional(1, 5) == a.functional(1, 5)
    a.close()
    b.close()


def test_solver_class_and_scenario_drop_in():
    """The plugin surface end to end: Scenario loop (device-resident and the reference's literal
    host-copy loop) -> identical results; drag/lift against the oracle-driven loop."""
    from cfd_hemodynamic_amd.scenarios.dfg_1 import DFG1Benchmark
    from oracle import orc
    kw = dict(m=12, quiet=True, options=dict(TIGHT))
    a = DFG1Benchmark("stabilized_schur", 0.01, 0.12, **kw)
    a.setup()  # the reference calls setup() twice (dfg_1.py:38 and simulation.py:269)
    a.solve(None, device_resident=True)
    b = DFG1Benchmark("stabilized_schur", 0.01, 0.12, **kw)
    b.solve(None, device_resident=False)
    assert a.num_steps == b.num_steps == 13
    assert np.array_equal(a.solver.u_sol.x.array, b.solver.u_sol.x.array)
    assert np.array_equal(a.solver.p_sol.x.array, b.solver.p_sol.x.array)
    assert abs(a.norm_v - b.norm_v) <= 1e-12 * a.norm_v
    # oracle-driven loop
    case = dfg_case(12)
    nv = case.nv
    O = make_oracle(case)
    x = np.zeros(3 * nv)
    O.set_un(np.zeros(2 * nv))
    opts = orc.default_opts(snes_rtol=1e-12, snes_stol=0.0, ksp_rtol=1e-10, pc_kind=1)
    for _ in range(13):
        x, _ = O.solve_step(x, opts)
        O.set_un(x[: 2 * nv])
    obst = case.markers["ft"].find(5)
    cd, cl = 500 * O.functional(x, 0, obst), 500 * O.functional(x, 1, obst)
    assert abs(a.drag - cd) <= 1e-8 * abs(cd) and abs(a.lift - cl) <= 1e-7 * abs(cl)
    assert abs(a.norm_v - O.functional(x, 2)) <= 1e-10 * a.norm_v
    xg = np.concatenate([a.solver.u_sol.x.array, a.solver.p_sol.x.array])
    assert np.linalg.norm(xg - x) <= 1e-9 * np.linalg.norm(x)
    assert a.p_diff is not None and np.isfinite(a.p_diff)
    assert a.solver.shear_stress.x.array.any()


def test_literal_state_copy_of_the_reference_loop_stays_on_the_device():
    """`u_prev.x.array[:] = u_sol.x.array[:]` (scenario.py:306-307) through the lazy array proxy: no whole-field
    transfer in either direction, bitwise the same fields as the `advance()` loop; ordinary host access still sees
    current data, and a host write to u_prev is still picked up by the next step."""
    from cfd_hemodynamic_amd.scenarios.dfg_1 import DFG1Benchmark
    a = DFG1Benchmark("stabilized_schur", 0.01, 1.0, m=12, quiet=True)
    b = DFG1Benchmark("stabilized_schur", 0.01, 1.0, m=12, quiet=True)
    t0 = dict(b.solver.transfers)
    for _ in range(3):
        a.solver.solveStep()
        a.solver.advance()
        b.solver.solveStep()
        b.solver.u_prev.x.array[:] = b.solver.u_sol.x.array[:]
        b.solver.p_prev.x.array[:] = b.solver.p_sol.x.array[:]
    assert b.solver.transfers == t0
    assert np.array_equal(np.asarray(a.solver.u_sol.x.array), np.asarray(b.solver.u_sol.x.array))
    assert np.array_equal(a.solver.u_prev.x.array.copy(), b.solver.u_prev.x.array.copy())  # downloads on demand
    assert b.solver.transfers["downloads"] == t0["downloads"] + 2
    assert np.array_equal(a.solver.p_prev.x.array[:], b.solver.p_sol.x.array[:])
    # numpy semantics of the proxy
    u = b.solver.u_sol.x.array
    assert u.shape == (2 * b.mesh.num_vertices,) and len(u) == u.size and u.dtype == np.float64
    assert np.linalg.norm(u, ord=np.inf) == np.abs(u).max() == u.reshape(-1, 2).__abs__().max()
    assert float((u - u)[0]) == 0.0 and (2.0 * u)[3] == 2.0 * u[3]
    # a host write goes through the host path and reaches the device at the next step
    b.solver.u_prev.x.array[:] = 0.0
    b.solver.p_prev.x.array[:] = np.zeros(b.mesh.num_vertices)
    n_up = b.solver.transfers["uploads"]
    b.solver.solveStep()
    assert b.solver.transfers["uploads"] == n_up + 1
    up, _ = b.solver.ctx.get_previous()
    assert not up.any()
    # assignment from a foreign array-like is not mistaken for the device idiom
    other = DFG1Benchmark("stabilized_schur", 0.01, 1.0, m=12, quiet=True)
    other.solver.solveStep()
    b.solver.u_prev.x.array[:] = other.solver.u_sol.x.array[:]
    assert np.array_equal(np.asarray(b.solver.u_prev.x.array), np.asarray(other.solver.u_sol.x.array))


def test_time_dependent_dirichlet_values():
    """bc.update() re-reads the source every step (stabilized_schur.py:170): a pulsatile inlet
    `v(t) = v0 (1 + 0.5 sin 2 pi t)` through the Function the BoundaryCondition wraps."""
    from cfd_hemodynamic_amd.scenarios.dfg_1 import DFG1Benchmark
    sc = DFG1Benchmark("stabilized_schur", 0.01, 0.05, m=8, quiet=True)
    src = sc.bcu[0].f
    base = src.x.array.copy()

    def cb(t):
        src.x.array[:] = base * (1.0 + 0.5 * np.sin(2 * np.pi * t))

    sc.solve(None, afterStepCallback=cb)
    u = sc.solver.u_sol.x.array.reshape(-1, 2)
    inl = sc.solver.bcu_d[0].dofs
    # the callback takes effect one step late: last step used t = T - dt
    t_used = sc.t_end - sc.dt
    assert np.allclose(u[inl], (base * (1.0 + 0.5 * np.sin(2 * np.pi * t_used))).reshape(-1, 2)[inl], atol=1e-14)


def test_dirichlet_sets_can_be_replaced_sparse_updates_leave_nothing_stale():
    """cfdh_clear_dirichlet / cfdh_add_dirichlet touch and upload only the vertices involved (a pulsatile inlet re-sends its
    objects every step).  Replace the whole Dirichlet set by a different one, append to it without a clear, change values only:
    after each change the device assembly equals a fresh context's (bitwise) and the oracle's."""
    case = dfg_case(12)
    m = case.mesh
    nv = m.num_vertices
    rng = np.random.default_rng(2)
    x = rng.standard_normal(3 * nv)
    un = rng.standard_normal(2 * nv)
    bnd = np.unique(m.facet_vertices)
    setA = [(0, bnd[: len(bnd) // 2].astype(np.int32), rng.standard_normal((len(bnd) // 2, 2)))]
    setB = [(0, bnd[len(bnd) // 3:].astype(np.int32), rng.standard_normal((len(bnd) - len(bnd) // 3, 2))),
            (1, bnd[:7].astype(np.int32), rng.standard_normal(7))]
    extra = (0, bnd[5:25].astype(np.int32), rng.standard_normal((20, 2)))   # overlaps setB: multiplicity 2 on those dofs

    def assembled(ctx):
        ctx.set_state(u_prev=un, p_prev=np.zeros(nv), u=x[: 2 * nv], p=x[2 * nv:])
        ctx.assemble(True)
        return np.concatenate(ctx.get_residual()), ctx.get_csr()

    def fresh(bcs):
        ctx2 = make_ctx(type("C", (), dict(mesh=m, bcs=bcs, dt=case.dt, rho=case.rho, mu=case.mu, f=case.f))())
        out = assembled(ctx2)
        ctx2.close()
        return out

    ctx = make_ctx(type("C", (), dict(mesh=m, bcs=setA, dt=case.dt, rho=case.rho, mu=case.mu, f=case.f))())
    F, J = assembled(ctx)
    F0, J0 = fresh(setA)
    assert np.array_equal(F, F0) and np.array_equal(J.data, J0.data)
    # replace the set
    ctx.clear_dirichlet()
    for fld, nodes, vals in setB:
        ctx.add_dirichlet(fld, nodes, vals)
    F, J = assembled(ctx)
    F0, J0 = fresh(setB)
    assert np.array_equal(F, F0) and np.array_equal(J.data, J0.data)
    # append an object without clearing
    ctx.add_dirichlet(*extra)
    F, J = assembled(ctx)
    F0, J0 = fresh(setB + [extra])
    assert np.array_equal(F, F0) and np.array_equal(J.data, J0.data)
    # values only (same nodes), as a time-dependent inlet does every step
    setB2 = [(setB[0][0], setB[0][1], 2.0 * setB[0][2]), setB[1]]
    ctx.clear_dirichlet()
    for fld, nodes, vals in setB2:
        ctx.add_dirichlet(fld, nodes, vals)
    F, J = assembled(ctx)
    F0, J0 = fresh(setB2)
    assert np.array_equal(F, F0) and np.array_equal(J.data, J0.data)
    O = make_oracle(type("C", (), dict(mesh=m, bcs=setB2, dt=case.dt, rho=case.rho, mu=case.mu, f=case.f))())
    O.set_un(un)
    Fo = O.assemble(x)
    assert np.abs(F - Fo).max() <= 1e-13 * np.abs(Fo).max()
    # cfdh_update_dirichlet: new values on the SAME dof sets without clear / add.  The first object of setB3 shares dofs with
    # `extra`, which was added later and therefore determines their value: the caller passes only the dofs the object owns.
    ctx.clear_dirichlet()
    later = (0, setB2[0][1][3:12].copy(), rng.standard_normal((9, 2)))  # a later object on nine dofs of the first one
    setB3 = setB2 + [later]
    for fld, nodes, vals in setB3:
        ctx.add_dirichlet(fld, nodes, vals)
    new_vals = -0.5 * setB3[0][2]
    own = ~np.isin(setB3[0][1], later[1])
    assert own.any() and not own.all()
    ctx.update_dirichlet(0, setB3[0][1][own], new_vals[own])
    F, J = assembled(ctx)
    F0, J0 = fresh([(0, setB3[0][1], new_vals), setB3[1], later])
    assert np.array_equal(F, F0) and np.array_equal(J.data, J0.data)
    free = np.setdiff1d(np.arange(nv, dtype=np.int32), np.concatenate([b[1] for b in setB3 if b[0] == 0]))[:3]
    with pytest.raises(Exception):
        ctx.update_dirichlet(0, free, np.zeros((3, 2)))   # not constrained: refused
    ctx.close()


# ---------------------------------------------------------------- full-size properties
@pytest.fixture(scope="module")
def big():
    case = dfg_case(200)  # BASELINE configs[2]: 336k vertices, 1.0M DOF
    ctx = make_ctx(case)
    yield case, ctx
    ctx.close()


def test_full_size_jacobian_consistency(big):
    """At ~1M DOF: J v equals the central difference of the device residual (size-independent
    property; rel 1e-6), and SpMV is linear to round-off."""
    case, ctx = big
    nv = case.nv
    assert 3 * nv > 1_000_000
    rng = np.random.default_rng(0)
    u = np.zeros((nv, 2))
    u[:, 0] = 4 * 0.3 * case.mesh.x[:, 1] * (0.41 - case.mesh.x[:, 1]) / 0.41**2
    u += 1e-3 * rng.uniform(-1, 1, u.shape)  # SURVEY.md 8d micro-benchmark state
    p = 1e-2 * rng.standard_normal(nv)
    for fld, nodes, vals in case.bcs:  # satisfy the Dirichlet data so that F is differentiable in x
        if fld == 0:
            u[nodes] = vals
        else:
            p[nodes] = vals
    un = u.ravel().copy()
    v = rng.standard_normal(3 * nv)
    isbc = np.zeros(3 * nv, bool)
    for fld, nodes, _ in case.bcs:
        if fld == 0:
            isbc[2 * nodes] = isbc[2 * nodes + 1] = True
        else:
            isbc[2 * nv + nodes] = True
    v[isbc] = 0.0
    x0 = np.concatenate([u.ravel(), p])
    ctx.set_state(u_prev=un, p_prev=np.zeros(nv), u=x0[: 2 * nv], p=x0[2 * nv:])
    ctx.assemble(True)
    Jv = ctx.spmv(v)
    eps = 1e-6

    def F_at(x):
        ctx.set_state(u=x[: 2 * nv], p=x[2 * nv:])
        ctx.assemble(False)
        return np.concatenate(ctx.get_residual())

    fd = (F_at(x0 + eps * v) - F_at(x0 - eps * v)) / (2 * eps)
    assert np.linalg.norm(fd[~isbc] - Jv[~isbc]) <= 1e-6 * np.linalg.norm(Jv[~isbc])
    w = rng.standard_normal(3 * nv)
    ctx.set_state(u=x0[: 2 * nv], p=x0[2 * nv:])
    ctx.assemble(True)
    lhs = ctx.spmv(2.0 * v - 3.0 * w)
    rhs = 2.0 * ctx.spmv(v) - 3.0 * ctx.spmv(w)
    assert np.abs(lhs - rhs).max() <= 1e-12 * np.abs(rhs).max()


def test_full_size_steps_converge_and_are_consistent(big):
    """Two steps from rest at ~1M DOF with default and with tight tolerances agree to the solver
    noise level; Newton converges in a handful of iterations."""
    case, ctx = big
    nv = case.nv
    z2, z1 = np.zeros(2 * nv), np.zeros(nv)
    res = []
    for tight in (False, True):
        o = ctx.default_options()
        if tight:
            o.snes_rtol, o.ksp_rtol, o.snes_stol = 1e-11, 1e-9, 0.0
        ctx.set_options(o)
        ctx.set_state(u_prev=z2, p_prev=z1, u=z2, p=z1)
        for _ in range(2):
            st = ctx.solve_step()
            assert st.reason > 0 and st.newton_its <= 6
            ctx.advance()
        res.append((ctx.functional(0, 5), ctx.functional(2), st.fnorm))
    assert abs(res[0][0] - res[1][0]) <= 1e-5 * abs(res[1][0])
    assert abs(res[0][1] - res[1][1]) <= 1e-7 * res[1][1]
    assert res[1][2] < 1e-8  # |F| after the tight solve, from |F0| ~ 3e2


def test_lid_cavity_config_c2():
    """BASELINE configs[1]: lid_driven2D P1/P1 nx=288 -> 250 563 DOF, singular pressure.  Size-independent
    checks: Newton/Krylov converge, the constant-pressure mode stays projected out, both preconditioners
    (SELFP/Chebyshev and Cahouet-Chabard/AMG) reach the same step solution."""
    case = lid_case(288, dt=0.01, mu=0.01)
    nv = case.nv
    assert 3 * nv == 250563
    sols = []
    for pc_type in (1, 0):
        ctx = make_ctx(case)
        o = ctx.default_options()
        o.pc_type = pc_type
        o.snes_rtol, o.ksp_rtol = 1e-10, 1e-8
        ctx.set_options(o)
        z2, z1 = np.zeros(2 * nv), np.zeros(nv)
        ctx.set_state(u_prev=z2, p_prev=z1, u=z2, p=z1)
        for _ in range(2):
            st = ctx.solve_step()
            assert st.reason > 0 and st.newton_its <= 6
            ctx.advance()
        u, p = ctx.get_solution()
        assert abs(p.mean()) <= 1e-10 * np.abs(p).max()
        sols.append((u, p, st.krylov_its))
        ctx.close()
    assert np.linalg.norm(sols[0][0] - sols[1][0]) <= 1e-7 * np.linalg.norm(sols[1][0])
    assert np.linalg.norm(sols[0][1] - sols[1][1]) <= 1e-6 * np.linalg.norm(sols[1][1])


def test_wall_shear_stress_device_matches_oracle():
    """cfdh_wall_shear_stress (the per-step assemble_wss of solverBase.py:163-195 on the device) against the C
    oracle's restatement (orc_wss), after a few scenario steps; the host restatement kept in SolverBase agrees too."""
    from cfd_hemodynamic_amd.scenarios.dfg_1 import DFG1Benchmark
    from cfd_hemodynamic_amd.solverBase import SolverBase
    sc = DFG1Benchmark("stabilized_schur", 0.01, 0.035, m=10, quiet=True)
    sc.solve(None, device_resident=True)
    dev = np.array(sc.solver.shear_stress.x.array, copy=True)
    case = dfg_case(10)
    O = make_oracle(case)
    x = np.concatenate([np.asarray(sc.solver.u_sol.x.array), np.asarray(sc.solver.p_sol.x.array)])
    ref = O.wall_shear_stress(x)
    assert np.abs(ref).max() > 1e-4
    assert np.abs(dev - ref).max() <= 1e-13 * np.abs(ref).max()
    SolverBase.assemble_wss(sc.solver)
    host = np.array(sc.solver.shear_stress.x.array, copy=True)
    assert np.abs(host - ref).max() <= 1e-13 * np.abs(ref).max()
    # zero away from the boundary
    interior = np.ones(sc.mesh.num_vertices, bool)
    interior[np.unique(sc.mesh.facet_vertices)] = False
    assert not dev.reshape(-1, 2)[interior].any()


def test_taylor_green_exact_solution_scenario(monkeypatch):
    """Analytic scenario (2-D counterpart of the reference's taylor_green.py): time-dependent Dirichlet data
    for u AND p from the exact solution, error log of the harness.  The device path must reproduce the
    error history of the oracle-driven harness and stay at the discretisation level."""
    import sys
    import types

    import oracle_solver
    from cfd_hemodynamic_amd.scenarios.taylor_green import TaylorGreenSimulation
    mod = types.ModuleType("cfd_hemodynamic_amd.solvers._oracle_double")
    mod.Solver = oracle_solver.Solver
    monkeypatch.setitem(sys.modules, "cfd_hemodynamic_amd.solvers._oracle_double", mod)
    kw = dict(nx=32, quiet=True, options=dict(snes_rtol=1e-11, snes_stol=0.0, ksp_rtol=1e-9))
    g = TaylorGreenSimulation("stabilized_schur", 0.002, 0.04, **kw)
    g.solve(None, device_resident=True)
    o = TaylorGreenSimulation("_oracle_double", 0.002, 0.04, nx=32, quiet=True,
                              options=dict(snes_rtol=1e-11, snes_stol=0.0, ksp_rtol=1e-11))
    o.solve(None)
    eg, eo = np.array(g.errors), np.array(o.errors)
    assert eg.shape == eo.shape == (21, 2) and eg[0, 1] == 0.0
    assert np.abs(eg[:, 1] - eo[:, 1]).max() < 1e-7
    assert eg[-1, 1] < 2.5e-3          # measured 1.92e-3 (dominated by the one-step lag of the boundary data)
    assert np.abs(g.solver.u_sol.x.array - o.solver.u_sol.x.array).max() < 1e-8
    # halving dt nearly halves the error (first order: the callback updates the Dirichlet data one step late,
    # SURVEY.md Appendix B 6), refining the mesh at fixed dt does not make it worse
    c = TaylorGreenSimulation("stabilized_schur", 0.004, 0.04, nx=32, quiet=True)
    c.solve(None)
    assert 1.4 < c.errors[-1][1] / eg[-1, 1] < 2.2
    b = TaylorGreenSimulation("stabilized_schur_bdf2", 0.002, 0.04, nx=32, quiet=True)
    b.solve(None)
    assert b.errors[-1][1] < 4e-3


def test_stalled_coarsening_is_closed_by_smoothing_not_a_dense_inverse():
    """When no connection is strong (here forced with amg_theta close to 1; in partitioned runs it happens on
    mass-dominated coarse levels) the hierarchy ends on a large level.  That level must be closed with smoothing
    sweeps: a dense inverse of a 10^4-row operator would hang the setup.  The solve then either converges or
    reports the reference's non-convergence error -- quickly."""
    import time
    case = dfg_case(48)
    assert case.nv > 2500
    ctx = make_ctx(case)
    o = ctx.default_options()
    o.amg_theta, o.ksp_max_it = 0.95, 60
    ctx.set_options(o)
    z2, z1 = np.zeros(2 * case.nv), np.zeros(case.nv)
    ctx.set_state(u_prev=z2, p_prev=z1, u=z2, p=z1)
    t0 = time.time()
    try:
        st = ctx.solve_step()
        assert st.reason > 0
    except RuntimeError as e:
        assert "Did not converge" in str(e)
    assert time.time() - t0 < 30.0
    assert ctx.info(6) == 1  # a one-level "hierarchy"
    ctx.close()


def test_fused_amg_cycle_is_the_same_preconditioner(monkeypatch):
    """The fused V-cycle (one kernel per level and direction on the composite operators G = R(I - A W),
    Sb = 2W - W A W, Sc = (I - W A)P, dense coarse solve folded into the level above) is the same linear map as the
    sweep-by-sweep Jacobi V(1,1) cycle: identical FGMRES iteration counts step by step, same step solutions."""
    case = dfg_case(64)   # four-level hierarchies, fine level in SELL format
    nv = case.nv
    runs = []
    monkeypatch.setenv("CFDH_AMG_HOST", "1")  # both cycles on the SAME (host-built) hierarchy; the device build makes the composites only
    for nofuse in ("1", "0"):
        monkeypatch.setenv("CFDH_NO_FUSED_AMG", nofuse)
        ctx = make_ctx(case)
        o = ctx.default_options()
        o.snes_rtol, o.ksp_rtol, o.snes_stol = 1e-11, 1e-9, 0.0
        ctx.set_options(o)
        z2, z1 = np.zeros(2 * nv), np.zeros(nv)
        ctx.set_state(u_prev=z2, p_prev=z1, u=z2, p=z1)
        its = []
        for _ in range(3):
            st = ctx.solve_step()
            its.append((st.newton_its, st.krylov_its))
            ctx.advance()
        runs.append((its, np.concatenate(ctx.get_solution()), ctx.info(25), ctx.info(6)))
        ctx.close()
    (its0, x0, f0, lev0), (its1, x1, f1, lev1) = runs
    assert f0 == 0 and f1 == 1 and lev0 == lev1 >= 3
    assert [n for n, _ in its0] == [n for n, _ in its1]
    assert all(abs(a - b) <= 1 for (_, a), (_, b) in zip(its0, its1)), (its0, its1)
    assert np.linalg.norm(x0 - x1) <= 1e-9 * np.linalg.norm(x0)


def _run_steps(case, nsteps=3, tight=True):
    nv = case.nv
    ctx = make_ctx(case)
    o = ctx.default_options()
    if tight:
        o.snes_rtol, o.ksp_rtol, o.snes_stol = 1e-11, 1e-9, 0.0
    ctx.set_options(o)
    z2, z1 = np.zeros(2 * nv), np.zeros(nv)
    ctx.set_state(u_prev=z2, p_prev=z1, u=z2, p=z1)
    its = []
    for _ in range(nsteps):
        st = ctx.solve_step()
        its.append((st.newton_its, st.krylov_its))
        ctx.advance()
    shape = [[ctx.info(b + l) for l in range(8)] for b in (30, 40, 50, 60)]
    out = (its, np.concatenate(ctx.get_solution()), shape, ctx.info(27), ctx.info(25))
    ctx.close()
    return out


def test_device_built_hierarchies_match_the_host_build(monkeypatch):
    """The AMG hierarchies are built on the GPU (csrc/cfdh_amg_dev.hip: hash SpGEMM per wavefront, transposition, composite
    operators, dense coarse inverse, SELL/fp32 formats).  With the host's aggregates handed in (CFDH_AMG_AGG=host) every
    level has the same size and entry count as the host-built hierarchy and FGMRES takes the same iterations; with the
    device aggregation (distance-2 independent set, front priority) the counts stay within 10 %.  Solutions agree to
    solver tolerance in all three."""
    case = dfg_case(64)
    monkeypatch.setenv("CFDH_AMG_HOST", "1")
    its_h, x_h, shape_h, us_h, fused_h = _run_steps(case)
    monkeypatch.setenv("CFDH_AMG_HOST", "0")
    monkeypatch.setenv("CFDH_AMG_AGG", "host")
    its_a, x_a, shape_a, us_a, fused_a = _run_steps(case)
    monkeypatch.delenv("CFDH_AMG_AGG")
    its_d, x_d, shape_d, us_d, fused_d = _run_steps(case)
    assert us_h == 0 and us_a > 0 and us_d > 0 and fused_h == fused_a == fused_d == 1
    assert shape_h[0] == shape_a[0] and shape_h[2] == shape_a[2], (shape_h, shape_a)       # rows per level, both hierarchies
    assert shape_h[1] == shape_a[1] and shape_h[3] == shape_a[3], (shape_h, shape_a)       # entries per level
    assert its_h == its_a, (its_h, its_a)
    assert [n for n, _ in its_h] == [n for n, _ in its_d]
    assert sum(k for _, k in its_d) <= 1.1 * sum(k for _, k in its_h), (its_h, its_d)
    for x in (x_a, x_d):
        assert np.linalg.norm(x - x_h) <= 1e-8 * np.linalg.norm(x_h)
