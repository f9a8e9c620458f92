"""Time-scheme generalisation (theta; a0,a1,a2) of the element kernel: the `stabilized_schur_bdf2`
variant (/root/reference/src/solvers/stabilized_schur_bdf2.py:79-110,298-326).
CPU part: twin vs finite differences, C oracle vs twin, algebraic properties.
GPU part (marked): HIP assembly and multi-step BDF2 runs against the oracle through the C-ABI."""
import numpy as np
import pytest

from util import dfg_case, lid_case, make_ctx, make_oracle, make_twin

SCHEMES = [(0.5, 1.0, -1.0, 0.0), (1.0, 1.0, -1.0, 0.0), (1.0, 1.5, -2.0, 0.5)]


def _set(pb, sch):
    pb.prm.theta, pb.prm.a0, pb.prm.a1, pb.prm.a2 = sch


def _rand_state(nv, seed):
    rng = np.random.default_rng(seed)
    return rng.standard_normal(3 * nv), rng.standard_normal((nv, 2)), rng.standard_normal((nv, 2))


@pytest.mark.parametrize("sch", SCHEMES + [(0.7, 1.2, -0.9, -0.3)])
def test_twin_jacobian_is_exact_derivative(sch):
    case = lid_case(3)
    pb = make_twin(case)
    _set(pb, sch)
    nv = case.nv
    xv, un, un2 = _rand_state(nv, 3)
    _, J = pb.assemble(xv, un, apply_bc=False, un2=un2)
    Jd = J.toarray()
    eps = 1e-6
    Jfd = np.zeros_like(Jd)
    for k in range(3 * nv):
        e = np.zeros(3 * nv)
        e[k] = eps
        Fp, _ = pb.assemble(xv + e, un, want_jac=False, apply_bc=False, un2=un2)
        Fm, _ = pb.assemble(xv - e, un, want_jac=False, apply_bc=False, un2=un2)
        Jfd[:, k] = (Fp - Fm) / (2 * eps)
    assert np.abs(Jd - Jfd).max() < 2e-8 * np.abs(Jd).max()


@pytest.mark.parametrize("sch", SCHEMES)
def test_oracle_matches_twin(sch):
    case = dfg_case(6)
    pb, O = make_twin(case), make_oracle(case)
    _set(pb, sch)
    O.set_scheme(*sch)
    xv, un, un2 = _rand_state(case.nv, 5)
    O.set_un(un)
    O.set_un2(un2)
    F, J = pb.assemble(xv, un, un2=un2)
    Fo = O.assemble(xv, True)
    Jo = O.csr()
    assert np.abs(F - Fo).max() < 1e-13 * np.abs(F).max()
    assert abs(J - Jo).max() < 1e-13 * abs(J).max()


def test_schemes_agree_on_a_stationary_history():
    """u = u_prev = u_prev2: the time term vanishes (a0+a1+a2 = 0) and theta drops out, so the
    midpoint and both BDF residuals coincide -- every scheme has the same steady states."""
    case = dfg_case(6)
    pb = make_twin(case)
    xv, _, _ = _rand_state(case.nv, 7)
    u = xv[: 2 * case.nv].reshape(-1, 2)
    Fs = []
    for sch in SCHEMES:
        _set(pb, sch)
        F, _ = pb.assemble(xv, u, want_jac=False, apply_bc=False, un2=u)
        Fs.append(F)
    assert np.abs(Fs[0] - Fs[1]).max() < 1e-12 * np.abs(Fs[0]).max()
    assert np.abs(Fs[0] - Fs[2]).max() < 1e-12 * np.abs(Fs[0]).max()


def _oracle_bdf2_steps(case, nsteps, rtol=1e-12):
    from oracle import orc
    O = make_oracle(case)
    o = orc.default_opts(pc_kind=1)
    o.snes_rtol, o.ksp_rtol = rtol, 1e-12
    nv = case.nv
    x = np.zeros(3 * nv)
    un, un2 = np.zeros(2 * nv), np.zeros(2 * nv)
    out = []
    for s in range(nsteps):
        O.set_scheme(1.0, *((1.0, -1.0, 0.0) if s == 0 else (1.5, -2.0, 0.5)))
        O.set_un(un)
        O.set_un2(un2)
        x, _ = O.solve_step(x, o)
        un2 = un.copy()
        un = x[: 2 * nv].copy()
        out.append(x.copy())
    return out


def test_oracle_bdf2_matches_twin_newton():
    """C oracle (Krylov) vs the twin's direct-factorisation Newton over BDF1 + BDF2 steps."""
    case = dfg_case(6)
    pb = make_twin(case)
    nv = case.nv
    xs = _oracle_bdf2_steps(case, 3)
    x = np.zeros(3 * nv)
    un, un2 = np.zeros((nv, 2)), np.zeros((nv, 2))
    for s in range(3):
        _set(pb, (1.0,) + ((1.0, -1.0, 0.0) if s == 0 else (1.5, -2.0, 0.5)))
        x = pb.newton(x, un, un2=un2)[0]
        un2 = un.copy()
        un = x[: 2 * nv].reshape(-1, 2).copy()
        assert np.abs(x - xs[s]).max() < 1e-9 * max(1.0, np.abs(x).max())


# ------------------------------------------------------------------------------------ GPU


@pytest.mark.gpu
@pytest.mark.parametrize("sch", SCHEMES)
def test_gpu_assembly_matches_oracle(sch):
    case = dfg_case(8)
    O, ctx = make_oracle(case), make_ctx(case)
    O.set_scheme(*sch)
    ctx.set_time_scheme(*sch)
    nv = case.nv
    xv, un, un2 = _rand_state(nv, 11)
    O.set_un(un)
    O.set_un2(un2)
    ctx.set_state(u_prev=un.reshape(-1), p_prev=np.zeros(nv), u=xv[: 2 * nv], p=xv[2 * nv:])
    ctx.set_previous2(un2.reshape(-1))
    assert np.array_equal(ctx.get_previous2(), un2.reshape(-1))
    Fo = O.assemble(xv, True)
    Jo = O.csr()
    ctx.assemble(True)
    ru, rp = ctx.get_residual()
    F = np.concatenate([ru, rp])
    J = ctx.get_csr()
    assert np.abs(F - Fo).max() < 1e-13 * np.abs(Fo).max()
    assert abs(J - Jo).max() < 1e-13 * abs(Jo).max()


@pytest.mark.gpu
def test_gpu_bdf2_solver_matches_oracle():
    """The drop-in `stabilized_schur_bdf2.Solver` under the Scenario loop (BDF1 first step, BDF2 after,
    u_prev2 shifted on the device) against the oracle driven by hand through the same sequence;
    device-resident and literal host-copy loops give the same fields."""
    from cfd_hemodynamic_amd.scenarios.dfg_1 import DFG1Benchmark

    nsteps = 4
    kw = dict(m=8, quiet=True, options=dict(snes_rtol=1e-12, snes_stol=0.0, ksp_rtol=1e-10))
    a = DFG1Benchmark("stabilized_schur_bdf2", 0.01, 0.01 * nsteps - 0.005, **kw)
    a.solve(None, device_resident=True)
    b = DFG1Benchmark("stabilized_schur_bdf2", 0.01, 0.01 * nsteps - 0.005, **kw)
    b.solve(None, device_resident=False)
    assert a.num_steps == b.num_steps == nsteps
    assert a.solver.step_count == nsteps and float(a.solver.bdf_a0.value) == 1.5 and float(a.solver.bdf_a2.value) == 0.5
    assert np.array_equal(a.solver.u_sol.x.array, b.solver.u_sol.x.array)
    case = dfg_case(8)
    xs = _oracle_bdf2_steps(case, nsteps)
    nv = case.nv
    x = np.concatenate([a.solver.u_sol.x.array, a.solver.p_sol.x.array])
    assert np.abs(x - xs[-1]).max() < 1e-8 * np.abs(xs[-1]).max()
    # u_prev2 holds the u_prev the last step used
    assert np.abs(a.solver.u_prev2.x.array - xs[-2][: 2 * nv]).max() < 1e-8
    assert np.abs(b.solver.u_prev2.x.array - xs[-2][: 2 * nv]).max() < 1e-8
    # and the BDF2 trajectory differs from the midpoint one (the scheme really switched)
    c = DFG1Benchmark("stabilized_schur", 0.01, 0.01 * nsteps - 0.005, **kw)
    c.solve(None, device_resident=True)
    assert np.abs(c.solver.u_sol.x.array - a.solver.u_sol.x.array).max() > 1e-4


# ------------------------------------------------- boundary terms: stabilized_schur_backflow
def test_backflow_twin_jacobian_and_oracle():
    """Do-nothing outlet + backflow term (stabilized_schur_backflow.py:107,158-176): twin Jacobian
    vs finite differences (u_prev random: the (.)_- switch is active on part of the outlet only),
    C oracle vs twin."""
    from util import stenosis_backflow_case
    case = stenosis_backflow_case(4, L=6.0, x_sten=3.0, beta=0.7)
    pb, O = make_twin(case), make_oracle(case)
    nv = case.nv
    xv, un, _ = _rand_state(nv, 13)
    out_nodes = np.unique(case.mesh.facet_vertices[case.backflow_facets])
    un[out_nodes, 0] = np.resize([-1.0, 0.5, -0.3, 0.8, -1.2], len(out_nodes))  # sign changes along the outlet
    O.set_un(un)
    F, J = pb.assemble(xv, un)
    Fo = O.assemble(xv, True)
    assert np.abs(F - Fo).max() < 1e-13 * np.abs(F).max()
    assert abs(J - O.csr()).max() < 1e-13 * abs(J).max()
    J = pb.assemble(xv, un, apply_bc=False)[1].copy()
    Jd = J.toarray()
    eps = 1e-6
    for k in np.concatenate([2 * out_nodes, 2 * out_nodes + 1]):
        e = np.zeros(3 * nv)
        e[k] = eps
        Fp, _ = pb.assemble(xv + e, un, want_jac=False, apply_bc=False)
        Fm, _ = pb.assemble(xv - e, un, want_jac=False, apply_bc=False)
        assert np.abs((Fp - Fm) / (2 * eps) - Jd[:, k]).max() < 2e-8 * np.abs(Jd).max()
    # the term is active, dissipative (adds a positive semi-definite boundary mass) and switches off for outflow
    pb.set_boundary_terms(False, None, 0.0)
    _, J0 = pb.assemble(xv, un, apply_bc=False)
    D = (J - J0).toarray()[: 2 * nv, : 2 * nv]
    assert np.abs(D).max() > 0 and np.linalg.eigvalsh(0.5 * (D + D.T)).min() > -1e-14
    pb.set_boundary_terms(False, case.backflow_facets, 0.7)
    un_out = np.tile([1.0, 0.0], (nv, 1))  # u_prev . n > 0 on the outlet (n = +x)
    Fa, _ = pb.assemble(xv, un_out, want_jac=False, apply_bc=False)
    pb.set_boundary_terms(False, None, 0.0)
    Fb, _ = pb.assemble(xv, un_out, want_jac=False, apply_bc=False)
    assert np.array_equal(Fa, Fb)


def test_backflow_quadrature_is_exact_without_sign_change():
    """With u_prev.n < 0 on a whole facet the integrand is a cubic: the 2-point rule must equal the
    closed form  -beta rho |e| sum_b ubar_b int s l_a l_b."""
    from oracle import np_twin as T
    x = np.array([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0]])
    cells = np.array([[0, 1, 2]])
    prm = T.Params(0.1, 2.0, 0.01, ds_terms=False, beta_backflow=0.5)
    rng = np.random.default_rng(0)
    u, un, p = rng.standard_normal((3, 2)), rng.standard_normal((3, 2)), np.zeros(3)
    # facet opposite vertex 1: vertices 2,0 on x=0, outward normal (-1,0); make u_prev.n < 0 there
    un[[0, 2], 0] = [0.3, 1.1]
    ff = np.array([8 << 1], dtype=np.uint8)
    Fe, _ = T.element_tensors(x, cells, u, un, p, prm, ff, want_jac=False)
    prm0 = T.Params(0.1, 2.0, 0.01, ds_terms=False, beta_backflow=0.0)
    Fe0, _ = T.element_tensors(x, cells, u, un, p, prm0, ff, want_jac=False)
    ub = 0.5 * (u + un)
    s = {0: -un[0, 0], 2: -un[2, 0]}
    # int_0^1 l_a l_b l_c over an edge of length 1: 1/4 (a=b=c), 1/12 otherwise
    I = lambda a, b, c: 0.25 if a == b == c else 1.0 / 12.0
    exp = np.zeros((3, 2))
    for a in (0, 2):
        for b in (0, 2):
            for c_ in (0, 2):
                exp[a] -= 0.5 * 2.0 * s[c_] * I(a, b, c_) * ub[b]
    assert np.allclose((Fe - Fe0)[0, :6].reshape(3, 2), exp, rtol=0, atol=1e-14)


@pytest.mark.gpu
def test_gpu_backflow_assembly_and_step_match_oracle():
    from oracle import orc
    from util import stenosis_backflow_case
    case = stenosis_backflow_case(8, L=12.0, x_sten=5.0, beta=0.5)
    O, ctx = make_oracle(case), make_ctx(case)
    nv = case.nv
    xv, un, _ = _rand_state(nv, 17)
    out_nodes = np.unique(case.mesh.facet_vertices[case.backflow_facets])
    un[out_nodes, 0] = np.resize([-1.0, 0.5, -0.3, 0.8, -1.2], len(out_nodes))  # sign changes along the outlet
    O.set_un(un)
    ctx.set_state(u_prev=un.reshape(-1), p_prev=np.zeros(nv), u=xv[: 2 * nv], p=xv[2 * nv:])
    Fo = O.assemble(xv, True)
    Jo = O.csr()
    ctx.assemble(True)
    ru, rp = ctx.get_residual()
    assert np.abs(np.concatenate([ru, rp]) - Fo).max() < 1e-13 * np.abs(Fo).max()
    assert abs(ctx.get_csr() - Jo).max() < 1e-13 * abs(Jo).max()
    # a few steps from rest with reversed flow at the outlet imposed through u_prev
    o = orc.default_opts(pc_kind=1)
    o.snes_rtol, o.snes_stol, o.ksp_rtol = 1e-12, 0.0, 1e-12
    opt = ctx.default_options()
    opt.snes_rtol, opt.snes_stol, opt.ksp_rtol = 1e-12, 0.0, 1e-10
    ctx.set_options(opt)
    x = np.zeros(3 * nv)
    up = np.zeros((nv, 2))
    up[:, 0] = -20.0 * (case.mesh.x[:, 0] / 12.0)  # inflow through the outlet: (u_prev.n)_- != 0
    O.set_un(up)
    ctx.set_state(u_prev=up.reshape(-1), p_prev=np.zeros(nv), u=x[: 2 * nv], p=x[2 * nv:])
    for s in range(3):
        x, _ = O.solve_step(x, o)
        O.set_un(x[: 2 * nv])
        ctx.solve_step()
        u, p = ctx.get_solution()
        ctx.advance()
        assert np.abs(np.concatenate([u, p]) - x).max() < 1e-8 * np.abs(x).max(), s
    ctx.close()


@pytest.mark.gpu
def test_gpu_backflow_solver_class_on_stenosis_scenario():
    """`--simulation stenosis --solver stabilized_schur_backflow --v_max ...`: constructor contract
    (v_max required), bcp ignored, and the scenario run against the oracle loop."""
    from cfd_hemodynamic_amd.scenarios.stenosis import StenosisSimulation
    from oracle import orc
    from util import stenosis_backflow_case
    # the Scenario wraps constructor errors like the reference does (scenario.py:95-103)
    with pytest.raises(RuntimeError, match="ValueError: v_max is required"):
        StenosisSimulation("stabilized_schur_backflow", 0.01, 0.02, ny=8, L=12.0, x_sten=5.0, quiet=True)
    kw = dict(grade="moderate", ny=8, L=12.0, x_sten=5.0, v_max=100.0, quiet=True, beta_backflow=0.2,
              options=dict(snes_rtol=1e-12, snes_stol=0.0, ksp_rtol=1e-10))
    sc = StenosisSimulation("stabilized_schur_backflow", 0.01, 0.035, **kw)
    assert sc.solver.bcp_d == []
    u0 = np.array(sc.solver.u_prev.x.array, dtype=float)  # the flow-rate-conserving initial profile (stenosis.py:219-259)
    sc.solve(None, device_resident=True)
    assert sc.num_steps == 4
    case = stenosis_backflow_case(8, L=12.0, x_sten=5.0, beta=0.2)
    O = make_oracle(case)
    nv = case.nv
    o = orc.default_opts(pc_kind=1)
    o.snes_rtol, o.snes_stol, o.ksp_rtol = 1e-12, 0.0, 1e-12
    x = np.concatenate([u0, np.zeros(nv)])
    O.set_un(u0)
    for _ in range(4):
        x, _ = O.solve_step(x, o)
        O.set_un(x[: 2 * nv])
    xg = np.concatenate([sc.solver.u_sol.x.array, sc.solver.p_sol.x.array])
    assert np.abs(xg - x).max() < 1e-8 * np.abs(x).max()
