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
