"""Parity at the sizes the claims are made on: the HIP path against the C oracle (oracle/cfdh_oracle.c) on
BASELINE configs[2] (dfg_1 block mesh m=200: 336 273 vertices, 1 008 819 DOF -- the bench workload) and
configs[1] (lid-driven cavity nx=288: 250 563 DOF).

 * assembly: residual F and every CSR value of the Jacobian at a state that violates the Dirichlet data
   (lifting exercised), SpMV with the assembled matrix -- 1e-13 relative (fp64, fixed summation orders differ);
 * two time steps from rest with both sides converged tightly (snes_rtol 1e-12, ksp_rtol 1e-10; the oracle runs
   pc_kind=2, its port of the Cahouet-Chabard/AMG preconditioner, because the restated ILU(0) configuration does
   not converge at this size): step solution <= 1e-9 relative, drag / lift / |u|_L2 / |p|_L2 <= 1e-6 relative
   (north_star's tolerance; /root/reference/src/scenarios/dfg_1.py:183-211, /root/reference/src/scenario.py:315-324).
"""
import numpy as np
import pytest

from util import dfg_case, lid_case, make_ctx, make_oracle

pytestmark = pytest.mark.gpu


def _threads():
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return max(1, min(n, 16))


@pytest.fixture(scope="module", params=["dfg_m200", "lid_nx288"])
def sized(request):
    if request.param == "dfg_m200":
        case = dfg_case(200)
        assert case.nv == 336273 and 3 * case.nv == 1008819
    else:
        case = lid_case(288, dt=0.01, mu=0.01)
        assert 3 * case.nv == 250563
    O = make_oracle(case)
    O.set_threads(_threads())
    ctx = make_ctx(case)
    yield request.param, case, O, ctx
    ctx.close()


def _state(name, case):
    """SURVEY.md 8d micro-benchmark state on the DFG mesh (inlet profile over the whole channel plus
    U(-1e-3,1e-3) noise), a random field on the cavity; neither satisfies the Dirichlet data."""
    nv = case.nv
    rng = np.random.default_rng(0)
    if name.startswith("dfg"):
        u = np.zeros((nv, 2))
        y = case.mesh.x[:, 1]
        u[:, 0] = 4 * 0.3 * y * (0.41 - y) / 0.41**2
        u += 1e-3 * rng.uniform(-1, 1, u.shape)
        p = 1e-2 * rng.standard_normal(nv)
        un = (u * (1.0 + 1e-2 * rng.standard_normal(u.shape))).ravel()
    else:
        u = 0.1 * rng.standard_normal((nv, 2))
        p = 0.1 * rng.standard_normal(nv)
        un = 0.1 * rng.standard_normal(2 * nv)
    return np.concatenate([u.ravel(), p]), un


def test_assembly_csr_and_spmv_match_oracle_at_size(sized):
    name, case, O, ctx = sized
    nv = case.nv
    xv, un = _state(name, case)
    O.set_un(un)
    F = O.assemble(xv)
    J = O.csr()
    ctx.set_state(u_prev=un, p_prev=np.zeros(nv), u=xv[: 2 * nv], p=xv[2 * nv:])
    ctx.assemble(True)
    Fg = np.concatenate(ctx.get_residual())
    assert np.abs(F - Fg).max() <= 1e-13 * np.abs(F).max()
    Jg = ctx.get_csr()
    assert Jg.nnz == 9 * ctx.info(3)
    J.sort_indices()
    Jg.sort_indices()
    # same pattern up to explicit zeros; compare entry-wise through the difference
    D = (J - Jg).tocsr()
    assert abs(D).max() <= 1e-13 * abs(J).max()
    # row-wise relative check as well (a packing error in a small-valued row must not hide behind the global maximum)
    rowmax = np.maximum(abs(J).max(axis=1).toarray().ravel(), 1e-300)
    drow = abs(D).max(axis=1).toarray().ravel()
    assert (drow / rowmax).max() <= 1e-11
    v = np.random.default_rng(3).standard_normal(3 * nv)
    Jv = J @ v
    assert np.abs(ctx.spmv(v) - Jv).max() <= 1e-13 * np.abs(Jv).max()
    del J, Jg, D


def test_two_tight_steps_match_oracle_at_size(sized):
    from oracle import orc
    name, case, O, ctx = sized
    nv = case.nv
    o = ctx.default_options()
    o.snes_rtol, o.snes_stol, o.ksp_rtol = 1e-12, 0.0, 1e-10
    ctx.set_options(o)
    z2, z1 = np.zeros(2 * nv), np.zeros(nv)
    ctx.set_state(u_prev=z2, p_prev=z1, u=z2, p=z1)
    x = np.zeros(3 * nv)
    O.set_un(z2)
    opts = orc.default_opts(snes_rtol=1e-12, snes_stol=0.0, ksp_rtol=1e-10, pc_kind=2)
    for step in range(2):
        st = ctx.solve_step()
        assert st.reason > 0 and st.newton_its <= 6
        xg = np.concatenate(ctx.get_solution())
        ctx.advance()
        x, so = O.solve_step(x, opts)
        O.set_un(x[: 2 * nv])
        assert np.linalg.norm(xg - x) <= 1e-9 * np.linalg.norm(x), (name, step)
        # velocity and pressure separately (the pressure block is the worse conditioned one)
        assert np.linalg.norm(xg[: 2 * nv] - x[: 2 * nv]) <= 1e-9 * np.linalg.norm(x[: 2 * nv])
        assert np.linalg.norm(xg[2 * nv:] - x[2 * nv:]) <= 1e-8 * np.linalg.norm(x[2 * nv:])
    l2u, l2p = O.functional(x, 2), O.functional(x, 3)
    assert abs(ctx.functional(2) - l2u) <= 1e-9 * l2u
    assert abs(ctx.functional(3) - l2p) <= 1e-8 * l2p
    if "ft" in case.markers:
        obst = case.markers["ft"].find(5)
        cd, cl = O.functional(x, 0, obst), O.functional(x, 1, obst)
        gd, gl = ctx.functional(0, 5), ctx.functional(1, 5)
        assert abs(gd - cd) <= 1e-6 * abs(cd) and abs(gl - cl) <= 1e-6 * abs(cl)
        # measured agreement is far inside north_star's 1e-6: keep the regression bar where it is
        assert abs(gd - cd) <= 1e-8 * abs(cd) and abs(gl - cl) <= 1e-7 * abs(cl) + 1e-14
