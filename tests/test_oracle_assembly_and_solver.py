"""C oracle vs NumPy twin on assembled systems, Dirichlet semantics, and full time steps."""
import numpy as np
import pytest

from oracle import orc
from util import dfg_case, lid_case, load_golden, make_oracle, make_twin


@pytest.mark.parametrize("case_fn,arg", [(dfg_case, 6), (lid_case, 8)])
def test_assembly_matches_twin(case_fn, arg):
    case = case_fn(arg)
    O, pb = make_oracle(case), make_twin(case)
    nv = case.nv
    rng = np.random.default_rng(5)
    xv = 0.1 * rng.standard_normal(3 * nv)
    un = 0.1 * rng.standard_normal((nv, 2))
    O.set_un(un)
    F = O.assemble(xv)
    J = O.csr()
    F2, J2 = pb.assemble(xv, un)
    assert np.abs(F - F2).max() <= 1e-13 * np.abs(F2).max()
    assert abs(J - J2).max() <= 1e-13 * abs(J2).max()


def test_dirichlet_semantics():
    """assemble_vector_block(..., x0=x, alpha=-1) / assemble_matrix_block (stabilized_schur.py:144-175):
    bc rows/cols zero, diagonal = number of bc objects holding the dof, F_bc = x - g,
    other rows lifted by J[:,bc] (g - x)."""
    case = dfg_case(6)
    O, pb = make_oracle(case), make_twin(case)
    nv = case.nv
    rng = np.random.default_rng(6)
    xv = 0.1 * rng.standard_normal(3 * nv)
    un = 0.1 * rng.standard_normal((nv, 2))
    O.set_un(un)
    F = O.assemble(xv)
    J = O.csr().toarray()
    isbc, g, mult = pb.isbc, pb.bcval, pb.bcmult
    assert np.allclose(F[isbc], (xv - g)[isbc], rtol=0, atol=0)
    Jb = J[isbc]
    assert np.array_equal(np.diag(J)[isbc], mult[isbc])
    off = Jb.copy()
    off[np.arange(isbc.sum()), np.nonzero(isbc)[0]] = 0
    assert np.abs(off).max() == 0.0 and np.abs(J[:, isbc][~isbc]).max() == 0.0
    # inlet/wall corner vertices are held by two DirichletBC objects (dfg_1.py:75: [inflow, obstacle, walls])
    assert mult.max() == 2.0
    # lifting: unconstrained assembly
    F0, J0 = pb.assemble(xv, un, apply_bc=False)
    lift = np.where(isbc, g - xv, 0.0)
    expect = (F0 + J0 @ lift)[~isbc]
    assert np.abs(F[~isbc] - expect).max() <= 1e-12 * np.abs(expect).max()


@pytest.mark.parametrize("name", ["dfg_m6", "lid_n8"])
def test_oracle_reproduces_golden_vectors(name):
    case, g = load_golden(name)
    O = make_oracle(case)
    nv = case.nv
    O.set_un(g["u_prev"])
    F = O.assemble(g["state"])
    assert np.abs(F - g["F"]).max() <= 1e-13 * np.abs(g["F"]).max()
    import scipy.sparse as sp
    J = O.csr()
    Jg = sp.csr_matrix((g["J_data"], g["J_indices"], g["J_indptr"]), shape=J.shape)  # explicit zeros pruned
    assert abs(J - Jg).max() <= 1e-13 * abs(Jg).max()
    # two time steps with the reference's solver configuration, tolerances tightened
    for pc_kind in (0, 1):
        x = np.zeros(3 * nv)
        O.set_un(np.zeros(2 * nv))
        opts = orc.default_opts(snes_rtol=1e-12, snes_stol=0.0, ksp_rtol=1e-10, sub_rtol=1e-8, pc_kind=pc_kind)
        for k in (1, 2):
            x, st = O.solve_step(x, opts)
            assert st.reason > 0
            O.set_un(x[: 2 * nv])
            ref = g["step%d" % k]
            assert np.linalg.norm(x - ref) <= 1e-9 * np.linalg.norm(ref)
        assert np.allclose([O.functional(x, 2), O.functional(x, 3)], g["l2"], rtol=1e-9)
        if "drag_lift" in g:
            dl = [O.functional(x, 0, g["obstacle_facets"]), O.functional(x, 1, g["obstacle_facets"])]
            assert np.allclose(dl, g["drag_lift"], rtol=1e-7, atol=1e-12)


def test_reference_defaults_converge_like_direct_newton():
    """PETSc-default tolerances (snes_rtol 1e-8, ksp_rtol 1e-5): the step solution is within
    ~1e-6 of the exactly converged one -- the solver noise the reference's own output carries."""
    case = dfg_case(8)
    O, pb = make_oracle(case), make_twin(case)
    nv = case.nv
    x = np.zeros(3 * nv)
    xt = np.zeros(3 * nv)
    ut = np.zeros((nv, 2))
    O.set_un(np.zeros(2 * nv))
    for _ in range(3):
        x, st = O.solve_step(x, orc.default_opts())
        O.set_un(x[: 2 * nv])
        xt[2 * nv:] -= xt[2 * nv:].mean()
        xt, _ = pb.newton(xt, ut, rtol=1e-13, atol=1e-13)
        ut = xt[: 2 * nv].reshape(-1, 2).copy()
        assert st.reason > 0 and st.newton_its <= 5
    assert np.linalg.norm(x - xt) <= 1e-5 * np.linalg.norm(xt)


def test_divergence_is_reported():
    case = dfg_case(6)
    O = make_oracle(case)
    O.set_un(np.zeros(2 * case.nv))
    with pytest.raises(RuntimeError, match="Did not converge"):
        O.solve_step(np.zeros(3 * case.nv), orc.default_opts(snes_max_it=1, snes_rtol=1e-14, snes_stol=0.0))


def test_projected_initial_guess_in_the_cpu_port():
    """orc_opts.ksp_guess: the CPU port of cfdh_options.ksp_guess (the bench's cpu_baseline leg runs with the GPU path's setting).
    Off by default -- every other oracle test runs the reference's zero guess; on, the converged steps are the same and the
    later steps need fewer iterations."""
    from oracle import orc
    case = dfg_case(14)
    nv = case.nv
    out = {}
    for guess in (0, 4):
        O = make_oracle(case)
        O.set_un(np.zeros(2 * nv))
        x = np.zeros(3 * nv)
        opts = orc.default_opts(pc_kind=2, ksp_guess=guess, snes_rtol=1e-11, snes_stol=0.0, ksp_rtol=1e-9)
        its = []
        for _ in range(9):
            x, st = O.solve_step(x, opts)
            assert st.reason > 0
            O.set_un(x[: 2 * nv])
            its.append(st.krylov_its)
        out[guess] = (x, its)
    assert orc.default_opts().ksp_guess == 0
    assert out[4][1][0] == out[0][1][0] and sum(out[4][1][4:]) < 0.9 * sum(out[0][1][4:])
    assert np.linalg.norm(out[4][0] - out[0][0]) <= 1e-8 * np.linalg.norm(out[0][0])
