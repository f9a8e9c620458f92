"""Loose known-answers from the data the reference ships / cites."""
import os

import numpy as np

from oracle import orc
from util import dfg_case, lid_case, make_oracle

HERE = os.path.dirname(os.path.abspath(__file__))


def _run_to_steady(O, nv, dt, tol, max_steps, opts):
    x = np.zeros(3 * nv)
    O.set_un(np.zeros(2 * nv))
    for i in range(max_steps):
        u_old = x[: 2 * nv].copy()
        x, st = O.solve_step(x, opts)
        O.set_un(x[: 2 * nv])
        rel = np.abs(x[: 2 * nv] - u_old).max() / max(np.abs(x[: 2 * nv]).max(), 1e-12) / dt
        if rel < tol:
            break
    return x, i + 1


def test_lid_cavity_re100_against_ghia():
    """Ghia et al. centre-line u(x=0.5, y) at Re=100 (tests/golden/ghia_re100_u_centerline.csv is the
    reference's src/benchmark_data/lid_driven2D/plot_u_y_Ghia100.csv)."""
    nx = 32
    case = lid_case(nx, dt=0.05, mu=0.01)
    O = make_oracle(case)
    x, n = _run_to_steady(O, case.nv, 0.05, 2e-3, 600, orc.default_opts(pc_kind=1))
    data = np.loadtxt(os.path.join(HERE, "golden", "ghia_re100_u_centerline.csv"), delimiter=",", skiprows=1)
    u = x[: 2 * case.nv].reshape(-1, 2)[:, 0].reshape(nx + 1, nx + 1)
    xs = np.linspace(0, 1, nx + 1)
    col = u[:, nx // 2]
    got = np.interp(data[:, 0], xs, col)
    inner = (data[:, 0] > 0.02) & (data[:, 0] < 0.98)
    assert np.abs(got[inner] - data[inner, 1]).max() < 0.03, np.abs(got - data[:, 1]).max()
    assert got[np.argmin(np.abs(data[:, 0] - 0.4531))] < -0.17  # the vortex-core minimum (-0.2109 in Ghia)


def test_dfg_2d1_drag_lift_order_of_magnitude():
    """DFG 2D-1 literature values C_D = 5.5795, C_L = 0.010619 (external): a coarse
    stabilised P1/P1 mesh lands within ~10 % / a factor 2 after the start-up transient."""
    case = dfg_case(12)
    O = make_oracle(case)
    x, n = _run_to_steady(O, case.nv, 0.01, 5e-3, 600, orc.default_opts(pc_kind=1))
    obst = case.markers["ft"].find(5)
    cd, cl = 500 * O.functional(x, 0, obst), 500 * O.functional(x, 1, obst)
    assert 4.6 < cd < 6.2, cd
    assert 0.0 < cl < 0.03, cl
