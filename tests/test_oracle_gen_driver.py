"""The C Newton / FGMRES / Cahouet-Chabard + AMG driver of oracle/cfdh_oracle.c over the generic element routines
(cfdh_oracle_gen.c, cfdh_oracle_gen3.c): the CPU solver behind `cpu_baseline` and the at-size parity of the P2 / Q1 bench lines
(missing 6).  Two steps of a channel / duct problem against the twin's Newton with a direct sparse solve.  CPU only."""
import numpy as np
import pytest

from gen3_util import ETYPE3, facet_node_set3, node_mesh3, problem3
from gen_util import ETYPE, facet_node_set, node_mesh, problem
from oracle import np_twin as T, np_twin_gen as G, np_twin_gen3 as G3, np_twin_nd as TN, orc, orcg, orcg3


@pytest.fixture(autouse=True)
def _c_element_routines(monkeypatch):
    monkeypatch.setattr(G, "element_tensors", orcg.element_tensors)
    monkeypatch.setattr(G3, "element_tensors", orcg3.element_tensors)


@pytest.mark.parametrize("kind,n", [("P2", 6), ("Q1", 10)])
def test_c_driver_on_2d_generic_elements_matches_the_twin(kind, n):
    m = node_mesh(kind, n)
    nv = m.num_vertices
    prm = T.Params(0.02, 1.0, 0.02, (0.0, 0.0))
    mid = m.facet_midpoints()
    top, bottom = np.nonzero(np.isclose(mid[:, 1], m.x[:, 1].max()))[0], np.nonzero(np.isclose(mid[:, 1], 0.0))[0]
    left, right = np.nonzero(np.isclose(mid[:, 0], 0.0))[0], np.nonzero(np.isclose(mid[:, 0], m.x[:, 0].max()))[0]
    walls = facet_node_set(m, np.concatenate([top, bottom]))
    inl = np.setdiff1d(facet_node_set(m, left), walls)
    y = m.x[inl, 1] / m.x[:, 1].max()
    outn = facet_node_set(m, right)
    pb = problem(kind, m, prm)
    O = orc.Oracle(m.x, m.cells, m.facet_cells, m.facet_local, prm.dt, prm.rho, prm.mu, prm.f, etg=ETYPE[kind])
    for fld, nodes, vals in [(0, walls, np.zeros((len(walls), 2))), (0, inl, np.stack([4 * y * (1 - y), 0 * y], 1)), (1, outn, np.zeros(len(outn)))]:
        (pb.add_bc_u if fld == 0 else pb.add_bc_p)(nodes, vals)
        (O.add_bc_u if fld == 0 else O.add_bc_p)(nodes, vals)
    opts = orc.default_opts(pc_kind=2, snes_rtol=1e-12, snes_stol=0.0, ksp_rtol=1e-10)
    x, xo, un = np.zeros(3 * nv), np.zeros(3 * nv), np.zeros((nv, 2))
    for step in range(2):
        O.set_un(un.ravel())
        xo, st = O.solve_step(xo, opts)
        assert st.reason > 0
        x, _ = pb.newton(x, un)
        un = x[: 2 * nv].reshape(-1, 2).copy()
        assert np.abs(xo[: 2 * nv] - x[: 2 * nv]).max() <= 1e-8 * np.abs(x[: 2 * nv]).max(), (kind, step)
        assert np.abs(xo[2 * nv:] - x[2 * nv:]).max() <= 1e-7 * np.abs(x[2 * nv:]).max(), (kind, step)


@pytest.mark.parametrize("kind,n", [("P2", 2), ("Q1", 3)])
def test_c_driver_on_3d_generic_elements_matches_the_twin(kind, n):
    m = node_mesh3(kind, n)
    nv = m.num_vertices
    prm = TN.Params(0.02, 1.0, 0.02, (0.0, 0.0, 0.0))
    mid = m.facet_midpoints()
    hi = m.x.max(axis=0)
    on = lambda d, v: np.nonzero(np.isclose(mid[:, d], v))[0]  # noqa: E731
    walls = facet_node_set3(m, np.concatenate([on(1, 0.0), on(1, hi[1]), on(2, 0.0), on(2, hi[2])]))
    inl = np.setdiff1d(facet_node_set3(m, on(0, 0.0)), walls)
    y, z = m.x[inl, 1] / hi[1], m.x[inl, 2] / hi[2]
    outn = facet_node_set3(m, on(0, hi[0]))
    pb = problem3(kind, m, prm)
    O = orc.Oracle(m.x, m.cells, m.facet_cells, m.facet_local, prm.dt, prm.rho, prm.mu, prm.f, etg=ETYPE3[kind])
    for fld, nodes, vals in [(0, walls, np.zeros((len(walls), 3))), (0, inl, np.stack([16 * y * (1 - y) * z * (1 - z), 0 * y, 0 * y], 1)), (1, outn, np.zeros(len(outn)))]:
        (pb.add_bc_u if fld == 0 else pb.add_bc_p)(nodes, vals)
        (O.add_bc_u if fld == 0 else O.add_bc_p)(nodes, vals)
    opts = orc.default_opts(pc_kind=2, snes_rtol=1e-12, snes_stol=0.0, ksp_rtol=1e-10)
    x, xo, un = np.zeros(4 * nv), np.zeros(4 * nv), np.zeros((nv, 3))
    for step in range(2):
        O.set_un(un.ravel())
        xo, st = O.solve_step(xo, opts)
        assert st.reason > 0
        x, _ = pb.newton(x, un)
        un = x[: 3 * nv].reshape(-1, 3).copy()
        assert np.abs(xo[: 3 * nv] - x[: 3 * nv]).max() <= 1e-8 * np.abs(x[: 3 * nv]).max(), (kind, step)
        assert np.abs(xo[3 * nv:] - x[3 * nv:]).max() <= 1e-7 * np.abs(x[3 * nv:]).max(), (kind, step)
