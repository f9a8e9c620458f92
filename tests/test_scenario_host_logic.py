"""Host logic of the plugin surface (Scenario loop, BoundaryCondition, Function) without a GPU:
the Scenario harness is driven by a test double backed by the CPU oracle."""
import sys
import types

import numpy as np
import pytest

import oracle_solver
from cfd_hemodynamic_amd.boundaryCondition import BoundaryCondition
from cfd_hemodynamic_amd.fem import Function, FunctionSpace
from cfd_hemodynamic_amd.mesh import create_unit_square


def test_function_interpolate_layout():
    m = create_unit_square(2)
    V = FunctionSpace(m, 2)
    f = Function(V)
    f.interpolate(lambda x: np.vstack((x[0] + 10 * x[1], -x[0])))
    a = f.x.array.reshape(-1, 2)
    assert np.allclose(a[:, 0], m.x[:, 0] + 10 * m.x[:, 1]) and np.allclose(a[:, 1], -m.x[:, 0])
    assert V.dofmap.index_map.size_global == m.num_vertices and V.dofmap.index_map_bs == 2


def test_boundary_condition_update_reinterpolates_source():
    m = create_unit_square(2)
    V = FunctionSpace(m, 2)
    src = Function(V)
    bc = BoundaryCondition(src)
    bc.initGeometrical(lambda x: np.isclose(x[0], 0.0))
    d = bc.getBC(V)
    assert set(d.dofs) == {0, 3, 6}
    src.x.array[:] = 7.0
    assert d.g.x.array.max() == 0.0
    d.update()  # boundaryCondition.py:48-51 (here restricted to the constrained blocks: the only values ever read)
    assert (d.g.x.array.reshape(-1, 2)[d.dofs] == 7.0).all()


def test_time_loop_counts_and_early_stop(oracle_backend):
    """scenario.py:243-307: float-accumulated `while t < T`; early-stop test at (i+1)%10==0
    comparing u_sol with the not-yet-updated u_prev (SURVEY.md Appendix B 4,5)."""
    from cfd_hemodynamic_amd.scenarios.lid_driven2D import LidDriven2DSimulation
    sc = LidDriven2DSimulation(oracle_backend, 0.01, 0.1, nx=6, mu=0.1, quiet=True)
    sc.solve(None)
    assert sc.num_steps == 11  # T=0.1, dt=0.01 -> 11 steps by float accumulation
    assert sc.solver.calls == 11
    # reference semantics of the state copy
    assert np.array_equal(sc.solver.u_prev.x.array, sc.solver.u_sol.x.array)
    # a loose tolerance stops at the first check, after step 9
    sc2 = LidDriven2DSimulation(oracle_backend, 0.01, 1.0, nx=6, mu=0.1, quiet=True)
    sc2.early_stop_tolerance = 1e9
    sc2.solve(None)
    assert sc2.num_steps == 9 and sc2.stopped_early


def test_unknown_solver_and_kwarg_filtering(oracle_backend):
    from cfd_hemodynamic_amd.scenarios.lid_driven2D import LidDriven2DSimulation
    # ImportError for an unknown plugin module, listing the ones present (scenario.py:61-72 of the reference)
    with pytest.raises(ImportError, match=r"no solver plugin 'does_not_exist'.*stabilized_schur"):
        LidDriven2DSimulation("does_not_exist", 0.01, 0.1, nx=4)
    sc = LidDriven2DSimulation(oracle_backend, 0.01, 0.02, nx=4, quiet=True, some_unused_kwarg=3)
    assert sc.solver.nv == 25


def test_poiseuille_steady_state(oracle_backend):
    """Analytic known answer of the reference (unit_square.py:100-104): u = (4y(1-y), 0) with
    mu=1: inlet profile, no-slip walls, p=0 outlet; discretisation-limited on P1."""
    from cfd_hemodynamic_amd.mesh import locate_entities_boundary
    from cfd_hemodynamic_amd.scenario import Scenario

    class Channel(Scenario):
        def __init__(self, solver_name):
            self._mesh = create_unit_square(16)
            self.quiet = True
            super().__init__(solver_name, "unit_square", 1.0, 1.0, 0.05, 2.0, (0, 0))
            self.setup()

        mesh = property(lambda self: self._mesh)

        @property
        def bcu(self):
            V = self.solver.V
            ui = Function(V)
            ui.interpolate(lambda x: np.vstack((4 * x[1] * (1 - x[1]), 0 * x[1])))
            b1 = BoundaryCondition(ui)
            b1.initTopological(1, locate_entities_boundary(self.mesh, 1, lambda x: np.isclose(x[0], 0)))
            u0 = Function(V)
            b2 = BoundaryCondition(u0)
            b2.initTopological(1, locate_entities_boundary(self.mesh, 1, lambda x: np.isclose(x[1], 0) | np.isclose(x[1], 1)))
            return [b1, b2]

        @property
        def bcp(self):
            p0 = Function(self.solver.Q)
            b = BoundaryCondition(p0)
            b.initTopological(1, locate_entities_boundary(self.mesh, 1, lambda x: np.isclose(x[0], 1)))
            return [b]

        def initial_velocity(self, x):
            return np.zeros((2, x.shape[1]))

        def exact_velocity(self, t):
            return lambda x: np.vstack((4 * x[1] * (1 - x[1]), 0 * x[1]))

    sc = Channel(oracle_backend)
    sc.early_stop_tolerance = 1e-6
    sc.solve(None)
    u_e = Function(sc.solver.V)
    u_e.interpolate(sc.exact_velocity(0))
    err = sc.compute_error(u_e, sc.solver.u_sol, sc.mesh)
    assert err < 3e-2, err  # stabilised P1/P1 on 16x16, outlet do-nothing terms of stabilized_schur.py:79
    # pressure drop of Poiseuille flow: dp/dx = -8 mu
    p = sc.solver.p_sol.x.array.reshape(17, 17)
    slope = (p[8, 4] - p[8, 12]) / 0.5  # interior, away from the inlet/outlet boundary terms
    assert abs(slope - 8.0) < 0.4, slope


def test_vtu_series_and_ffr_outputs(oracle_backend, tmp_path):
    """Output either side of the path (scenario.py:208-228,258-263; stenosis.py:158-211): the five
    time series are written every step (t=0 included) and read back bit-exactly; FFR = p(L,R)/p(0,R)."""
    from cfd_hemodynamic_amd.io import read_vtu
    from cfd_hemodynamic_amd.scenarios.stenosis import StenosisSimulation
    sc = StenosisSimulation(oracle_backend, 0.01, 0.025, ny=4, L=8.0, x_sten=4.0, v_max=50.0, quiet=True)
    out = str(tmp_path / "run")
    sc.solve(out)
    assert sc.num_steps == 3
    import os
    for name in ("v", "p", "u_residual", "p_residual", "wss"):
        pvd = open(os.path.join(out, name + ".pvd")).read()
        assert pvd.count("<DataSet") == 4  # t = 0 and three steps
        assert os.path.exists(os.path.join(out, "%s_%06d.vtu" % (name, 3)))
    v = read_vtu(os.path.join(out, "v_000003.vtu"))
    assert np.array_equal(v["cells"], sc.mesh.cells)
    assert np.array_equal(v["points"][:, :2], sc.mesh.x)
    assert np.array_equal(v["v"][:, :2], sc.solver.u_sol.x.array.reshape(-1, 2)) and not v["v"][:, 2].any()
    p = read_vtu(os.path.join(out, "p_000003.vtu"))
    assert np.array_equal(p["p"].ravel(), sc.solver.p_sol.x.array)
    w = read_vtu(os.path.join(out, "wss_000003.vtu"))
    assert np.array_equal(w["shear_stress"][:, :2], sc.solver.shear_stress.x.array.reshape(-1, 2))
    # FFR: pressure at the inlet / outlet centre points; p = 0 at the outlet here, so FFR = 0 and p_proximal > 0
    txt = open(os.path.join(out, "ffr.txt")).read()
    assert "FFR = p_distal / p_proximal" in txt
    assert sc.p_proximal > 0 and abs(sc.p_distal) < 1e-12 and sc.ffr == pytest.approx(0.0, abs=1e-12)
    # write_every thins the series, 0 disables it
    sc2 = StenosisSimulation(oracle_backend, 0.01, 0.025, ny=4, L=8.0, x_sten=4.0, v_max=50.0, quiet=True)
    out2 = str(tmp_path / "run2")
    sc2.solve(out2, write_every=2)
    assert open(os.path.join(out2, "v.pvd")).read().count("<DataSet") == 2
    sc3 = StenosisSimulation(oracle_backend, 0.01, 0.025, ny=4, L=8.0, x_sten=4.0, v_max=50.0, quiet=True)
    out3 = str(tmp_path / "run3")
    sc3.solve(out3, write_every=0)
    assert not os.path.exists(os.path.join(out3, "v.pvd")) and os.path.exists(os.path.join(out3, "final.npz"))


def test_p1_point_evaluation():
    m = create_unit_square(5)
    f = 2.0 * m.x[:, 0] - 3.0 * m.x[:, 1] + 0.5  # P1 reproduces affine functions exactly
    pts = [(0.13, 0.77), (1.0, 1.0), (0.0, 0.4), (1.5, 0.5)]
    v = m.eval_p1(f, pts)
    assert np.allclose(v[:3], [2 * x - 3 * y + 0.5 for x, y in pts[:3]], atol=1e-13)
    assert np.isnan(v[3])


def test_dfg_scenario_from_msh_file(oracle_backend, tmp_path):
    """`mesh_file=` (a gmsh .msh with the reference's markers) drives the same scenario as the generator."""
    from cfd_hemodynamic_amd.mesh import create_dfg_channel
    from cfd_hemodynamic_amd.meshio import write_msh
    from cfd_hemodynamic_amd.scenarios.dfg_1 import DFG1Benchmark
    m0, ft0 = create_dfg_channel(6)
    path = str(tmp_path / "dfg6.msh")
    write_msh(path, m0, ft0)
    a = DFG1Benchmark(oracle_backend, 0.01, 0.025, m=6, quiet=True)
    a.solve(None)
    b = DFG1Benchmark(oracle_backend, 0.01, 0.025, mesh_file=path, quiet=True)
    b.solve(None)
    # same mesh, same numbering: equal up to the run-to-run round-off of the threaded oracle reductions
    assert np.array_equal(a.mesh.cells, b.mesh.cells) and np.array_equal(a.mesh.facet_marker, b.mesh.facet_marker)
    assert np.allclose(a.solver.u_sol.x.array, b.solver.u_sol.x.array, rtol=0, atol=1e-10)
    assert abs(a.drag - b.drag) < 1e-8 and abs(a.lift - b.lift) < 1e-8 and abs(a.p_diff - b.p_diff) < 1e-8


def test_epsilon_and_sigma_of_solver_base():
    """solverBase.py:176-182 `epsilon(u) = sym(nabla_grad(u))`, `sigma(u, p, mu) = 2 mu epsilon(u) - p I`: the mirror evaluates them
    per cell of a P1 field (exact for fields that are linear in x), in 2-D and on tetrahedra."""
    from cfd_hemodynamic_amd.mesh3d import create_unit_cube
    from cfd_hemodynamic_amd.solverBase import SolverBase
    A2 = np.array([[0.3, -1.2], [0.7, 0.5]])   # u_j = sum_i x_i A[i, j]  =>  nabla_grad(u) = A
    A3 = np.array([[0.3, -1.2, 0.4], [0.7, 0.5, -0.1], [0.2, 0.9, 1.1]])
    for mesh, A in ((create_unit_square(5), A2), (create_unit_cube(3), A3)):
        d = A.shape[0]
        V, Q = FunctionSpace(mesh, d), FunctionSpace(mesh, 1)
        u, p = Function(V), Function(Q)
        u.x.array[:] = (np.asarray(mesh.x) @ A).ravel()
        p.x.array[:] = 2.0 + np.asarray(mesh.x)[:, 0]
        E = SolverBase.epsilon(u)
        assert E.shape == (len(mesh.cells), d, d)
        assert np.abs(E - 0.5 * (A + A.T)[None]).max() <= 1e-12
        S = SolverBase.sigma(u, p, 0.04)
        pc = 2.0 + np.asarray(mesh.x)[np.asarray(mesh.cells)][:, :, 0].mean(axis=1)
        assert np.abs(S - (0.04 * (A + A.T)[None] - pc[:, None, None] * np.eye(d)[None])).max() <= 1e-12


def test_bench_contract_host_side():
    """bench.py without a GPU: `--gpus N` never silently runs one rank (it refuses when fewer devices are visible than ranks), the
    run length of every config is the reference's T / dt, and the kernel-source fingerprint that ties profiles/pmc_traffic.json to
    a build is a pure function of the sources."""
    import os
    import subprocess
    import types
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=120, env=dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES=""))
    assert r.returncode == 2 and "--gpus 2" in (r.stdout + r.stderr) and not r.stdout.strip().startswith("{")
    for cfg, dt, n in (("c3", 0.01, 100), ("c2", 0.01, 1000), ("c4", 0.01, 100), ("c5", 0.001, 1000), ("c5b", 0.01, 100)):
        assert bench.run_length(types.SimpleNamespace(config=cfg, dt=dt)) == n
    a, b = bench.kernel_source_sha16(), bench.kernel_source_sha16()
    assert a == b and len(a) == 16 and int(a, 16) >= 0
