"""The sharded solve on real kernels: 2 and 3 ranks sharing the single test GPU (halo + all-reduce
staged through gloo) reproduce the single-rank solution."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, out, timeout=600, **extra_env):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2", CFDH_HOST_THREADS="2", **extra_env)
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_gpu_rank_worker.py"), out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    try:
        outs = [p.communicate(timeout=timeout)[0].decode() for p in procs]
    except subprocess.TimeoutExpired:
        for p in procs:
            p.kill()
        raise
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o[-3000:])
    return np.load(out)


def test_partitioned_solve_matches_single_rank(tmp_path):
    from cfd_hemodynamic_amd.scenarios.dfg_1 import DFG1Benchmark
    tight = dict(snes_rtol=1e-12, snes_stol=0.0, ksp_rtol=1e-10)
    ref = DFG1Benchmark("stabilized_schur", 0.01, 0.05, m=16, quiet=True, options=tight)
    ref.solve(None)
    u0, p0 = ref.solver.u_sol.x.array.copy(), ref.solver.p_sol.x.array.copy()
    for world in (2, 3):
        r = _run(world, str(tmp_path / ("w%d.npz" % world)))
        assert int(r["steps"]) == ref.num_steps
        assert np.linalg.norm(r["u"] - u0) <= 1e-9 * np.linalg.norm(u0)
        assert np.linalg.norm(r["p"] - p0) <= 1e-8 * np.linalg.norm(p0)
        assert abs(float(r["drag"]) - ref.drag) <= 1e-8 * abs(ref.drag)
        assert abs(float(r["lift"]) - ref.lift) <= 1e-7 * abs(ref.lift)
        assert abs(float(r["norm_v"]) - ref.norm_v) <= 1e-10 * ref.norm_v


def test_rccl_binding_single_rank():
    """RCCL is bound at run time (dlopen): create the unique id and a 1-rank communicator and run a
    step with the communicator attached (exercises ncclCommInitRank / the in-stream call sites)."""
    from cfd_hemodynamic_amd import _lib
    from util import dfg_case, make_ctx
    case = dfg_case(8)
    nv = case.nv
    ctx = make_ctx(case)
    uid = _lib.rccl_unique_id()
    assert len(uid) == 128 and any(uid)
    ctx.set_halo([], [0], [], [0], [])
    ctx.comm_init_rccl(uid, 0, 1)
    z2, z1 = np.zeros(2 * nv), np.zeros(nv)
    ctx.set_state(u_prev=z2, p_prev=z1, u=z2, p=z1)
    st = ctx.solve_step()
    assert st.reason > 0
    ctx.close()


def test_partitioned_bdf2_matches_single_rank(tmp_path):
    """The BDF2 variant through the same partition / halo plan (u_prev2 travels with the shift on each rank)."""
    from cfd_hemodynamic_amd.scenarios.dfg_1 import DFG1Benchmark
    tight = dict(snes_rtol=1e-12, snes_stol=0.0, ksp_rtol=1e-10)
    ref = DFG1Benchmark("stabilized_schur_bdf2", 0.01, 0.05, m=16, quiet=True, options=tight)
    ref.solve(None)
    u0, p0 = ref.solver.u_sol.x.array.copy(), ref.solver.p_sol.x.array.copy()
    r = _run(2, str(tmp_path / "bdf2.npz"), CFDH_TEST_SOLVER="stabilized_schur_bdf2")
    assert int(r["steps"]) == ref.num_steps
    assert np.linalg.norm(r["u"] - u0) <= 1e-9 * np.linalg.norm(u0)
    assert np.linalg.norm(r["p"] - p0) <= 1e-8 * np.linalg.norm(p0)
    assert abs(float(r["drag"]) - ref.drag) <= 1e-8 * abs(ref.drag)


def test_rccl_failure_falls_back_to_host_exchange(tmp_path):
    """Two ranks on ONE GPU cannot form an RCCL communicator (duplicate device): the init error must be
    caught on every rank and the job must continue, correctly, on the host-staged exchange."""
    from cfd_hemodynamic_amd.scenarios.dfg_1 import DFG1Benchmark
    tight = dict(snes_rtol=1e-12, snes_stol=0.0, ksp_rtol=1e-10)
    ref = DFG1Benchmark("stabilized_schur", 0.01, 0.05, m=16, quiet=True, options=tight)
    ref.solve(None)
    r = _run(2, str(tmp_path / "fb.npz"), timeout=240, CFDH_TEST_BACKEND="rccl")
    assert str(r["backend"]) == "host" and "RCCL" in str(r["fallback"])
    assert int(r["rccl_attached"]) == 0 and int(r["allgather"]) == 0
    assert np.linalg.norm(r["u"] - ref.solver.u_sol.x.array) <= 1e-9 * np.linalg.norm(ref.solver.u_sol.x.array)


def test_rccl_code_path_with_shared_memory_stand_in(tmp_path):
    """The RCCL branch of cfdh_comm.cpp (ncclCommInitRank, grouped ncclSend/ncclRecv halo with its counts and
    offsets, in-stream ncclAllReduce, enum values) with 2 and 3 ranks: CFDH_RCCL_LIB points the library at
    tests/fake_rccl (same entry points, data through POSIX shared memory), so the code that runs on the
    multi-GPU node is the code tested here; only the transport underneath the NCCL API differs."""
    from cfd_hemodynamic_amd.scenarios.dfg_1 import DFG1Benchmark
    fake = os.path.join(HERE, "fake_rccl", "libfake_rccl.so")
    if not os.path.exists(fake):
        subprocess.check_call(["make", "-C", os.path.join(HERE, "fake_rccl"), "-s"])
    tight = dict(snes_rtol=1e-12, snes_stol=0.0, ksp_rtol=1e-10)
    ref = DFG1Benchmark("stabilized_schur", 0.01, 0.05, m=16, quiet=True, options=tight)
    ref.solve(None)
    ref_krylov = sum(st.krylov_its for _, st in ref.step_stats)
    u0, p0 = ref.solver.u_sol.x.array.copy(), ref.solver.p_sol.x.array.copy()
    for world in (2, 3):
        r = _run(world, str(tmp_path / ("rccl%d.npz" % world)), timeout=300, CFDH_TEST_BACKEND="rccl", CFDH_RCCL_LIB=fake)
        assert str(r["backend"]) == "rccl" and str(r["fallback"]) in ("", "None")
        # the NCCL-API communicator is attached and the pressure right-hand side travels by (padded) all-gather
        nv_global = len(u0) // 2
        assert int(r["rccl_attached"]) == 1 and -(-nv_global // world) <= int(r["allgather"]) <= nv_global // world + 2
        assert int(r["steps"]) == ref.num_steps
        assert np.linalg.norm(r["u"] - u0) <= 1e-9 * np.linalg.norm(u0)
        assert np.linalg.norm(r["p"] - p0) <= 1e-8 * np.linalg.norm(p0)
        assert abs(float(r["drag"]) - ref.drag) <= 1e-8 * abs(ref.drag)
        # overlapping velocity cycle; pressure cycle with the finest level distributed and the coarse levels replicated
        assert int(r["ras"]) == 1 and int(r["dist_coarse"]) > 0
        # ... so the iteration count stays close to one rank's
        assert int(r["krylov"]) <= 1.5 * ref_krylov


def test_partitioned_lid_cavity_and_backflow_stenosis(tmp_path):
    """The other boundary situations of a partitioned run, 3 ranks through the RCCL stand-in: a singular pressure
    (lid cavity: the constant is projected out on every rank consistently) and the do-nothing outlet of the backflow
    variant (no pressure Dirichlet set; the preconditioner's Laplacian takes its Dirichlet rows from the outflow
    vertices, globally)."""
    from cfd_hemodynamic_amd.scenarios.lid_driven2D import LidDriven2DSimulation
    from cfd_hemodynamic_amd.scenarios.stenosis import StenosisSimulation
    fake = os.path.join(HERE, "fake_rccl", "libfake_rccl.so")
    if not os.path.exists(fake):
        subprocess.check_call(["make", "-C", os.path.join(HERE, "fake_rccl"), "-s"])
    # (the singular cavity system does not reach ksp_rtol 1e-10: its right-hand side has a tiny component outside the
    # range of the Jacobian)
    tight = dict(snes_rtol=1e-11, snes_stol=0.0, ksp_rtol=1e-9)
    refs = {
        "lid": LidDriven2DSimulation("stabilized_schur", 0.01, 0.035, nx=48, mu=0.01, quiet=True, options=tight),
        "stenosis_backflow": StenosisSimulation("stabilized_schur_backflow", 0.01, 0.035, ny=12, L=30.0, x_sten=10.0, v_max=60.0,
                                                quiet=True, beta_backflow=0.2, options=tight),
    }
    for case, ref in refs.items():
        ref.solve(None)
        r = _run(3, str(tmp_path / (case + ".npz")), timeout=300, CFDH_TEST_BACKEND="rccl", CFDH_RCCL_LIB=fake, CFDH_TEST_CASE=case,
                 CFDH_TEST_SNES_RTOL="1e-11", CFDH_TEST_KSP_RTOL="1e-9")
        u0, p0, p = ref.solver.u_sol.x.array, ref.solver.p_sol.x.array, r["p"]
        if case == "lid":
            p0, p = p0 - p0.mean(), p - p.mean()
        assert int(r["steps"]) == ref.num_steps
        assert np.linalg.norm(r["u"] - u0) <= 1e-8 * np.linalg.norm(u0), case
        assert np.linalg.norm(p - p0) <= 1e-7 * np.linalg.norm(p0), case
        assert int(r["krylov"]) <= 1.5 * sum(st.krylov_its for _, st in ref.step_stats), case


def test_part_without_exterior_facets_takes_part_in_facet_functionals(tmp_path):
    """Collective discipline: rank 1 owns an island in the interior of the cavity, so its part has no exterior facet.
    Drag/lift-type functionals must still run their reduction on that rank (skipping it would leave the other
    rank blocked in the all-reduce, or pair it with the next unrelated collective)."""
    from cfd_hemodynamic_amd.scenarios.lid_driven2D import LidDriven2DSimulation
    fake = os.path.join(HERE, "fake_rccl", "libfake_rccl.so")
    if not os.path.exists(fake):
        subprocess.check_call(["make", "-C", os.path.join(HERE, "fake_rccl"), "-s"])
    # singular system (no pressure condition): the tolerances of the other singular-cavity partition test.  (Round 2 ran this
    # case at 1e-10 / 1e-8 after a 1000-iteration stall at 1e-11 / 1e-9 on an earlier commit; at HEAD the stall does not
    # reproduce -- tools/mr_island.sh 1e-11 1e-9: 17 FGMRES iterations per Newton step on both ranks.)
    tight = dict(snes_rtol=1e-11, snes_stol=0.0, ksp_rtol=1e-9)
    ref = LidDriven2DSimulation("stabilized_schur", 0.01, 0.035, nx=48, mu=0.01, quiet=True, options=tight)
    ref.solve(None)
    fd, fl = ref.solver.functional(0, 0), ref.solver.functional(1, 0)
    for backend, extra in (("host", {}), ("rccl", {"CFDH_RCCL_LIB": fake})):
        out = str(tmp_path / ("island_%s.npz" % backend))
        r = _run(2, out, timeout=300, CFDH_TEST_BACKEND=backend, CFDH_TEST_CASE="lid", CFDH_TEST_PARTITION="interior_island",
                 CFDH_TEST_SNES_RTOL="1e-11", CFDH_TEST_KSP_RTOL="1e-9", **extra)
        assert int(np.load(out + ".nfac1.npy")[0]) == 0 and int(np.load(out + ".nfac0.npy")[0]) > 0
        assert abs(float(r["fd_all"]) - fd) <= 1e-7 * abs(fd) + 1e-12
        assert abs(float(r["fl_all"]) - fl) <= 1e-7 * abs(fl) + 1e-12
        assert np.linalg.norm(r["u"] - ref.solver.u_sol.x.array) <= 1e-8 * np.linalg.norm(ref.solver.u_sol.x.array)


def test_partitioned_run_with_output_folder_writes_each_file_once(tmp_path):
    """`mesh.comm` of a partitioned run carries the real rank: only rank 0 creates the folder and writes the VTU
    series, norms.txt, final.npz and drag_lift.txt, while the collective field gathers behind `x.array` run on all
    ranks (a rank-0-only gather would deadlock; N ranks writing the same paths would race)."""
    from cfd_hemodynamic_amd.io import read_vtu
    from cfd_hemodynamic_amd.scenarios.dfg_1 import DFG1Benchmark
    tight = dict(snes_rtol=1e-12, snes_stol=0.0, ksp_rtol=1e-10)
    ref = DFG1Benchmark("stabilized_schur", 0.01, 0.05, m=16, quiet=True, options=tight)
    ref.solve(None)
    outdir = tmp_path / "run"
    r = _run(2, str(tmp_path / "o.npz"), timeout=300, CFDH_TEST_OUTDIR=str(outdir))
    names = sorted(os.listdir(outdir))
    for f in ("norms.txt", "final.npz", "drag_lift.txt", "v.pvd", "p.pvd", "wss.pvd"):
        assert f in names
    assert sum(n.startswith("v_") and n.endswith(".vtu") for n in names) == ref.num_steps + 1
    fin = np.load(outdir / "final.npz")
    assert np.linalg.norm(fin["velocity"] - ref.solver.u_sol.x.array) <= 1e-9 * np.linalg.norm(ref.solver.u_sol.x.array)
    last = read_vtu(str(outdir / ("v_%06d.vtu" % ref.num_steps)))
    assert np.allclose(last["v"][:, :2].ravel(), fin["velocity"], rtol=0, atol=1e-14)
    assert abs(float(open(outdir / "drag_lift.txt").read().split()[1]) - ref.drag) <= 1e-8 * abs(ref.drag)
    # communication per FGMRES iteration of this configuration (host-staged backend, 2 ranks)
    ar, halo, sync, its = (int(v) for v in r["counters"][:4])
    assert its > 0 and ar / its < 8 and halo / its < 8


@pytest.mark.parametrize("world", [4, 5])
def test_rccl_path_at_quarter_million_dof_with_4_and_5_ranks(tmp_path, world):
    """DFG mesh m=100 (84 k vertices, 252 k DOF) through the RCCL code path (shared-memory stand-in) with 4 and 5 ranks
    -- the test box admits six processes on its GPU and the test process itself is one; the 8-rank run is the driver's.  Solution equal to
    one rank's to 1e-9, FGMRES iterations within 1.3x, distributed finest pressure level and overlapping velocity
    cycle in use, and the communication of one Krylov iteration counted."""
    from cfd_hemodynamic_amd.scenarios.dfg_1 import DFG1Benchmark
    fake = os.path.join(HERE, "fake_rccl", "libfake_rccl.so")
    if not os.path.exists(fake):
        subprocess.check_call(["make", "-C", os.path.join(HERE, "fake_rccl"), "-s"])
    tight = dict(snes_rtol=1e-12, snes_stol=0.0, ksp_rtol=1e-10)
    ref = DFG1Benchmark("stabilized_schur", 0.01, 0.025, m=100, quiet=True, options=tight)
    assert 3 * ref.mesh.num_vertices > 250000
    ref.solve(None)
    ref_krylov = sum(st.krylov_its for _, st in ref.step_stats)
    u0, p0 = np.asarray(ref.solver.u_sol.x.array).copy(), np.asarray(ref.solver.p_sol.x.array).copy()
    r = _run(world, str(tmp_path / ("big%d.npz" % world)), timeout=600, CFDH_TEST_BACKEND="rccl", CFDH_RCCL_LIB=fake,
             CFDH_TEST_M="100", CFDH_TEST_T="0.025")
    assert str(r["backend"]) == "rccl" and int(r["rccl_attached"]) == 1
    assert int(r["steps"]) == ref.num_steps == 3
    assert np.linalg.norm(r["u"] - u0) <= 1e-9 * np.linalg.norm(u0)
    assert np.linalg.norm(r["p"] - p0) <= 1e-8 * np.linalg.norm(p0)
    assert abs(float(r["drag"]) - ref.drag) <= 1e-8 * abs(ref.drag)
    assert int(r["ras"]) == 1 and int(r["dist_coarse"]) > 0
    assert int(r["krylov"]) <= 1.3 * ref_krylov, (int(r["krylov"]), ref_krylov)
    ar, halo, sync, its, ag, size = (int(v) for v in r["counters"])
    assert size == world and its == int(r["krylov"])
    # per FGMRES iteration (round 4): 3 halo exchanges (iterate before J z, z_p, overlap residual -- the ghost layer of the
    # pressure cycle's right-hand side is no longer exchanged) and 3 all-reduces here (coarse pressure rhs, Gram-Schmidt
    # coefficients, and -- because at ksp_rtol 1e-10 every iteration takes the second Gram-Schmidt pass -- its coefficients and
    # norm in ONE reduction; 2 at the reference's tolerance), plus the per-solve and per-Newton-step reductions
    assert 2.9 <= halo / its <= 3.6 and 2.9 <= ar / its <= 3.6 and sync / its <= 3.2, (halo / its, ar / its, sync / its)


@pytest.mark.parametrize("case,world,backend", [("q1_pipe", 2, "host"), ("q1_pipe", 3, "rccl"), ("p2_dfg", 2, "rccl"), ("p2_dfg", 3, "host"),
                                                ("q1_hex", 3, "rccl"), ("q1_hex", 2, "host"), ("p2_tet", 2, "rccl"), ("p2_tet", 3, "host")])
def test_generic_elements_partitioned_over_ranks(tmp_path, case, world, backend):
    """Round 4: P2/P2 triangles / tetrahedra and Q1/Q1 quadrilaterals / hexahedra in a partitioned run -- the NODE mesh is partitioned like a
    vertex mesh (owned nodes, every cell touching one, ghost nodes; cfdh_create_elem_part), halo exchange and reductions as for P1,
    the replicated global pressure space assembled from the element's own stiffness.  Solution equal to one rank's to 1e-9 / 1e-8,
    both transports (host-staged, RCCL code path through the shared-memory stand-in)."""
    tight = dict(snes_rtol=1e-12, snes_stol=0.0, ksp_rtol=1e-10)
    if case == "q1_pipe":
        from cfd_hemodynamic_amd.scenarios.unit_square_pipe import UnitSquarePipeSimulation
        ref = UnitSquarePipeSimulation("stabilized_schur", 0.01, 0.035, p_inlet=7.47, p_outlet=0.0, nx=96, ny=10, L=14.0, quiet=True, options=tight)
    elif case in ("q1_hex", "p2_tet"):
        from cfd_hemodynamic_amd.scenarios.unit_cube_pipe import UnitCubePipeSimulation
        kw = dict(nx=24, ny=4, nz=4, L=9.0) if case == "q1_hex" else dict(nx=10, ny=2, nz=2, L=7.5, cell_type="tetrahedron", p_grade=2)
        ref = UnitCubePipeSimulation("stabilized_schur", 0.01, 0.035, p_inlet=4.0, p_outlet=0.0, quiet=True, options=tight, **kw)
    else:
        from cfd_hemodynamic_amd.scenarios.dfg_1 import DFG1Benchmark
        ref = DFG1Benchmark("stabilized_schur_backflow", 0.01, 0.035, m=10, quiet=True, v_max=0.3, p_grade=2, beta_backflow=0.2, options=tight)
    ref.solve(None)
    ref_krylov = sum(st.krylov_its for _, st in ref.step_stats)
    u0, p0 = np.asarray(ref.solver.u_sol.x.array).copy(), np.asarray(ref.solver.p_sol.x.array).copy()
    env = dict(CFDH_TEST_CASE=case, CFDH_TEST_BACKEND=backend)
    if backend == "rccl":
        fake = os.path.join(HERE, "fake_rccl", "libfake_rccl.so")
        if not os.path.exists(fake):
            subprocess.check_call(["make", "-C", os.path.join(HERE, "fake_rccl"), "-s"])
        env["CFDH_RCCL_LIB"] = fake
    r = _run(world, str(tmp_path / ("%s_%d.npz" % (case, world))), timeout=600, **env)
    assert str(r["backend"]) == backend and int(r["steps"]) == ref.num_steps
    assert np.linalg.norm(r["u"] - u0) <= 1e-9 * np.linalg.norm(u0)
    assert np.linalg.norm(r["p"] - p0) <= 1e-8 * np.linalg.norm(p0)
    assert abs(float(r["norm_v"]) - ref.norm_v) <= 1e-9 * ref.norm_v
    assert int(r["krylov"]) <= 1.6 * ref_krylov, (int(r["krylov"]), ref_krylov)


@pytest.mark.parametrize("case", ["", "bif3d"])
def test_partitioned_preconditioner_built_on_the_device_equals_the_host_assembled_one(tmp_path, case):
    """Round 4: in a partitioned run the level-0 operators of the preconditioner are built where the Jacobian lives -- the ghost rows
    of the overlapping velocity proxy arrive by halo exchanges of device buffers (cfdh_proxy_ras_dev), H drops its ghost columns in
    the kernel, nothing is downloaded.  Same operators as the host assembly of rounds 2-3 (CFDH_PC_HOST_ASSEMBLY=1): same FGMRES
    iteration count, same solution."""
    env = dict(CFDH_TEST_CASE=case) if case else {}
    a = _run(3, str(tmp_path / "dev.npz"), **env)
    b = _run(3, str(tmp_path / "host.npz"), CFDH_PC_HOST_ASSEMBLY="1", **env)
    assert int(a["ras"]) == 1 and int(b["ras"]) == 1
    assert int(a["krylov"]) == int(b["krylov"])
    assert np.linalg.norm(a["u"] - b["u"]) <= 1e-12 * np.linalg.norm(b["u"])
    assert np.linalg.norm(a["p"] - b["p"]) <= 1e-11 * np.linalg.norm(b["p"])


def test_two_layers_of_overlap_do_not_cost_more_iterations_than_one(tmp_path):
    """Round 4: the parts carry two cell layers of overlap by default (parallel.PartComm.make_part; on the bench meshes they bring the
    4-rank iteration count from 1.2 x to 1.07 x one rank's).  Here: the quarter-million-DOF mesh at the reference's
    tolerances with one and with two layers -- same solution up to solver noise, no more iterations with two."""
    fake = os.path.join(HERE, "fake_rccl", "libfake_rccl.so")
    if not os.path.exists(fake):
        subprocess.check_call(["make", "-C", os.path.join(HERE, "fake_rccl"), "-s"])
    env = dict(CFDH_TEST_BACKEND="rccl", CFDH_RCCL_LIB=fake, CFDH_TEST_M="100", CFDH_TEST_T="0.055", CFDH_TEST_SNES_RTOL="1e-8",
               CFDH_TEST_KSP_RTOL="1e-5")
    one = _run(4, str(tmp_path / "l1.npz"), timeout=600, CFDH_OVERLAP_LAYERS="1", **env)
    two = _run(4, str(tmp_path / "l2.npz"), timeout=600, CFDH_OVERLAP_LAYERS="2", **env)
    assert np.linalg.norm(two["u"] - one["u"]) <= 1e-4 * np.linalg.norm(one["u"])
    assert int(two["krylov"]) <= int(one["krylov"]), (int(two["krylov"]), int(one["krylov"]))


def test_communication_per_iteration_at_the_reference_tolerances(tmp_path):
    """The same 4-rank run at PETSc-default tolerances (what the timed loops run): iterations are launched ahead of the host's
    bookkeeping and nothing takes the second Gram-Schmidt pass, so one FGMRES iteration costs 3 halo exchanges, 2 all-reduces
    (coarse pressure right-hand side; Gram-Schmidt coefficients) and a fraction of a host synchronisation, plus the per-solve and
    per-Newton-step reductions; iteration count within 1.3x of one rank's."""
    from cfd_hemodynamic_amd.scenarios.dfg_1 import DFG1Benchmark
    fake = os.path.join(HERE, "fake_rccl", "libfake_rccl.so")
    if not os.path.exists(fake):
        subprocess.check_call(["make", "-C", os.path.join(HERE, "fake_rccl"), "-s"])
    ref = DFG1Benchmark("stabilized_schur", 0.01, 0.055, m=100, quiet=True)
    ref.solve(None)
    ref_krylov = sum(st.krylov_its for _, st in ref.step_stats)
    r = _run(4, str(tmp_path / "dflt.npz"), timeout=600, CFDH_TEST_BACKEND="rccl", CFDH_RCCL_LIB=fake, CFDH_TEST_M="100", CFDH_TEST_T="0.055",
             CFDH_TEST_SNES_RTOL="1e-8", CFDH_TEST_KSP_RTOL="1e-5")
    assert str(r["backend"]) == "rccl" and int(r["steps"]) == ref.num_steps == 6
    assert int(r["krylov"]) <= 1.3 * ref_krylov, (int(r["krylov"]), ref_krylov)
    u0 = np.asarray(ref.solver.u_sol.x.array)
    assert np.linalg.norm(r["u"] - u0) <= 1e-4 * np.linalg.norm(u0)   # both sides stop at rtol 1e-5 / 1e-8: solver noise
    ar, halo, sync, its, ag, size = (int(v) for v in r["counters"])
    assert 2.9 <= halo / its <= 3.9 and 1.9 <= ar / its <= 3.2 and sync / its <= 1.2, (halo / its, ar / its, sync / its)


def test_config4_stenosis_partitioned_over_4_ranks_at_full_size(tmp_path):
    """BASELINE configs[3]: stenosis "moderate" ~2 M DOF, element partition over 4 ranks with RCCL halo + dot all-reduce --
    here 4 ranks on the one test GPU through the RCCL stand-in (same library code).  Two steps; solution equal to the
    single-rank run, iteration count close to it."""
    from cfd_hemodynamic_amd.scenarios.stenosis import StenosisSimulation
    fake = os.path.join(HERE, "fake_rccl", "libfake_rccl.so")
    if not os.path.exists(fake):
        subprocess.check_call(["make", "-C", os.path.join(HERE, "fake_rccl"), "-s"])
    tight = dict(snes_rtol=1e-11, snes_stol=0.0, ksp_rtol=1e-9)
    ref = StenosisSimulation("stabilized_schur", 0.01, 0.015, grade="moderate", ny=115, v_max=100.0, quiet=True, options=tight)
    assert 3 * ref.mesh.num_vertices == 2036496
    ref.solve(None)
    ref_krylov = sum(st.krylov_its for _, st in ref.step_stats)
    u0, p0 = np.asarray(ref.solver.u_sol.x.array).copy(), np.asarray(ref.solver.p_sol.x.array).copy()
    del ref
    r = _run(4, str(tmp_path / "c4.npz"), timeout=900, CFDH_TEST_BACKEND="rccl", CFDH_RCCL_LIB=fake, CFDH_TEST_CASE="stenosis_c4",
             CFDH_TEST_SNES_RTOL="1e-11", CFDH_TEST_KSP_RTOL="1e-9")
    assert int(r["steps"]) == 2 and str(r["backend"]) == "rccl"
    assert np.linalg.norm(r["u"] - u0) <= 1e-8 * np.linalg.norm(u0)
    assert np.linalg.norm(r["p"] - p0) <= 1e-7 * np.linalg.norm(p0)
    assert int(r["ras"]) == 1 and int(r["dist_coarse"]) > 0
    assert int(r["krylov"]) <= 1.5 * ref_krylov, (int(r["krylov"]), ref_krylov)


def test_config5_tree_partitioned_over_4_ranks(tmp_path):
    """BASELINE configs[4] partitioned: stenosis + 3-generation vascular tree (cut-cell mesh, eight `p = 0` outlets;
    stenosis_with_tree.py:114-142), pulsatile inlet re-sent through `bc.update()` every step, dt = 0.001 -- four ranks through the
    RCCL stand-in on a 60 k-vertex mesh of the domain.  The partition cuts through the tree (a rank may hold several outlets or none);
    solution, outlet flow rates and the inlet data of the last step equal the single-rank run's."""
    from cfd_hemodynamic_amd.scenarios.stenosis_with_tree import StenosisWithTreeSimulation
    fake = os.path.join(HERE, "fake_rccl", "libfake_rccl.so")
    if not os.path.exists(fake):
        subprocess.check_call(["make", "-C", os.path.join(HERE, "fake_rccl"), "-s"])
    tight = dict(snes_rtol=1e-11, snes_stol=0.0, ksp_rtol=1e-9)
    ref = StenosisWithTreeSimulation("stabilized_schur", 0.001, 0.0035, grade="moderate", res=5e-5, pulse_amplitude=0.5, ramp_time=0.005,
                                     inlet_max_velocity=0.05, quiet=True, options=tight)
    assert ref.mesh.num_vertices > 60000 and len(ref.mesh.outlet_caps) == 8
    ref.solve(None)
    ref_krylov = sum(st.krylov_its for _, st in ref.step_stats)
    u0, p0 = np.asarray(ref.solver.u_sol.x.array).copy(), np.asarray(ref.solver.p_sol.x.array).copy()
    q0 = ref.outlet_flow_rates()
    r = _run(4, str(tmp_path / "c5.npz"), timeout=600, CFDH_TEST_BACKEND="rccl", CFDH_RCCL_LIB=fake, CFDH_TEST_CASE="tree_c5",
             CFDH_TEST_SNES_RTOL="1e-11", CFDH_TEST_KSP_RTOL="1e-9")
    assert int(r["steps"]) == ref.num_steps == 4 and str(r["backend"]) == "rccl" and int(r["rccl_attached"]) == 1
    assert np.linalg.norm(r["u"] - u0) <= 1e-8 * np.linalg.norm(u0)
    assert np.linalg.norm(r["p"] - p0) <= 1e-7 * np.linalg.norm(p0)
    assert np.abs(r["outlet_flows"] - q0).max() <= 1e-6 * np.abs(q0).max() and (q0 > 0).all()
    assert abs(float(r["inlet_peak"]) - float(np.abs(np.asarray(ref._u_inlet.x.array)).max())) <= 1e-14
    assert int(r["ras"]) == 1 and int(r["dist_coarse"]) > 0
    assert int(r["krylov"]) <= 1.5 * ref_krylov, (int(r["krylov"]), ref_krylov)


@pytest.mark.parametrize("world", [2, 4])
def test_tetrahedra_partitioned_over_ranks(tmp_path, world):
    """The 3-D path on more than one rank (BASELINE config 5b, simple_bifurcation.py:77-133): element partition of the
    tetrahedra, halo records of four doubles (u_x, u_y, u_z, p), overlapping velocity cycle with three components,
    distributed finest level of the replicated pressure hierarchy -- through the RCCL code path (shared-memory stand-in).
    Solution, L2 norms and the three boundary fluxes equal the single-rank run's."""
    from cfd_hemodynamic_amd.scenarios.simple_bifurcation import MicrovasculatureSimulation
    fake = os.path.join(HERE, "fake_rccl", "libfake_rccl.so")
    if not os.path.exists(fake):
        subprocess.check_call(["make", "-C", os.path.join(HERE, "fake_rccl"), "-s"])
    tight = dict(snes_rtol=1e-11, snes_stol=0.0, ksp_rtol=1e-9, remove_p_mean=0)
    ref = MicrovasculatureSimulation("stabilized_schur", 0.01, 0.025, res=8e-4, quiet=True, options=tight)
    ref.solve(None)
    ref_krylov = sum(st.krylov_its for _, st in ref.step_stats)
    u0, p0 = np.asarray(ref.solver.u_sol.x.array).copy(), np.asarray(ref.solver.p_sol.x.array).copy()
    q0 = np.array(ref.flow_rates())
    r = _run(world, str(tmp_path / ("bif%d.npz" % world)), timeout=600, CFDH_TEST_BACKEND="rccl", CFDH_RCCL_LIB=fake, CFDH_TEST_CASE="bif3d",
             CFDH_TEST_SNES_RTOL="1e-11", CFDH_TEST_KSP_RTOL="1e-9")
    assert int(r["steps"]) == ref.num_steps == 3 and str(r["backend"]) == "rccl" and int(r["rccl_attached"]) == 1
    assert np.linalg.norm(r["u"] - u0) <= 1e-8 * np.linalg.norm(u0)
    assert np.linalg.norm(r["p"] - p0) <= 1e-7 * np.linalg.norm(p0)
    # norms of fields that agree to 1e-8 / 1e-7: the same bounds (the runs stop at snes_rtol on different Krylov paths)
    assert abs(float(r["norm_v"]) - ref.norm_v) <= 1e-8 * ref.norm_v and abs(float(r["norm_p"]) - ref.norm_p) <= 1e-7 * ref.norm_p
    assert np.abs(r["flows"] - q0).max() <= 1e-7 * np.abs(q0).max()
    assert int(r["ras"]) == 1 and int(r["dist_coarse"]) > 0
    assert int(r["krylov"]) <= 1.5 * ref_krylov, (int(r["krylov"]), ref_krylov)
