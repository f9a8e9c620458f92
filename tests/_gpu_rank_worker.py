"""Worker of tests/test_gpu_multirank.py: one rank of an N-process job whose ranks all use GPU 0
(the test box has one GPU), exchanging halos and dot products through the host-staged gloo path.
Same kernels, same partition/halo plan and the same Krylov code as the RCCL path; only the
transport of the two exchanges differs."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from cfd_hemodynamic_amd.parallel import PartComm  # noqa: E402
from cfd_hemodynamic_amd.scenarios.dfg_1 import DFG1Benchmark  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    out = sys.argv[1]
    dist.init_process_group("gloo", rank=rank, world_size=world)
    partitioner = None
    if os.environ.get("CFDH_TEST_PARTITION") == "interior_island":
        # rank 1 owns a disc well inside the domain: its part (owned vertices + one layer of cells) has no exterior facet
        def partitioner(x, nparts):
            c = 0.5 * (x.min(axis=0) + x.max(axis=0))
            r = 0.2 * (x.max(axis=0) - x.min(axis=0)).min()
            return (np.linalg.norm(x - c, axis=1) < r).astype(np.int32)
    comm = PartComm(rank, world, os.environ.get("CFDH_TEST_BACKEND", "host"), partitioner=partitioner)
    tight = dict(snes_rtol=float(os.environ.get("CFDH_TEST_SNES_RTOL", "1e-12")), snes_stol=0.0,
                 ksp_rtol=float(os.environ.get("CFDH_TEST_KSP_RTOL", "1e-10")))
    case = os.environ.get("CFDH_TEST_CASE", "dfg")
    if case == "lid":       # singular pressure (no pressure condition)
        from cfd_hemodynamic_amd.scenarios.lid_driven2D import LidDriven2DSimulation
        sc = LidDriven2DSimulation("stabilized_schur", 0.01, 0.035, nx=48, mu=0.01, quiet=True, device=0, comm=comm, options=tight,
                                   verbose=int(os.environ.get("CFDH_TEST_VERBOSE", "0")))
    elif case == "stenosis_c4":  # BASELINE config 4 at size: stenosis "moderate", reference geometry
        from cfd_hemodynamic_amd.scenarios.stenosis import StenosisSimulation
        sc = StenosisSimulation("stabilized_schur", 0.01, float(os.environ.get("CFDH_TEST_T", "0.015")), grade="moderate",
                                ny=int(os.environ.get("CFDH_TEST_NY", "115")), v_max=100.0, quiet=True, device=0, comm=comm, options=tight)
    elif case == "tree_c5":  # BASELINE config 5: stenosis + vascular tree (cut-cell mesh, eight p = 0 outlets), pulsatile inlet, dt = 0.001
        from cfd_hemodynamic_amd.scenarios.stenosis_with_tree import StenosisWithTreeSimulation
        sc = StenosisWithTreeSimulation("stabilized_schur", 0.001, float(os.environ.get("CFDH_TEST_T", "0.0035")), grade="moderate",
                                        res=float(os.environ.get("CFDH_TEST_RES", "5e-5")), pulse_amplitude=0.5, ramp_time=0.005,
                                        inlet_max_velocity=0.05, quiet=True, device=0, comm=comm, options=tight)
    elif case == "bif3d":  # tetrahedra: the 3-D bifurcation (BASELINE config 5b), inlet parabola, two p = 0 outlets
        from cfd_hemodynamic_amd.scenarios.simple_bifurcation import MicrovasculatureSimulation
        tight["remove_p_mean"] = 0
        sc = MicrovasculatureSimulation("stabilized_schur", 0.01, float(os.environ.get("CFDH_TEST_T", "0.025")),
                                        res=float(os.environ.get("CFDH_TEST_RES", "8e-4")), quiet=True, device=0, comm=comm, options=tight)
    elif case == "stenosis_backflow":  # do-nothing outlet, backflow facet term, no pressure Dirichlet set
        from cfd_hemodynamic_amd.scenarios.stenosis import StenosisSimulation
        sc = StenosisSimulation("stabilized_schur_backflow", 0.01, 0.035, ny=12, L=30.0, x_sten=10.0, v_max=60.0, quiet=True,
                                beta_backflow=0.2, device=0, comm=comm, options=tight)
    elif case == "q1_pipe":  # quadrilateral cells (Q1/Q1), pressure-driven channel, pressure Dirichlet at both ends
        from cfd_hemodynamic_amd.scenarios.unit_square_pipe import UnitSquarePipeSimulation
        sc = UnitSquarePipeSimulation("stabilized_schur", 0.01, 0.035, p_inlet=7.47, p_outlet=0.0, nx=96, ny=10, L=14.0, quiet=True, device=0, comm=comm,
                                      options=tight)
    elif case == "p2_dfg":  # `--solver stabilized_schur_backflow --p_grade 2` (P2/P2), do-nothing outlet + backflow term
        sc = DFG1Benchmark("stabilized_schur_backflow", 0.01, 0.035, m=int(os.environ.get("CFDH_TEST_M", "10")), quiet=True, v_max=0.3, p_grade=2,
                           beta_backflow=0.2, device=0, comm=comm, options=tight)
    elif case in ("q1_hex", "p2_tet"):  # SURVEY 8f-4 in 3-D: hexahedral cells (Q1/Q1) / P2/P2 tetrahedra on the reference's duct
        from cfd_hemodynamic_amd.scenarios.unit_cube_pipe import UnitCubePipeSimulation
        kw = dict(nx=24, ny=4, nz=4, L=9.0) if case == "q1_hex" else dict(nx=10, ny=2, nz=2, L=7.5, cell_type="tetrahedron", p_grade=2)
        sc = UnitCubePipeSimulation("stabilized_schur", 0.01, 0.035, p_inlet=4.0, p_outlet=0.0, quiet=True, device=0, comm=comm, options=tight, **kw)
    else:
        sc = DFG1Benchmark(os.environ.get("CFDH_TEST_SOLVER", "stabilized_schur"), 0.01, float(os.environ.get("CFDH_TEST_T", "0.05")),
                           m=int(os.environ.get("CFDH_TEST_M", "16")), quiet=True, device=0, comm=comm, options=tight)
    outdir = os.environ.get("CFDH_TEST_OUTDIR") or None
    ctx = sc.solver.ctx
    ctx.profile_reset()  # zero the communication counters after setup
    sc.solve(outdir)
    counters = [ctx.info(k) for k in (13, 14, 15, 16, 17, 18)]
    # facet functionals are collective: a rank whose part has no exterior facet must still take part
    fd_all, fl_all = (sc.solver.functional(0, 0), sc.solver.functional(1, 0)) if sc.mesh.geometry.dim == 2 else (0.0, 0.0)
    nfac_local = len(sc.solver._part.facet_cells)
    u = sc.solver.u_sol.x.array.copy()   # gathers the owned slices of every rank
    p = sc.solver.p_sol.x.array.copy()
    extra = {}
    if case == "bif3d":
        extra["flows"] = np.array(sc.flow_rates())
    if case == "tree_c5":
        extra["outlet_flows"] = sc.outlet_flow_rates()
        extra["inlet_peak"] = float(np.abs(np.asarray(sc._u_inlet.x.array)).max())
    if rank == 0:
        np.savez(out, u=u, p=p, **extra, drag=getattr(sc, "drag", 0.0), lift=getattr(sc, "lift", 0.0), norm_v=sc.norm_v, norm_p=sc.norm_p, steps=sc.num_steps,
                 krylov=sum(s.krylov_its for _, s in sc.step_stats), backend=comm.backend,
                 allgather=sc.solver.ctx.info(9), rccl_attached=sc.solver.ctx.info(10),
                 dist_coarse=sc.solver.ctx.info(11), ras=sc.solver.ctx.info(12),
                 fallback=str(getattr(comm, "fallback_reason", "")), counters=counters, fd_all=fd_all, fl_all=fl_all)
    np.save(out + ".nfac%d.npy" % rank, np.array([nfac_local]))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
