"""Developer tool: partitioned runs of the other scenarios / solver plugins against their single-rank results
(ranks share GPU 0; run under torch.distributed.run, e.g. with CFDH_TEST_BACKEND=rccl CFDH_RCCL_LIB=tests/fake_rccl/...)."""
import os, sys, time
import numpy as np
import torch.distributed as dist
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from cfd_hemodynamic_amd.parallel import PartComm
from cfd_hemodynamic_amd.scenarios.lid_driven2D import LidDriven2DSimulation
from cfd_hemodynamic_amd.scenarios.stenosis import StenosisSimulation
from cfd_hemodynamic_amd.scenarios.dfg_1 import DFG1Benchmark
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
tight = dict(snes_rtol=1e-11, snes_stol=0.0, ksp_rtol=1e-9)
cases = [
    ("lid nx=64", lambda **k: LidDriven2DSimulation("stabilized_schur", 0.01, 0.045, nx=64, mu=0.01, quiet=True, options=tight, **k)),
    ("lid nx=64 bdf2", lambda **k: LidDriven2DSimulation("stabilized_schur_bdf2", 0.01, 0.045, nx=64, mu=0.01, quiet=True, options=tight, **k)),
    ("stenosis p=0 outlet", lambda **k: StenosisSimulation("stabilized_schur", 0.01, 0.045, ny=16, L=40.0, x_sten=12.0, v_max=60.0, quiet=True, options=tight, **k)),
    ("stenosis backflow", lambda **k: StenosisSimulation("stabilized_schur_backflow", 0.01, 0.045, ny=16, L=40.0, x_sten=12.0, v_max=60.0, quiet=True, beta_backflow=0.2, options=tight, **k)),
    ("dfg m=24 full factor", lambda **k: DFG1Benchmark("stabilized_schur", 0.01, 0.045, m=24, quiet=True, options=dict(tight, schur_full=1), **k)),
]
for name, make in cases:
    ref = None
    if rank == 0:
        s0 = make(); s0.solve(None)
        ref = (s0.solver.u_sol.x.array.copy(), s0.solver.p_sol.x.array.copy(), sum(st.krylov_its for _, st in s0.step_stats))
    comm = PartComm(rank, world, os.environ.get("CFDH_TEST_BACKEND", "host"))
    sc = make(device=0, comm=comm)
    sc.solve(None)
    u, p = sc.solver.u_sol.x.array.copy(), sc.solver.p_sol.x.array.copy()
    its = sum(st.krylov_its for _, st in sc.step_stats)
    if rank == 0:
        pm = ref[1] - ref[1].mean() if "lid" in name else ref[1]
        pq = p - p.mean() if "lid" in name else p
        print("%-24s ranks %d  |du|/|u| %.2e  |dp|/|p| %.2e  krylov %d (1 rank: %d)  ras %d dist %d" % (
            name, world, np.linalg.norm(u - ref[0]) / np.linalg.norm(ref[0]), np.linalg.norm(pq - pm) / max(np.linalg.norm(pm), 1e-300),
            its, ref[2], sc.solver.ctx.info(12), sc.solver.ctx.info(11)), flush=True)
    dist.barrier()
dist.destroy_process_group()
