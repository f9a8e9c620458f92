"""Iteration counts of the partitioned solve vs number of ranks (all ranks share GPU 0; developer tool)."""
import os, sys, time
import numpy as np
import torch.distributed as dist
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from cfd_hemodynamic_amd.parallel import PartComm
from cfd_hemodynamic_amd.scenarios.dfg_1 import DFG1Benchmark
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
comm = PartComm(rank, world, os.environ.get("CFDH_TEST_BACKEND", "host"))
m = int(sys.argv[1])
t0 = time.time()
sc = DFG1Benchmark("stabilized_schur", 0.01, 1.0, m=m, quiet=True, device=0, comm=comm, verbose=int(os.environ.get("CFDH_VERBOSE", "0")))
print("rank", rank, "setup done", round(time.time() - t0, 1), "s", flush=True)
its = []
for s in range(8):
    sc.solver.solveStep(); sc.solver.advance(); its.append(sc.solver.last_stats.krylov_its)
    if rank == 0:
        print("  step", s, "krylov", its[-1], round(time.time() - t0, 1), "s", flush=True)
if rank == 0:
    print("ranks", world, "m", m, "krylov per step", its, "drag", sc.drag_lift()[0], flush=True)
else:
    sc.drag_lift()
dist.barrier(); dist.destroy_process_group()
