"""Where the set-up time of a run goes (mesh generation, context creation, boundary data, first step incl. hierarchy builds)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
t0 = time.perf_counter()
from cfd_hemodynamic_amd.mesh import create_dfg_channel
from cfd_hemodynamic_amd.scenarios.dfg_1 import DFG1Benchmark
m = int(sys.argv[1]) if len(sys.argv) > 1 else 200
t1 = time.perf_counter()
mesh, ft = create_dfg_channel(m)
t2 = time.perf_counter()
print("imports %.2f s, mesh generation %.2f s (%d vertices)" % (t1 - t0, t2 - t1, mesh.num_vertices), flush=True)
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
sc = DFG1Benchmark("stabilized_schur", 0.01, 1.0, m=m, quiet=True)
pr.disable()
t3 = time.perf_counter()
print("scenario construction (mesh again + Solver + setup) %.2f s" % (t3 - t2), flush=True)
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
for k in range(3):
    t = time.perf_counter()
    sc.solver.solveStep(); sc.solver.advance()
    st = sc.solver.last_stats
    print("step %d: %.1f ms (pc setup %.1f ms, assemble %.1f, solve %.1f)" % (k, 1e3 * (time.perf_counter() - t), st.ms_pc_setup, st.ms_assemble, st.ms_solve), flush=True)
