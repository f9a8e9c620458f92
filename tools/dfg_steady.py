"""DFG 2D-1 run to T with the shipped scenario on the GPU: C_D, C_L, dp against the benchmark's reference
values (Schaefer & Turek 1996 / featflow: C_D = 5.57953523384, C_L = 0.010618948146, dp = 0.11752016697)."""
import sys, time
sys.path.insert(0, '.')
from cfd_hemodynamic_amd.scenarios.dfg_1 import DFG1Benchmark
m = int(sys.argv[1]) if len(sys.argv) > 1 else 200
T = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
solver = sys.argv[3] if len(sys.argv) > 3 else "stabilized_schur"
t0 = time.time()
sc = DFG1Benchmark(solver, 0.01, T, m=m, quiet=True)
sc.early_stop_tolerance = 1e-7
sc.solve(None)
print("m %d nv %d solver %s steps %d (early stop %s) wall %.1f s  C_D %.6f  C_L %.6f  dp %.6f  |u|_L2 %.6f" % (
    m, sc.mesh.num_vertices, solver, sc.num_steps, sc.stopped_early, time.time() - t0, sc.drag, sc.lift, sc.p_diff, sc.norm_v), flush=True)
