"""P2/P2 in a regime where the stabilisation parameter is NOT dominated by its viscous limit h^2 / (4 nu): DFG 2D-1 (nu = 1e-3) with
`--solver stabilized_schur_backflow --p_grade 2 --v_max 0.3`: steps survived, iterations per step.
  python tools/p2_dfg_run.py [m=60] [steps=100]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cfd_hemodynamic_amd.scenarios.dfg_1 import DFG1Benchmark
m = int(sys.argv[1]) if len(sys.argv) > 1 else 60
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dt = float(sys.argv[3]) if len(sys.argv) > 3 else 0.01
sc = DFG1Benchmark("stabilized_schur_backflow", dt, 1.0, m=m, quiet=True, v_max=0.3, p_grade=2, beta_backflow=0.2, verbose=int(os.environ.get("VERBOSE", "0")))
s = sc.solver
log = []
for k in range(steps):
    try:
        s.solveStep(); s.advance()
    except Exception as e:
        print("FAILED at step", k + 1, str(e)[:120]); break
    log.append((s.last_stats.newton_its, s.last_stats.krylov_its))
print("dfg P2 m", m, "nodes", sc.solver.V.mesh.num_vertices, "steps", len(log), log)
print("drag-like functionals:", s.functional(2), s.functional(3))
