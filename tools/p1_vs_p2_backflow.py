"""P1 vs P2 on the backflow stenosis at the same node count: FGMRES iterations per step (and pc_type 0 vs 1 for P2)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cfd_hemodynamic_amd.scenarios.stenosis import StenosisSimulation
for pg, ny, opts in [(1, 80, {}), (2, 40, {})] + [(2, 40, eval(a)) for a in sys.argv[1:]]:
    try:
        sc = StenosisSimulation("stabilized_schur_backflow", 0.01, 1.0, ny=ny, v_max=20.0, p_grade=pg, beta_backflow=0.2, quiet=True, options=opts)
        its, nw = [], []
        for k in range(4):
            t0 = time.perf_counter(); sc.solver.solveStep(); sc.solver.advance(); dt = time.perf_counter() - t0
            its.append(sc.solver.last_stats.krylov_its); nw.append(sc.solver.last_stats.newton_its)
        print("p_grade", pg, "ny", ny, opts, "nodes", sc.solver.V.mesh.num_vertices, "its", its, "newton", nw, "last step ms %.1f" % (1e3 * dt))
    except Exception as e:
        print("p_grade", pg, opts, "FAILED", str(e)[:200])
