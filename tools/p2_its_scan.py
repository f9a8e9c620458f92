"""P1 vs P2 FGMRES iterations per Newton step on a SHORT backflow stenosis (L = 16, stenosis at x = 8; the geometry of
tools/p2_schur_study.py, whose exact-sub-solve counts are 13-20) as the mesh is refined -- to see whether the P2 counts of the
product grow with the mesh (hierarchy problem) or are high at every size (operator problem).
Usage: python tools/p2_its_scan.py [ny_p2 ...]   (env knobs CFDH_* apply)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cfd_hemodynamic_amd.scenarios.stenosis import StenosisSimulation  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [6, 12, 24, 48]
for ny2 in sizes:
    for pg, ny in ((1, 2 * ny2), (2, ny2)):
        try:
            sc = StenosisSimulation("stabilized_schur_backflow", 0.01, 1.0, ny=ny, L=16.0, x_sten=8.0, v_max=20.0, p_grade=pg,
                                    beta_backflow=0.2, quiet=True)
            its, nw = [], []
            for k in range(3):
                sc.solver.solveStep(); sc.solver.advance()
                its.append(sc.solver.last_stats.krylov_its); nw.append(sc.solver.last_stats.newton_its)
            c = sc.solver.ctx
            lev = [(c.info(30 + l), c.info(50 + l)) for l in range(6)]
            print("p_grade", pg, "ny", ny, "nodes", sc.solver.V.mesh.num_vertices, "its", its, "newton", nw,
                  "per newton %.1f" % (sum(its) / max(1, sum(nw))), "levels hA/hL rows", lev, flush=True)
        except Exception as e:
            print("p_grade", pg, "ny", ny, "FAILED", str(e)[:200], flush=True)
