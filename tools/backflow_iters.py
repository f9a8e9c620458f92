"""Developer tool: iteration counts / timings of the backflow (do-nothing outlet) variant."""
import sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from cfd_hemodynamic_amd.scenarios.stenosis import StenosisSimulation
ny = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for solver, kw in (("stabilized_schur", {}), ("stabilized_schur_backflow", dict(beta_backflow=0.2))):
    sc = StenosisSimulation(solver, 0.01, 1.0, ny=ny, v_max=100.0, quiet=True, **kw)
    S = sc.solver
    t0 = time.perf_counter()
    for s in range(12):
        S.solveStep(); S.advance()
        st = S.last_stats
        if s < 3 or s % 4 == 3:
            print(solver, "step", s, "newton", st.newton_its, "krylov", st.krylov_its, "ms", round(st.ms_total, 2), "refresh", st.pc_refreshes, flush=True)
    print(solver, "nv", sc.mesh.num_vertices, "12 steps", round(time.perf_counter() - t0, 3), "s", flush=True)
