import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cfd_hemodynamic_amd.scenarios.stenosis_with_tree import StenosisWithTreeSimulation
res, vmax, dt = float(sys.argv[1]), float(sys.argv[2]), 1e-3
opts = dict(ksp_max_it=int(sys.argv[3]) if len(sys.argv) > 3 else 400)
for kv in sys.argv[4:]:
    k, v = kv.split("=")
    opts[k] = float(v) if "." in v or "e" in v else int(v)
sc = StenosisWithTreeSimulation("stabilized_schur", dt, 1.0, grade="moderate", res=res, pulse_amplitude=0.5, quiet=True,
                                inlet_max_velocity=vmax, options=opts, verbose=2)
for k in range(2):
    sc.set_inlet_time((k + 1) * dt)
    print("=== step", k, flush=True)
    try:
        sc.solver.solveStep(); sc.solver.advance()
    except RuntimeError as e:
        print("FAILED", e); break
    st = sc.solver.last_stats
    print("newton", st.newton_its, "krylov", st.krylov_its, flush=True)
