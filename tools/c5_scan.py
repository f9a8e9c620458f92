"""Config 5 (stenosis + tree) on the device: Newton / FGMRES iteration counts and step times over mesh sizes."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cfd_hemodynamic_amd.scenarios.stenosis_with_tree import StenosisWithTreeSimulation

dt = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-3
vmax = float(sys.argv[3]) if len(sys.argv) > 3 else 1.5
nsteps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
opts = dict(ksp_max_it=3000)
ramp = 0.0
for kv in sys.argv[5:]:
    k, v = kv.split("=")
    if k == "ramp":
        ramp = float(v)
        continue
    opts[k] = float(v) if "." in v or "e" in v else int(v)
for res in [float(v) for v in sys.argv[1].split(",")]:
    t0 = time.time()
    sc = StenosisWithTreeSimulation(os.environ.get("C5_SOLVER", "stabilized_schur"), dt, 1.0, grade="moderate", res=res, pulse_amplitude=0.5, quiet=True,
                                    inlet_max_velocity=vmax, options=opts, ramp_time=ramp)
    print("res %g: %d vertices, %d DOF, setup %.1fs" % (res, sc.mesh.num_vertices, 3 * sc.mesh.num_vertices, time.time() - t0), flush=True)
    for k in range(nsteps):
        sc.set_inlet_time((k + 1) * dt)
        t0 = time.time()
        try:
            sc.solver.solveStep()
            sc.solver.advance()
        except RuntimeError as e:
            print("   step %d FAILED: %s" % (k, e), flush=True)
            break
        st = sc.solver.last_stats
        q = sc.outlet_flow_rates()
        print("   step %d: newton %d krylov %d |F| %.2e  %.0f ms (pc setup %.0f ms)  qout/qin %.4f" % (
            k, st.newton_its, st.krylov_its, st.fnorm, 1e3 * (time.time() - t0), st.ms_pc_setup,
            q.sum() / (2.0 / 3.0 * vmax * 0.003 * sc.inlet_factor((k + 1) * dt))), flush=True)
    del sc
