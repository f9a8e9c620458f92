"""3-D bifurcation (simple_bifurcation) on the device over mesh sizes: setup time, Newton / FGMRES counts, step time, flux balance."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfd_hemodynamic_amd.scenarios.simple_bifurcation import MicrovasculatureSimulation
dt = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
for res in [float(v) for v in sys.argv[1].split(",")]:
    t0 = time.time()
    sc = MicrovasculatureSimulation("stabilized_schur", dt, 1.0, res=res, quiet=True)
    nv = sc.mesh.num_vertices
    print("res %g: %d vertices, %d cells, %d DOF, setup %.1f s" % (res, nv, sc.mesh.num_cells, 4 * nv, time.time() - t0), flush=True)
    for k in range(nsteps):
        t0 = time.time()
        sc.solver.solveStep(); sc.solver.advance()
        st = sc.solver.last_stats
        qi, q1, q2 = sc.flow_rates()
        print("   step %d: newton %d krylov %d |F| %.2e  %.0f ms (asm %.1f, solve %.1f, pc setup %.0f)  qout/qin %.4f  q1/q2 %.4f" % (
            k, st.newton_its, st.krylov_its, st.fnorm, 1e3 * (time.time() - t0), st.ms_assemble, st.ms_solve, st.ms_pc_setup, (q1 + q2) / qi, q1 / q2), flush=True)
    del sc
