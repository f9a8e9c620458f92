"""Lid-driven cavity nx=288 on the device to steady state; centre-line u against the Ghia tables (tests/golden)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cfd_hemodynamic_amd.scenarios.lid_driven2D import LidDriven2DSimulation
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 288
for Re, dt in ((100, 0.05), (400, 0.05), (1000, 0.05)):
    data = np.loadtxt(os.path.join(ROOT, "tests", "golden", "ghia_re%d_u_centerline.csv" % Re), delimiter=",", skiprows=1)
    sc = LidDriven2DSimulation("stabilized_schur_bdf2", dt, 1e9, nx=nx, mu=1.0 / Re, quiet=True)
    t0 = time.time(); its = 0
    for k in range(4000):
        sc.solver.solveStep()
        its += sc.solver.last_stats.krylov_its
        rel = sc.solver.functional(6) / max(sc.solver.functional(4), 1e-12) / dt
        sc.solver.advance()
        if rel < 1e-4:
            break
    got = sc.centerline_u(data[:, 0])
    err = np.abs(got - data[:, 1])
    print("Re %d: %d steps (t=%.1f), %.1f s, %d krylov its, max |u - ghia| = %.4f at y=%.4f, rel_diff/dt %.2e, u_min %.5f" % (
        Re, k + 1, (k + 1) * dt, time.time() - t0, its, err.max(), data[np.argmax(err), 0], rel, got.min()), flush=True)
    print("   errs:", np.round(err, 4).tolist(), flush=True)
