import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfd_hemodynamic_amd.scenarios.simple_bifurcation import MicrovasculatureSimulation
res = float(sys.argv[1]); nsteps = int(sys.argv[2])
opts = {}
for kv in sys.argv[3:]:
    k, v = kv.split("="); opts[k] = float(v) if "." in v or "e" in v else int(v)
sc = MicrovasculatureSimulation("stabilized_schur", 0.01, 1.0, res=res, quiet=True, verbose=1, options=opts)
for k in range(nsteps):
    print("=== step", k, flush=True)
    sc.solver.solveStep(); sc.solver.advance()
    qi, q1, q2 = sc.flow_rates()
    print("newton", sc.solver.last_stats.newton_its, "krylov", sc.solver.last_stats.krylov_its, "qout/qin %.4f" % ((q1 + q2) / qi), flush=True)
