"""Option scan of the 3-D bifurcation: iterations and time per step for a few preconditioner settings."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfd_hemodynamic_amd.scenarios.simple_bifurcation import MicrovasculatureSimulation
res = float(sys.argv[1])
variants = [eval(a) for a in sys.argv[2:]] or [dict()]
for v in variants:
    opts = dict(remove_p_mean=0); opts.update(v)
    sc = MicrovasculatureSimulation("stabilized_schur", 0.01, 1.0, res=res, quiet=True, options=opts)
    tt = []
    for k in range(3):
        t0 = time.perf_counter()
        sc.solver.solveStep(); sc.solver.advance()
        tt.append(time.perf_counter() - t0)
        st = sc.solver.last_stats
        print(v, "step", k, "newton", st.newton_its, "krylov", st.krylov_its, "ms %.1f" % (1e3 * tt[-1]), "pc_setup %.1f" % st.ms_pc_setup, flush=True)
    print(v, "levels", [sc.solver.ctx.info(k) for k in (6, 19, 20, 21, 22, 23, 24)], "qout/qin %.4f" % ((lambda q: (q[1] + q[2]) / q[0])(sc.flow_rates())), flush=True)
    sc.solver.ctx.close()
