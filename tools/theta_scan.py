"""Strength threshold of the aggregation (amg_theta) on the 2-D configurations: iterations and time per step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cfd_hemodynamic_amd.scenarios.dfg_1 import DFG1Benchmark
from cfd_hemodynamic_amd.scenarios.stenosis import StenosisSimulation
from cfd_hemodynamic_amd.scenarios.stenosis_with_tree import StenosisWithTreeSimulation
from cfd_hemodynamic_amd.scenarios.lid_driven2D import LidDriven2DSimulation
thetas = [float(v) for v in sys.argv[1].split(",")]
def cases(th):
    o = dict(amg_theta=th)
    yield "dfg m=200", DFG1Benchmark("stabilized_schur", 0.01, 1.0, m=200, quiet=True, options=o), None
    yield "lid nx=288", LidDriven2DSimulation("stabilized_schur", 0.01, 10.0, nx=288, mu=0.01, quiet=True, options=o), None
    yield "stenosis ny=80", StenosisSimulation("stabilized_schur", 0.01, 1.0, grade="moderate", ny=80, v_max=100.0, quiet=True, options=o), None
    yield "tree 2e-5", StenosisWithTreeSimulation("stabilized_schur", 1e-3, 1.0, grade="moderate", res=2e-5, pulse_amplitude=0.5, quiet=True,
                                                   inlet_max_velocity=0.05, ramp_time=0.03, options=o), 1e-3
for th in thetas:
    for name, sc, dt in cases(th):
        kits = 0; tl = []
        try:
            for k in range(12):
                if dt: sc.set_inlet_time((k + 1) * dt)
                t = time.perf_counter(); sc.solver.solveStep(); sc.solver.advance(); tl.append(time.perf_counter() - t)
                kits += sc.solver.last_stats.krylov_its
            print("theta %.3f  %-16s krylov %5d  ms/step(last 6) %.2f  levels %d" % (th, name, kits, 1e3 * np.mean(tl[-6:]), sc.solver.ctx.info(6)), flush=True)
        except RuntimeError as e:
            print("theta %.3f  %-16s FAILED %s" % (th, name, str(e)[:60]), flush=True)
        sc.solver.ctx.close()
