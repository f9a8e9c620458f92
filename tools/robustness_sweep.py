"""Developer tool: run the shipped scenarios / solver plugins for a few steps at moderate size with default
options and print iteration counts and timings (looks for stagnation, refresh storms, non-convergence)."""
import sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from cfd_hemodynamic_amd.scenarios.dfg_1 import DFG1Benchmark
from cfd_hemodynamic_amd.scenarios.lid_driven2D import LidDriven2DSimulation
from cfd_hemodynamic_amd.scenarios.stenosis import StenosisSimulation

def run(name, sc, nsteps):
    S = sc.solver
    t0 = time.perf_counter(); its = []; newt = []; refresh = 0
    try:
        for s in range(nsteps):
            S.solveStep(); S.assemble_wss(); S.advance()
            st = S.last_stats; its.append(st.krylov_its); newt.append(st.newton_its); refresh += st.pc_refreshes
        print("%-46s nv %7d  newton %s  krylov %s  refreshes %d  %.1f ms/step (last half)" % (
            name, sc.mesh.num_vertices, newt[:3] + newt[-2:], its[:3] + its[-2:], refresh,
            1e3 * (time.perf_counter() - t0) / nsteps), flush=True)
    except RuntimeError as e:
        print("%-46s FAILED at step %d: %s" % (name, len(its), e), flush=True)

for solver in ("stabilized_schur", "stabilized_schur_bdf2"):
    run("dfg_1 m=120 " + solver, DFG1Benchmark(solver, 0.01, 1.0, m=120, quiet=True), 20)
    run("dfg_1 m=120 dt=0.001 " + solver, DFG1Benchmark(solver, 0.001, 1.0, m=120, quiet=True), 20)
    run("lid nx=256 mu=0.01 " + solver, LidDriven2DSimulation(solver, 0.01, 1.0, nx=256, mu=0.01, quiet=True), 20)
    run("lid nx=256 mu=0.001 " + solver, LidDriven2DSimulation(solver, 0.01, 1.0, nx=256, mu=0.001, quiet=True), 20)
    for vm in (50.0, 300.0):
        run("stenosis ny=48 v_max=%g %s" % (vm, solver), StenosisSimulation(solver, 0.01, 1.0, ny=48, v_max=vm, quiet=True), 20)
for vm in (50.0, 300.0):
    run("stenosis ny=48 v_max=%g backflow" % vm, StenosisSimulation("stabilized_schur_backflow", 0.01, 1.0, ny=48, v_max=vm, quiet=True, beta_backflow=0.2), 20)
run("dfg_1 m=120 tight tol", DFG1Benchmark("stabilized_schur", 0.01, 1.0, m=120, quiet=True, options=dict(snes_rtol=1e-12, snes_stol=0.0, ksp_rtol=1e-10)), 10)
