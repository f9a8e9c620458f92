"""Operating range of config 5 (stenosis + tree, pulsatile inlet): steps survived at a given inlet velocity.
  python tools/c5_range.py <v_max> <res> <steps> [pc_refresh] [ramp] [solver]     (env CFDH_* knobs apply; solver: stabilized_schur |
  stabilized_schur_bdf2 -- the BDF2 variant has no odd-even pressure mode)"""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
v, res, steps = float(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3])
pcr = int(sys.argv[4]) if len(sys.argv) > 4 else 0
ramp = float(sys.argv[5]) if len(sys.argv) > 5 else 0.03
solver = sys.argv[6] if len(sys.argv) > 6 else "stabilized_schur"
args = types.SimpleNamespace(config="c5", m=200, nx=288, ny=115, res=res, res3=2e-4, dt=0.001, ramp=ramp, v_max=v)
sc = bench.make_scenario(args, solver, device=0, verbose=int(os.environ.get("VERBOSE", "0")), options=dict(pc_refresh=pcr))
s = sc.solver
log = []
t0 = time.perf_counter()
for k in range(steps):
    bench.step_hook(sc, k, args.dt)
    try:
        s.solveStep(); s.advance()
    except Exception as e:
        print("FAILED at step", k + 1, str(e)[:160]); break
    st = s.last_stats
    log.append((st.newton_its, st.krylov_its, st.pc_refreshes))
print("%s v_max %g res %g nv %d pc_refresh %d ramp %g: %d steps in %.1f s; (newton, krylov, rebuilds) per step: %s" % (
    solver, v, res, sc.mesh.num_vertices, pcr, ramp, len(log), time.perf_counter() - t0, log))
