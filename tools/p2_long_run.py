"""How far does the P2/P2 backflow stenosis of `bench.py --config p2` run?  (steps survived, iterations; env knobs CFDH_* apply)"""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
guess = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ny = int(sys.argv[3]) if len(sys.argv) > 3 else 40
krtol = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-5
vmax = float(sys.argv[5]) if len(sys.argv) > 5 else 20.0
severity = float(sys.argv[6]) if len(sys.argv) > 6 else None   # None: the reference's effective 0.567
args = types.SimpleNamespace(config="p2", m=200, nx=2935, ny=ny, res=7.3e-6, res3=2e-4, dt=0.01, ramp=0.03, v_max=vmax)
kw = {} if severity is None else dict(severity=severity)
sc = bench.make_scenario(args, "stabilized_schur", device=0, options=dict(ksp_guess=guess, ksp_rtol=krtol), **kw)
s = sc.solver
log = []
for k in range(steps):
    try:
        s.solveStep(); s.advance()
    except Exception as e:
        print("FAILED at step", k + 1, str(e)[:120]); break
    log.append((s.last_stats.newton_its, s.last_stats.krylov_its))
print("v_max", vmax, "severity", severity, "ny", ny, "ksp_rtol", krtol, "ksp_guess", guess, "fp32", os.environ.get("CFDH_KRYLOV_FP32", "auto"), "steps", len(log), log)
