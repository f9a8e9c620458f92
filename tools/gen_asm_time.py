"""Kernel time of the generic assembly (Q1 bench mesh): `ctx.assemble()` repeated, HIP events of the library (kind 0).
Usage: python tools/gen_asm_time.py"""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
args = types.SimpleNamespace(config="q1", m=200, nx=2935, ny=55, res=7.3e-6, res3=2e-4, dt=0.01, ramp=0.03, v_max=100.0)
sc = bench.make_scenario(args, "stabilized_schur", device=0)
s = sc.solver
for _ in range(3):
    s.solveStep(); s.advance()
ctx = s.ctx
for mode, na in ((True, "0"), (False, "0")):
    ctx.profile_reset(); ctx.profile_enable(True)
    for _ in range(10):
        ctx.assemble(mode)
    ctx.profile_enable(False)
    ms, n = ctx.profile_get(0)
    print("noatomic", na, "want_jacobian", mode, "launches", n, "avg us %.1f" % (1e3 * ms / max(n, 1)))
