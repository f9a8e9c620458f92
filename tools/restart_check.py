"""FGMRES restart path: a small restart length must give the same converged step as the default (200)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from util import dfg_case, make_ctx
case = dfg_case(24); nv = case.nv
sols = []
for restart in (200, 7, 3):
    ctx = make_ctx(case)
    o = ctx.default_options(); o.snes_rtol, o.snes_stol, o.ksp_rtol, o.ksp_restart = 1e-12, 0.0, 1e-10, restart
    ctx.set_options(o)
    z2, z1 = np.zeros(2*nv), np.zeros(nv)
    ctx.set_state(u_prev=z2, p_prev=z1, u=z2, p=z1)
    for _ in range(2):
        st = ctx.solve_step(); ctx.advance()
    sols.append(np.concatenate(ctx.get_solution()))
    print("restart", restart, "newton", st.newton_its, "krylov", st.krylov_its, "fnorm %.2e" % st.fnorm)
    ctx.close()
for s in sols[1:]:
    print("rel diff vs restart 200:", np.linalg.norm(s - sols[0]) / np.linalg.norm(sols[0]))
