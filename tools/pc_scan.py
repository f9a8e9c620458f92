"""Scan preconditioner options on the bench workload (developer tool)."""
import sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from util import dfg_case, make_ctx
m = int(sys.argv[1]); nsteps = int(sys.argv[2])
case = dfg_case(m); nv = case.nv
for extra in [dict(), dict(schur_full=0), dict(amg_max_coarse=1500), dict(amg_max_coarse=1500, schur_full=0), dict(amg_theta=0.04), dict(amg_theta=0.15), dict(amg_smooth_ratio=4.0), dict(amg_smooth_ratio=16.0)]:
    ctx = make_ctx(case)
    o = ctx.default_options()
    for k, v in extra.items(): setattr(o, k, v)
    ctx.set_options(o)
    z2, z1 = np.zeros(2 * nv), np.zeros(nv)
    ctx.set_state(u_prev=z2, p_prev=z1, u=z2, p=z1)
    kits = nits = 0; tl = []
    for s in range(nsteps):
        ts = time.time(); st = ctx.solve_step(); ctx.advance(); tl.append(time.time() - ts)
        kits += st.krylov_its; nits += st.newton_its
    print(extra, "krylov", kits, "newton", nits, "ms/step(last 5)", round(1e3 * np.mean(tl[-5:]), 2), "first step s", round(tl[0], 2), "levels", ctx.info(6), flush=True)
    ctx.close()
