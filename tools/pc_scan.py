"""Scan preconditioner options on the bench workload (developer tool)."""
import sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from util import dfg_case, make_ctx
m = int(sys.argv[1]); nsteps = int(sys.argv[2])
case = dfg_case(m); nv = case.nv
variants = [eval(a) for a in sys.argv[3:]] or [dict(), dict(amg_smooth_ratio=4.0), dict(amg_smooth_ratio=6.0), dict(amg_smooth_ratio=12.0), dict(amg_theta=0.06),
            dict(amg_theta=0.1), dict(amg_max_coarse=500), dict(amg_max_coarse=2000), dict(cc_smooth_degree=3), dict(schur_full=1)]
for extra in variants:
    ctx = make_ctx(case)
    o = ctx.default_options()
    for k, v in extra.items(): setattr(o, k, v)
    ctx.set_options(o)
    z2, z1 = np.zeros(2 * nv), np.zeros(nv)
    ctx.set_state(u_prev=z2, p_prev=z1, u=z2, p=z1)
    kits = nits = 0; tl = []; ref = 0
    for s in range(nsteps):
        ts = time.time(); st = ctx.solve_step(); ctx.advance(); tl.append(time.time() - ts)
        kits += st.krylov_its; nits += st.newton_its; ref += st.pc_refreshes
    print(extra, "krylov", kits, "newton", nits, "ms/step(last 10)", round(1e3 * np.mean(tl[-10:]), 2), "first step s", round(tl[0], 2), "refreshes", ref, "levels", ctx.info(6), "drag", ctx.functional(0, 5) * 500, flush=True)
    ctx.close()
