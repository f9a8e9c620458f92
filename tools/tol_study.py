"""How tight must Newton/FGMRES be for drag/lift/|u| parity at 1e-6?  (developer study)"""
import sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from util import dfg_case, make_ctx
m = int(sys.argv[1]); nsteps = int(sys.argv[2])
case = dfg_case(m); nv = case.nv
def run(snes_rtol, ksp_rtol, extra=None):
    ctx = make_ctx(case)
    o = ctx.default_options(); o.snes_rtol = snes_rtol; o.ksp_rtol = ksp_rtol
    for k, v in (extra or {}).items(): setattr(o, k, v)
    ctx.set_options(o)
    z2, z1 = np.zeros(2*nv), np.zeros(nv)
    ctx.set_state(u_prev=z2, p_prev=z1, u=z2, p=z1)
    kits = nits = 0; t0 = time.time()
    for _ in range(nsteps):
        st = ctx.solve_step(); ctx.advance(); kits += st.krylov_its; nits += st.newton_its
    t = time.time() - t0
    r = (ctx.functional(0, 5), ctx.functional(1, 5), ctx.functional(2), kits, nits, t)
    ctx.close(); return r
ref = run(1e-13, 1e-11, dict(snes_stol=0.0))
print("ref", ref, flush=True)
for s_, k_ in [(1e-8, 1e-5), (1e-8, 1e-6), (1e-8, 1e-7), (1e-9, 1e-7), (1e-10, 1e-8), (1e-8, 1e-8)]:
    r = run(s_, k_)
    print("snes %.0e ksp %.0e: drag %.2e lift %.2e l2 %.2e | krylov %d newton %d  %.2fs" % (s_, k_, abs(r[0]/ref[0]-1), abs(r[1]/ref[1]-1), abs(r[2]/ref[2]-1), r[3], r[4], r[5]), flush=True)
print("--- PC parameter scan at default tolerances")
for extra in [dict(), dict(cheb_degree=2), dict(cheb_degree=4), dict(cheb_degree=4, cheb_ratio=20.0), dict(amg_smooth_degree=3), dict(amg_smooth_degree=1), dict(schur_full=0), dict(amg_theta=0.04), dict(amg_theta=0.15), dict(cheb_degree=2, amg_smooth_degree=1)]:
    r = run(1e-8, 1e-5, extra)
    print(extra, "krylov %d newton %d  %.2fs" % (r[3], r[4], r[5]), flush=True)
