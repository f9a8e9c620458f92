"""Quick on-GPU sanity run (developer tool): assembly/SpMV parity and a few steps."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from util import dfg_case, lid_case, make_oracle, make_ctx
from oracle import orc

m = int(sys.argv[1]) if len(sys.argv) > 1 else 12
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
for name, case in (("dfg", dfg_case(m)), ("lid", lid_case(2 * m))):
    nv = case.nv
    O = make_oracle(case)
    ctx = make_ctx(case)
    rng = np.random.default_rng(0)
    xv = 0.1 * rng.standard_normal(3 * nv)
    un = 0.1 * rng.standard_normal(2 * nv)
    O.set_un(un)
    F = O.assemble(xv)
    J = O.csr()
    ctx.set_state(u_prev=un, p_prev=np.zeros(nv), u=xv[:2 * nv], p=xv[2 * nv:])
    ctx.assemble(True)
    ru, rp = ctx.get_residual()
    Fg = np.concatenate([ru, rp])
    Jg = ctx.get_csr()
    print(name, "nv", nv, "F rel err", np.abs(F - Fg).max() / np.abs(F).max(), "J rel err", abs(J - Jg).max() / abs(J).max())
    y = ctx.spmv(xv)
    print("  spmv err", np.abs(y - J @ xv).max() / np.abs(J @ xv).max())
    # time steps
    o = ctx.default_options(); o.verbose = 1; o.snes_rtol = 1e-10; o.ksp_rtol = 1e-8
    ctx.set_options(o)
    ctx.set_state(u_prev=np.zeros(2 * nv), p_prev=np.zeros(nv), u=np.zeros(2 * nv), p=np.zeros(nv))
    x_o = np.zeros(3 * nv)
    opts = orc.default_opts(snes_rtol=1e-10, ksp_rtol=1e-8, sub_rtol=1e-6)
    for s in range(nsteps):
        t0 = time.time(); st = ctx.solve_step(); tg = time.time() - t0
        u, p = ctx.get_solution(); ctx.advance()
        O.set_un(x_o[:2 * nv]); t0 = time.time(); x_o, so = O.solve_step(x_o, opts); to = time.time() - t0
        xg = np.concatenate([u, p])
        print("  step", s, "gpu newton", st.newton_its, "krylov", st.krylov_its, "reason", st.reason, "%.1f ms (asm %.1f solve %.1f pc %.1f)" % (st.ms_total, st.ms_assemble, st.ms_solve, st.ms_pc_setup),
              "| orc newton", so.newton_its, "krylov", so.krylov_its, "%.0f ms" % (to * 1e3), "| rel diff", np.linalg.norm(xg - x_o) / np.linalg.norm(x_o))
    if name == "dfg":
        print("  drag gpu", 500 * ctx.functional(0, 5), "orc", 500 * O.functional(x_o, 0, case.markers["ft"].find(5)),
              "lift", 500 * ctx.functional(1, 5), 500 * O.functional(x_o, 1, case.markers["ft"].find(5)))
    print("  L2 gpu", ctx.functional(2), ctx.functional(3), "orc", O.functional(x_o, 2), O.functional(x_o, 3))
    ctx.close()
