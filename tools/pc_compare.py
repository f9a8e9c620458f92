"""Compare preconditioner variants on the bench workload (developer tool)."""
import sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from util import dfg_case, lid_case, make_ctx
m = int(sys.argv[1]); nsteps = int(sys.argv[2])
for name, case in (("dfg", dfg_case(m)), ("lid", lid_case(int(1.4 * m)))):
    nv = case.nv
    for pc_type in (0, 1):
        for tight in (0, 1):
            ctx = make_ctx(case)
            o = ctx.default_options(); o.pc_type = pc_type; o.verbose = int(sys.argv[3]) if len(sys.argv) > 3 else 0
            if tight: o.snes_rtol, o.snes_stol, o.ksp_rtol = 1e-12, 0.0, 1e-10
            ctx.set_options(o)
            z2, z1 = np.zeros(2 * nv), np.zeros(nv)
            ctx.set_state(u_prev=z2, p_prev=z1, u=z2, p=z1)
            kits = nits = 0; t0 = time.time(); tl = []
            try:
                for s in range(nsteps):
                    ts = time.time(); st = ctx.solve_step(); ctx.advance(); tl.append(time.time() - ts)
                    kits += st.krylov_its; nits += st.newton_its
                fd = ctx.functional(0, 5) if name == "dfg" else 0.0
                print(name, "nv", nv, "pc", pc_type, "tight", tight, "krylov", kits, "newton", nits, "ms/step(last)", 1e3 * np.mean(tl[-3:]), "drag", fd, "l2", ctx.functional(2), flush=True)
            except RuntimeError as e:
                print(name, "pc", pc_type, "tight", tight, "FAIL", str(e)[:100], flush=True)
            ctx.close()
