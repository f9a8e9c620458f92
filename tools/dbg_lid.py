import sys, os, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from util import lid_case, dfg_case, make_ctx
case = lid_case(24); nv = case.nv
for deg in (1, 2):
    for tight in (0, 1):
        ctx = make_ctx(case)
        o = ctx.default_options(); o.amg_smooth_degree = deg; o.verbose = 0; o.ksp_max_it = 300
        if tight: o.snes_rtol, o.snes_stol, o.ksp_rtol = 1e-12, 0.0, 1e-10
        ctx.set_options(o)
        z2, z1 = np.zeros(2*nv), np.zeros(nv)
        ctx.set_state(u_prev=z2, p_prev=z1, u=z2, p=z1)
        try:
            for s in range(2):
                st = ctx.solve_step(); ctx.advance()
                print('graph', os.environ.get('CFDH_NO_GRAPH'), 'deg', deg, 'tight', tight, 'step', s, 'newton', st.newton_its, 'krylov', st.krylov_its, flush=True)
        except RuntimeError as e:
            print('graph', os.environ.get('CFDH_NO_GRAPH'), 'deg', deg, 'tight', tight, 'FAIL', str(e)[:80], flush=True)
        ctx.close()
