"""The restated reference solver configuration (oracle pc_kind=0: FGMRES(200) + PCFIELDSPLIT Schur FULL/SELFP with
GMRES(30)+ILU(0) on A00 and ILU(0) on Sp, stabilized_schur.py:226-273) over DFG meshes of growing size: FGMRES iterations of
the first step, against the oracle's port of the GPU preconditioner (pc_kind=2).
CPU only.  Output kept under profiles/r02_oracle_ilu_vs_amg.log."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from util import dfg_case, make_oracle
from oracle import orc
for m in (25, 50, 75, 100):
    case = dfg_case(m); nv = case.nv
    for pck in (0, 2):
        O = make_oracle(case); O.set_threads(8)
        O.set_un(np.zeros(2 * nv))
        opts = orc.default_opts(pc_kind=pck, ksp_max_it=1000)
        t0 = time.time()
        try:
            x, st = O.solve_step(np.zeros(3 * nv), opts)
            print("m=%3d %7d vertices pc_kind=%d: newton %d, fgmres %4d (inner A00-GMRES its %6d), %.1f s" % (m, nv, pck, st.newton_its, st.krylov_its, st.sub_its, time.time() - t0), flush=True)
        except RuntimeError as e:
            print("m=%3d %7d vertices pc_kind=%d: FAILED after %.1f s: %s" % (m, nv, pck, time.time() - t0, e), flush=True)
