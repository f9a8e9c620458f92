"""Time the assembly / SpMV / moments kernels in isolation at bench size (developer tool)."""
import sys, os
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from util import dfg_case, make_ctx
m = int(sys.argv[1]) if len(sys.argv) > 1 else 200
case = dfg_case(m); nv = case.nv
ctx = make_ctx(case)
rng = np.random.default_rng(0)
u = np.zeros((nv, 2)); u[:, 0] = 4 * 0.3 * case.mesh.x[:, 1] * (0.41 - case.mesh.x[:, 1]) / 0.41**2
u += 1e-3 * rng.uniform(-1, 1, u.shape)
ctx.set_state(u_prev=u.ravel(), p_prev=np.zeros(nv), u=u.ravel(), p=np.zeros(nv))
ctx.assemble(True)
ref = ctx.get_csr().data.copy()
ctx.profile_enable(True)
for _ in range(20):
    ctx.assemble(True)
v = rng.standard_normal(3 * nv)
ms, n = ctx.profile_get(0)
print("asm avg us", 1e3 * ms / n, "GB/s(624B/vtx)", 624.0 * nv / (ms / n * 1e-3) / 1e9, flush=True)
ms, n = ctx.profile_get(2)
print("  moments avg us", 1e3 * ms / max(n, 1))

