"""P2 / Q1 solver debugging: cavity on a small node mesh, verbose FGMRES."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from cfd_hemodynamic_amd import _lib
from gen_util import LIB_ETYPE, facet_node_set, node_mesh
kind, n, pc = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
m = node_mesh(kind, n)
nv = m.num_vertices
ctx = _lib.Context(m.x, m.cells, m.facet_cells, m.facet_local, np.zeros(m.num_facets, np.int32), etype=LIB_ETYPE[kind])
ctx.set_params(0.02, 1.0, 0.02, f=(0.0, 0.0))
mid = m.facet_midpoints()
top = np.nonzero(np.isclose(mid[:, 1], m.x[:, 1].max()))[0]
rest = np.setdiff1d(np.arange(m.num_facets), top)
walls = facet_node_set(m, rest)
lid = np.setdiff1d(facet_node_set(m, top), walls)
ctx.add_dirichlet(0, walls, np.zeros((len(walls), 2)))
ctx.add_dirichlet(0, lid, np.tile([1.0, 0.0], (len(lid), 1)))
o = ctx.default_options()
o.snes_rtol, o.snes_stol, o.ksp_rtol, o.verbose, o.pc_type, o.ksp_max_it = 1e-10, 0.0, 1e-8, int(os.environ.get("VERBOSE", "1")), pc, 300
ctx.set_options(o)
ctx.set_state(u_prev=np.zeros(2 * nv), p_prev=np.zeros(nv), u=np.zeros(2 * nv), p=np.zeros(nv))
try:
    st = ctx.solve_step()
    print("converged", st.newton_its, st.krylov_its)
except Exception as e:
    print("FAILED", e)
