"""Nodal meshes for the element types beyond P1 triangles:

* `QuadMesh` / `create_rectangle(..., cell_type="quadrilateral")` -- the quadrilateral rectangle of
  /root/reference/src/scenarios/unit_square_pipe.py:101-105 (`create_rectangle(MPI.COMM_WORLD, [[0, 0], [L, H]], [NX, NY],
  cell_type=CellType.quadrilateral)`), Q1 nodes = vertices, DOLFINx local order (0,0),(1,0),(0,1),(1,1);
* `NodeMesh(mesh, 2)` -- the P2 nodes of a straight-sided triangle mesh (`p_grade = 2`,
  stabilized_schur_backflow.py:84-87): vertices first, then one node per edge; cells [nc,6] in DOLFINx local order
  (edge node i opposite vertex i).

Both expose what `fem.FunctionSpace`, `locate_dofs_topological`, `Function.interpolate` and the solver plugins read from a
mesh (`x`, `cells`, `num_vertices` = number of NODES, `facet_cells/local/vertices/marker`, `geometry.x`, `topology`), so the
function-space layer and the boundary-condition objects work on nodes unchanged.  `etype` is the CFDH_ELEM_* code of the C-ABI.
"""
from __future__ import annotations

import numpy as np

from .mesh import Mesh, _Geometry, _SerialComm

ELEM_P1, ELEM_P2_TRIANGLE, ELEM_Q1_QUADRILATERAL, ELEM_P1_GENERIC = 0, 1, 2, 3
_QUAD_FACETS = np.array([[0, 1], [0, 2], [1, 3], [2, 3]])


class _Topo:
    def __init__(self, name):
        self.dim = 2
        self._name = name

    def cell_name(self):
        return self._name

    def create_connectivity(self, d0, d1):
        return None


class QuadMesh:
    """Quadrilateral mesh of parallelograms.  cells int32 [nc,4] in DOLFINx order; exterior facets with the local facet
    numbering 0:(0,1) 1:(0,2) 2:(1,3) 3:(2,3)."""
    etype = ELEM_Q1_QUADRILATERAL

    def __init__(self, cells, x, comm=None, name="mesh"):
        self.cells = np.ascontiguousarray(cells, dtype=np.int32)
        self.x = np.ascontiguousarray(x, dtype=np.float64)[:, :2].copy()
        assert self.cells.ndim == 2 and self.cells.shape[1] == 4
        p = self.x[self.cells]
        if np.abs(p[:, 3] - (p[:, 1] + p[:, 2] - p[:, 0])).max() > 1e-10 * np.abs(p).max():
            raise ValueError("only parallelogram cells (affine Q1) are supported")
        self.name = name
        self.comm = comm or _SerialComm()
        self.topology = _Topo("quadrilateral")
        self.geometry = _Geometry(self)
        c = self.cells
        e = c[:, _QUAD_FACETS].reshape(-1, 2)                      # [4 nc, 2], facet index = 4 cell + local
        key = np.sort(e, axis=1).astype(np.int64)
        key = key[:, 0] * (len(self.x) + 1) + key[:, 1]
        _, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)
        single = np.nonzero(cnt[inv] == 1)[0]
        self.facet_cells = (single // 4).astype(np.int32)
        self.facet_local = (single % 4).astype(np.int32)
        self.facet_vertices = e[single].astype(np.int32)
        self.facet_marker = np.zeros(len(single), dtype=np.int32)

    num_vertices = property(lambda self: self.x.shape[0])
    num_cells = property(lambda self: self.cells.shape[0])
    num_facets = property(lambda self: len(self.facet_cells))

    def facet_midpoints(self):
        fv = self.facet_vertices
        return 0.5 * (self.x[fv[:, 0]] + self.x[fv[:, 1]])

    def h(self, dim=2, entities=None):
        c = self.cells if entities is None else self.cells[np.asarray(entities)]
        p = self.x[c]
        return np.max([np.linalg.norm(p[:, a] - p[:, b], axis=1) for a in range(4) for b in range(a + 1, 4)], axis=0)

    def set_facet_markers(self, facets, values):
        self.facet_marker[np.asarray(facets, dtype=np.int64)] = np.asarray(values, dtype=np.int32)


def create_rectangle(p0, p1, n, cell_type="quadrilateral", comm=None):
    """`dolfinx.mesh.create_rectangle(comm, [p0, p1], n, cell_type=...)` for quadrilaterals (unit_square_pipe.py:101-105);
    triangles: `mesh.create_unit_square` and friends."""
    if str(cell_type).split(".")[-1] != "quadrilateral":
        raise ValueError("create_rectangle builds quadrilateral meshes; triangle generators live in cfd_hemodynamic_amd.mesh")
    nx, ny = int(n[0]), int(n[1])
    xs = np.linspace(p0[0], p1[0], nx + 1)
    ys = np.linspace(p0[1], p1[1], ny + 1)
    X, Y = np.meshgrid(xs, ys, indexing="ij")
    x = np.stack([X.ravel(), Y.ravel()], axis=1)
    i, j = np.meshgrid(np.arange(nx), np.arange(ny), indexing="ij")
    v = (i * (ny + 1) + j).ravel()
    cells = np.stack([v, v + (ny + 1), v + 1, v + (ny + 1) + 1], axis=1)
    return QuadMesh(cells, x, comm=comm)


class NodeMesh:
    """P2 nodes of a triangle `Mesh`: the same cells and exterior facets, three more nodes per cell / one more per facet."""
    etype = ELEM_P2_TRIANGLE

    def __init__(self, mesh: Mesh):
        self.base = mesh
        c = mesh.cells.astype(np.int64)
        nv = mesh.num_vertices
        loc = [(1, 2), (0, 2), (0, 1)]
        e = np.concatenate([np.sort(c[:, l], axis=1) for l in loc])
        ue, inv = np.unique(e, axis=0, return_inverse=True)
        inv = np.asarray(inv).reshape(3, len(c)).T
        self.x = np.vstack([mesh.x, 0.5 * (mesh.x[ue[:, 0]] + mesh.x[ue[:, 1]])])
        self.cells = np.hstack([c, nv + inv]).astype(np.int32)
        self.edges = ue.astype(np.int32)
        self.num_base_vertices = nv
        self.facet_cells, self.facet_local = mesh.facet_cells, mesh.facet_local
        # nodes of an exterior facet: its two vertices and the edge node opposite to the facet's local vertex
        self.facet_vertices = np.hstack([mesh.facet_vertices, self.cells[mesh.facet_cells, 3 + mesh.facet_local][:, None]]).astype(np.int32)
        self.name, self.comm = mesh.name, mesh.comm
        self.topology = mesh.topology
        self.geometry = _Geometry(self)

    facet_marker = property(lambda self: self.base.facet_marker)  # tags are set on the geometric mesh
    num_vertices = property(lambda self: self.x.shape[0])
    num_cells = property(lambda self: self.cells.shape[0])
    num_facets = property(lambda self: len(self.facet_cells))

    def facet_midpoints(self):
        return self.base.facet_midpoints()

    def h(self, dim=2, entities=None):
        return self.base.h(dim, entities)

    def set_facet_markers(self, facets, values):
        self.base.set_facet_markers(facets, values)


def dof_mesh(mesh, degree):
    """The mesh whose "vertices" are the nodes of Lagrange elements of the given degree on `mesh`."""
    degree = int(degree)
    if isinstance(mesh, QuadMesh):
        if degree != 1:
            raise NotImplementedError("quadrilateral cells: Q1 only")
        return mesh
    if degree == 1:
        return mesh
    if degree == 2 and mesh.geometry.dim == 2:
        if getattr(mesh, "_p2_nodes", None) is None:
            mesh._p2_nodes = NodeMesh(mesh)
        return mesh._p2_nodes
    raise NotImplementedError("Lagrange degree %d on %s cells" % (degree, mesh.topology.cell_name()))
