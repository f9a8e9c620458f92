"""Nodal meshes for the element types beyond P1 triangles:

* `QuadMesh` / `create_rectangle(..., cell_type="quadrilateral")` -- the quadrilateral rectangle of
  /root/reference/src/scenarios/unit_square_pipe.py:101-105 (`create_rectangle(MPI.COMM_WORLD, [[0, 0], [L, H]], [NX, NY],
  cell_type=CellType.quadrilateral)`), Q1 nodes = vertices, DOLFINx local order (0,0),(1,0),(0,1),(1,1);
* `NodeMesh(mesh, 2)` -- the P2 nodes of a straight-sided triangle mesh (`p_grade = 2`,
  stabilized_schur_backflow.py:84-87): vertices first, then one node per edge; cells [nc,6] in DOLFINx local order
  (edge node i opposite vertex i).

* `HexMesh` / `create_box(..., cell_type="hexahedron")` -- the hexahedral box of
  /root/reference/src/scenarios/unit_cube_pipe.py:103-109, Q1 nodes = vertices, DOLFINx local order v = i + 2 j + 4 k;
* `NodeMesh3D(mesh)` -- the P2 nodes of a straight-sided tetrahedral mesh (`p_grade = 2` on the 3-D meshes of
  scenario_factory.py:47-49): vertices first, then one node per edge; cells [nc,10] in DOLFINx local order (edge nodes in Basix
  edge order (2,3) (1,3) (1,2) (0,3) (0,2) (0,1)).

All expose what `fem.FunctionSpace`, `locate_dofs_topological`, `Function.interpolate` and the solver plugins read from a
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


_HEX_FACETS = np.array([[0, 1, 2, 3], [0, 1, 4, 5], [0, 2, 4, 6], [1, 3, 5, 7], [2, 3, 6, 7], [4, 5, 6, 7]])
_TET_EDGES = [(2, 3), (1, 3), (1, 2), (0, 3), (0, 2), (0, 1)]


class _Topo3:
    dim = 3

    def __init__(self, name):
        self._name = name

    def cell_name(self):
        return self._name

    def create_connectivity(self, d0, d1):
        return None


class _Geom3:
    dim = 3

    def __init__(self, mesh):
        self._mesh = mesh

    @property
    def x(self):
        return self._mesh.x


class HexMesh:
    """Hexahedral mesh of parallelepipeds.  cells int32 [nc,8] in DOLFINx order (vertex v = i + 2 j + 4 k); exterior facets
    with the local facet numbering 0:(0,1,2,3) 1:(0,1,4,5) 2:(0,2,4,6) 3:(1,3,5,7) 4:(2,3,6,7) 5:(4,5,6,7)."""
    etype = ELEM_Q1_QUADRILATERAL  # CFDH_ELEM_Q1: quadrilaterals for gdim 2, hexahedra for gdim 3

    def __init__(self, cells, x, comm=None, name="mesh"):
        self.cells = np.ascontiguousarray(cells, dtype=np.int32)
        self.x = np.ascontiguousarray(x, dtype=np.float64)[:, :3].copy()
        assert self.cells.ndim == 2 and self.cells.shape[1] == 8
        p = self.x[self.cells]
        for v in range(8):
            i, j, k = v & 1, (v >> 1) & 1, (v >> 2) & 1
            if np.abs(p[:, v] - (p[:, 0] + i * (p[:, 1] - p[:, 0]) + j * (p[:, 2] - p[:, 0]) + k * (p[:, 4] - p[:, 0]))).max() > 1e-10 * np.abs(p).max():
                raise ValueError("only parallelepiped cells (affine Q1) are supported")
        self.name = name
        self.comm = comm or _SerialComm()
        self.topology = _Topo3("hexahedron")
        self.geometry = _Geom3(self)
        q = self.cells[:, _HEX_FACETS].reshape(-1, 4).astype(np.int64)   # facet index = 6 cell + local
        srt = np.sort(q, axis=1)
        n1 = len(self.x) + 1
        key = ((srt[:, 0] * n1 + srt[:, 1]) * n1 + srt[:, 2])            # three smallest vertices identify a quadrilateral facet
        _, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)
        single = np.nonzero(cnt[np.asarray(inv).ravel()] == 1)[0]
        self.facet_cells = (single // 6).astype(np.int32)
        self.facet_local = (single % 6).astype(np.int32)
        self.facet_vertices = q[single].astype(np.int32)
        self.facet_marker = np.zeros(len(single), dtype=np.int32)

    num_vertices = property(lambda self: self.x.shape[0])
    num_cells = property(lambda self: self.cells.shape[0])
    num_facets = property(lambda self: len(self.facet_cells))

    def facet_midpoints(self):
        return self.x[self.facet_vertices].mean(axis=1)

    def h(self, dim=3, entities=None):
        c = self.cells if entities is None else self.cells[np.asarray(entities)]
        p = self.x[c]
        return np.max([np.linalg.norm(p[:, a] - p[:, b], axis=1) for a in range(8) for b in range(a + 1, 8)], axis=0)

    def cell_volumes(self):
        p = self.x[self.cells]
        return np.abs(np.linalg.det(np.stack([p[:, 1] - p[:, 0], p[:, 2] - p[:, 0], p[:, 4] - p[:, 0]], axis=2)))

    def set_facet_markers(self, facets, values):
        self.facet_marker[np.asarray(facets, dtype=np.int64)] = np.asarray(values, dtype=np.int32)


def create_box(p0, p1, n, cell_type="hexahedron", comm=None):
    """`dolfinx.mesh.create_box(comm, [p0, p1], n, cell_type=CellType.hexahedron)` (unit_cube_pipe.py:103-109)."""
    if str(cell_type).split(".")[-1] != "hexahedron":
        raise ValueError("create_box builds hexahedral meshes; tetrahedral generators live in cfd_hemodynamic_amd.mesh3d")
    nx, ny, nz = int(n[0]), int(n[1]), int(n[2])
    xs, ys, zs = (np.linspace(p0[d], p1[d], m + 1) for d, m in enumerate((nx, ny, nz)))
    X, Y, Z = np.meshgrid(xs, ys, zs, indexing="ij")
    x = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    i, j, k = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    sx, sy = (ny + 1) * (nz + 1), nz + 1
    v = (i * sx + j * sy + k).ravel()
    cells = np.stack([v + a * sx + b * sy + c for c in (0, 1) for b in (0, 1) for a in (0, 1)], axis=1)
    return HexMesh(cells, x, comm=comm)


class NodeMesh3D:
    """P2 nodes of a tetrahedral `Mesh3D`: the same cells and exterior facets, six more nodes per cell / three more per facet."""
    etype = ELEM_P2_TRIANGLE  # CFDH_ELEM_P2: triangles for gdim 2, tetrahedra for gdim 3

    def __init__(self, mesh):
        self.base = mesh
        c = mesh.cells.astype(np.int64)
        nv = mesh.num_vertices
        e = np.concatenate([np.sort(c[:, list(l)], axis=1) for l in _TET_EDGES])
        ue, inv = np.unique(e, axis=0, return_inverse=True)
        inv = np.asarray(inv).reshape(6, len(c)).T
        self.x = np.vstack([mesh.x, 0.5 * (mesh.x[ue[:, 0]] + mesh.x[ue[:, 1]])])
        self.cells = np.hstack([c, nv + inv]).astype(np.int32)
        self.edges = ue.astype(np.int32)
        self.num_base_vertices = nv
        self.facet_cells, self.facet_local = mesh.facet_cells, mesh.facet_local
        # nodes of an exterior facet (opposite local vertex f): its three vertices and the nodes of the three edges among them
        fe = np.zeros((4, 3), dtype=np.int64)
        for f in range(4):
            fe[f] = [4 + k for k, (a, b) in enumerate(_TET_EDGES) if a != f and b != f]
        self.facet_vertices = np.hstack([mesh.facet_vertices, np.take_along_axis(self.cells[mesh.facet_cells], fe[mesh.facet_local], axis=1)]).astype(np.int32)
        self.name, self.comm = mesh.name, mesh.comm
        self.topology = mesh.topology
        self.geometry = _Geom3(self)

    facet_marker = property(lambda self: self.base.facet_marker)
    num_vertices = property(lambda self: self.x.shape[0])
    num_cells = property(lambda self: self.cells.shape[0])
    num_facets = property(lambda self: len(self.facet_cells))

    def facet_midpoints(self):
        return self.base.facet_midpoints()

    def h(self, dim=3, entities=None):
        return self.base.h(dim, entities)

    def cell_volumes(self):
        return self.base.cell_volumes()

    def set_facet_markers(self, facets, values):
        self.base.set_facet_markers(facets, values)


def dof_mesh(mesh, degree):
    """The mesh whose "vertices" are the nodes of Lagrange elements of the given degree on `mesh`."""
    degree = int(degree)
    if isinstance(mesh, (QuadMesh, HexMesh)):
        if degree != 1:
            raise NotImplementedError("quadrilateral / hexahedral cells: Q1 only")
        return mesh
    if degree == 1:
        return mesh
    if degree == 2 and mesh.geometry.dim == 2:
        if getattr(mesh, "_p2_nodes", None) is None:
            mesh._p2_nodes = NodeMesh(mesh)
        return mesh._p2_nodes
    if degree == 2 and mesh.geometry.dim == 3:
        if getattr(mesh, "_p2_nodes", None) is None:
            mesh._p2_nodes = NodeMesh3D(mesh)
        return mesh._p2_nodes
    raise NotImplementedError("Lagrange degree %d on %s cells" % (degree, mesh.topology.cell_name()))
