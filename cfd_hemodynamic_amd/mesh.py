"""Minimal simplicial mesh layer + deterministic synthetic mesh generators.

The reference takes its meshes from DOLFINx (`create_unit_square`,
`gmshio.model_to_mesh`; /root/reference/src/scenarios/lid_driven2D.py:29,
/root/reference/src/scenarios/dfg_1.py:97-171).  Neither DOLFINx nor gmsh exist
on the GPU box, so this module supplies the small attribute surface the
reference's Scenario/Solver code touches (`mesh.topology.dim`,
`mesh.geometry.dim`, `mesh.geometry.x`, `mesh.comm`, `mesh.h`,
`locate_entities_boundary`, `meshtags(...).find`) on top of plain NumPy arrays,
plus own generators for the BASELINE configs.

2-D affine P1 triangles only (the BASELINE configs C1-C4 are 2-D).
"""
from __future__ import annotations

import numpy as np


class _SerialComm:
    """Stand-in for `mesh.comm` (mpi4py communicator in the reference).

    One process per GPU: rank/size come from torch.distributed when it is
    initialised, else 0/1.  Only what the reference's Scenario loop calls."""

    def __init__(self, rank=0, size=1):
        self.rank = rank
        self.size = size

    def barrier(self):
        return None

    def allreduce(self, v, op=None):  # serial: identity
        return v

    def gather(self, v, root=0):
        return [v]


class PartCommView:
    """`mesh.comm` of a partitioned run.  Every rank holds the whole (replicated) mesh and the harness only ever
    reduces values that are already global (fields gathered through `x.array`, functionals reduced inside the
    library), so `allreduce` is the identity; `rank` and `barrier` are the launcher's, which is what the
    rank-0 guards around printing and file output need."""

    def __init__(self, part_comm):
        self._pc = part_comm
        self.rank, self.size = part_comm.rank, part_comm.size

    def barrier(self):
        self._pc.barrier()

    def allreduce(self, v, op=None):
        return v

    def gather(self, v, root=0):
        return [v] if self.rank == root else None


class _Topology:
    def __init__(self, mesh):
        self._mesh = mesh
        self.dim = 2

    def cell_name(self):
        return "triangle"

    def create_connectivity(self, d0, d1):  # connectivity is always available
        return None


class _Geometry:
    def __init__(self, mesh):
        self._mesh = mesh
        self.dim = 2

    @property
    def x(self):
        """[nv,3] coordinates (z=0), as DOLFINx exposes them."""
        m = self._mesh
        if getattr(self, "_x3", None) is None or self._x3.shape[0] != m.num_vertices:
            self._x3 = np.zeros((m.num_vertices, 3))
            self._x3[:, :2] = m.x
        return self._x3


class Mesh:
    """Affine triangle mesh.

    cells : int32 [nc,3] vertex ids; x : float64 [nv,2].
    Exterior facets (edges with exactly one adjacent cell) are enumerated once:
    facet_cells[f] = owning cell, facet_local[f] = local index of the vertex
    OPPOSITE to the facet (UFC/Basix numbering of triangle facets),
    facet_vertices[f] = its two vertex ids.
    """

    def __init__(self, cells, x, comm=None, name="mesh"):
        cells = np.ascontiguousarray(cells, dtype=np.int32)
        x = np.ascontiguousarray(x, dtype=np.float64)[:, :2].copy()
        assert cells.ndim == 2 and cells.shape[1] == 3
        # positive orientation
        a = x[cells[:, 0]]
        b = x[cells[:, 1]]
        c = x[cells[:, 2]]
        det = (b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (c[:, 0] - a[:, 0])
        if np.any(det == 0.0):
            raise ValueError("degenerate cell in mesh")
        flip = det < 0
        if flip.any():
            cells = cells.copy()
            cells[flip, 1], cells[flip, 2] = cells[flip, 2].copy(), cells[flip, 1].copy()
        self.cells = cells
        self.x = x
        self.name = name
        self.comm = comm or _SerialComm()
        self.topology = _Topology(self)
        self.geometry = _Geometry(self)
        self._build_facets()

    @property
    def num_vertices(self):
        return self.x.shape[0]

    @property
    def num_cells(self):
        return self.cells.shape[0]

    def _build_facets(self):
        c = self.cells
        nc = c.shape[0]
        # local facet i is opposite local vertex i: vertices (i+1, i+2)
        e = np.stack([c[:, [1, 2]], c[:, [2, 0]], c[:, [0, 1]]], axis=1).reshape(-1, 2)
        es = np.sort(e, axis=1).astype(np.int64)
        key = es[:, 0] * (self.num_vertices + 1) + es[:, 1]
        order = np.argsort(key, kind="stable")
        ks = key[order]
        first = np.ones(len(ks), bool)
        first[1:] = ks[1:] != ks[:-1]
        last = np.ones(len(ks), bool)
        last[:-1] = ks[1:] != ks[:-1]
        single = order[first & last]
        single.sort()
        self.facet_cells = (single // 3).astype(np.int32)
        self.facet_local = (single % 3).astype(np.int32)
        self.facet_vertices = e[single].astype(np.int32)
        self.facet_marker = np.zeros(len(single), dtype=np.int32)
        assert nc > 0

    @property
    def num_facets(self):
        return len(self.facet_cells)

    def facet_midpoints(self):
        fv = self.facet_vertices
        return 0.5 * (self.x[fv[:, 0]] + self.x[fv[:, 1]])

    def h(self, dim=2, entities=None):
        """Greatest vertex-vertex distance per cell (DOLFINx `mesh.h`;
        used at /root/reference/src/solvers/stabilized_schur.py:85-88)."""
        c = self.cells if entities is None else self.cells[np.asarray(entities)]
        p = self.x[c]
        d01 = np.linalg.norm(p[:, 0] - p[:, 1], axis=1)
        d12 = np.linalg.norm(p[:, 1] - p[:, 2], axis=1)
        d20 = np.linalg.norm(p[:, 2] - p[:, 0], axis=1)
        return np.maximum(d01, np.maximum(d12, d20))

    def cell_areas(self):
        p = self.x[self.cells]
        return 0.5 * np.abs(
            (p[:, 1, 0] - p[:, 0, 0]) * (p[:, 2, 1] - p[:, 0, 1])
            - (p[:, 1, 1] - p[:, 0, 1]) * (p[:, 2, 0] - p[:, 0, 0])
        )

    def eval_p1(self, values, points, tol=1e-12):
        """Point evaluation of a scalar P1 field (the bb_tree / compute_colliding_cells / Function.eval
        sequence of dfg_1.py:213-253 and stenosis.py:163-190): value at each point from the first cell that
        contains it, NaN when no cell does."""
        X = self.x[self.cells]
        d = (X[:, 1, 0] - X[:, 0, 0]) * (X[:, 2, 1] - X[:, 0, 1]) - (X[:, 1, 1] - X[:, 0, 1]) * (X[:, 2, 0] - X[:, 0, 0])
        vals = np.asarray(values, dtype=np.float64)
        out = np.full(len(points), np.nan)
        for k, pt in enumerate(points):
            l1 = ((pt[0] - X[:, 0, 0]) * (X[:, 2, 1] - X[:, 0, 1]) - (pt[1] - X[:, 0, 1]) * (X[:, 2, 0] - X[:, 0, 0])) / d
            l2 = ((X[:, 1, 0] - X[:, 0, 0]) * (pt[1] - X[:, 0, 1]) - (X[:, 1, 1] - X[:, 0, 1]) * (pt[0] - X[:, 0, 0])) / d
            l0 = 1.0 - l1 - l2
            ok = np.nonzero((l0 >= -tol) & (l1 >= -tol) & (l2 >= -tol))[0]
            if len(ok):
                c = ok[0]
                out[k] = l0[c] * vals[self.cells[c, 0]] + l1[c] * vals[self.cells[c, 1]] + l2[c] * vals[self.cells[c, 2]]
        return out

    def set_facet_markers(self, facets, values):
        self.facet_marker[np.asarray(facets, dtype=np.int64)] = np.asarray(values, dtype=np.int32)


class MeshTags:
    """Facet tags with the `find(marker)` the scenarios use
    (/root/reference/src/scenarios/dfg_1.py:61)."""

    def __init__(self, mesh, dim, indices, values, name="Facet markers"):
        self.mesh = mesh
        self.dim = dim
        self.indices = np.asarray(indices, dtype=np.int32)
        self.values = np.asarray(values, dtype=np.int32)
        self.name = name
        mesh.set_facet_markers(self.indices, self.values)

    def find(self, value):
        return self.indices[self.values == value]


def meshtags(mesh, dim, indices, values):
    return MeshTags(mesh, dim, indices, values)


def locate_entities_boundary(mesh, dim, marker):
    """Exterior facets ALL of whose vertices satisfy `marker(x[3,n])`
    (DOLFINx rule)."""
    assert dim == mesh.topology.dim - 1
    X = mesh.geometry.x.T  # [3,nv]
    ok = np.asarray(marker(X), dtype=bool)
    fv = mesh.facet_vertices
    sel = ok[np.asarray(fv)].all(axis=1)   # edges, triangles, quadrilateral facets of hexahedra alike
    return np.nonzero(sel)[0].astype(np.int32)


# --------------------------------------------------------------------------
# generators
# --------------------------------------------------------------------------

def _split_quads(x, quads):
    """Split quads (v00,v10,v11,v01) along the shorter diagonal (ties: 00-11)."""
    q = np.asarray(quads, dtype=np.int64)
    d0 = np.linalg.norm(x[q[:, 0]] - x[q[:, 2]], axis=1)
    d1 = np.linalg.norm(x[q[:, 1]] - x[q[:, 3]], axis=1)
    use0 = d0 <= d1 * (1.0 + 1e-12)
    t = np.empty((len(q), 2, 3), dtype=np.int64)
    # diagonal 00-11
    t[use0, 0] = q[use0][:, [0, 1, 2]]
    t[use0, 1] = q[use0][:, [0, 2, 3]]
    # diagonal 10-01
    n0 = ~use0
    t[n0, 0] = q[n0][:, [0, 1, 3]]
    t[n0, 1] = q[n0][:, [1, 2, 3]]
    return t.reshape(-1, 3)


def create_unit_square(nx, ny=None, comm=None):
    """Unit square, nx*ny squares each split along (i,j)-(i+1,j+1)
    (DOLFINx `create_unit_square` default diagonal "right";
    /root/reference/src/scenarios/lid_driven2D.py:29)."""
    ny = nx if ny is None else ny
    xs = np.linspace(0.0, 1.0, nx + 1)
    ys = np.linspace(0.0, 1.0, ny + 1)
    X, Y = np.meshgrid(xs, ys, indexing="xy")  # row-major in y
    x = np.stack([X.ravel(), Y.ravel()], axis=1)
    j, i = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
    v0 = (j * (nx + 1) + i).ravel()
    v1 = v0 + 1
    v2 = v0 + (nx + 1)
    v3 = v2 + 1
    cells = np.empty((2 * nx * ny, 3), dtype=np.int64)
    cells[0::2] = np.stack([v0, v1, v3], axis=1)
    cells[1::2] = np.stack([v0, v2, v3], axis=1)
    return Mesh(cells, x, comm=comm, name="unit_square")


DFG_L = 2.2
DFG_H = 0.41
DFG_C = (0.2, 0.2)
DFG_R = 0.05


def create_dfg_channel(m, comm=None):
    """DFG 2D-1 domain [0,2.2]x[0,0.41] minus the disk c=(0.2,0.2), r=0.05
    (geometry of /root/reference/src/scenarios/dfg_1.py:97-110), meshed by an
    own deterministic block generator (gmsh is not available):

      * an O-grid of 4m x m quads between the cylinder and the box
        [0,0.41]^2, radially graded (ratio ~5.2 between outer and inner
        spacing, mirroring the reference's LcMin=r/6 ... LcMax=H/13 grading,
        dfg_1.py:146-155), and
      * a uniform block of nx x m quads for x in [0.41, 2.2],

    every quad split along its shorter diagonal.  Nv ~ 8.4 m^2:
    m=18 -> ~2.7k vertices (C1, ~8k DOF); m=200 -> ~336k vertices (C3, ~1M DOF).

    Returns (mesh, facet_tags) with the reference's markers
    inlet=2, outlet=3, wall=4, obstacle=5 (dfg_1.py:18-22).
    """
    m = int(m)
    assert m >= 4
    L, H, r = DFG_L, DFG_H, DFG_R
    cx, cy = DFG_C
    xb = H  # box [0,xb]x[0,H] around the cylinder
    t = np.arange(m) / m
    zeros = np.zeros(m)
    # box boundary, counter-clockwise from (0,0): bottom, right, top, left
    S = np.concatenate(
        [
            np.stack([xb * t, zeros], 1),
            np.stack([xb + zeros, H * t], 1),
            np.stack([xb * (1 - t), H + zeros], 1),
            np.stack([zeros, H * (1 - t)], 1),
        ]
    )
    d = S - np.array([cx, cy])
    C = np.array([cx, cy]) + r * d / np.linalg.norm(d, axis=1)[:, None]
    nr = m
    q = 5.2 ** (1.0 / (nr - 1))
    s = (q ** np.arange(nr + 1) - 1.0) / (q**nr - 1.0)
    s[-1] = 1.0
    nk = 4 * m
    # O-grid vertices: id = k*(nr+1)+j
    P = C[:, None, :] + s[None, :, None] * (S - C)[:, None, :]
    # snap the outer ring exactly onto the box
    P[:, nr, :] = S
    xo = P.reshape(-1, 2)
    k = np.arange(nk)
    kp = (k + 1) % nk
    jj = np.arange(nr)
    K, J = np.meshgrid(k, jj, indexing="ij")
    KP = np.meshgrid(kp, jj, indexing="ij")[0]
    q_o = np.stack(
        [
            (K * (nr + 1) + J).ravel(),
            (K * (nr + 1) + J + 1).ravel(),
            (KP * (nr + 1) + J + 1).ravel(),
            (KP * (nr + 1) + J).ravel(),
        ],
        axis=1,
    )
    # channel block
    nx = int(round((L - xb) / (H / m)))
    n_o = xo.shape[0]
    xi = xb + (L - xb) * np.arange(1, nx + 1) / nx
    yj = H * np.arange(m + 1) / m
    XI, YJ = np.meshgrid(xi, yj, indexing="ij")
    xc = np.stack([XI.ravel(), YJ.ravel()], axis=1)  # id = n_o + (i-1)*(m+1)+j

    def col0(j):
        # vertex on x=xb at y=H*j/m: O-grid outer ring, side 1 (k=m+j), j=m -> k=2m
        return (m + j) * (nr + 1) + nr

    def cid(i, j):
        if i == 0:
            return col0(j)
        return n_o + (i - 1) * (m + 1) + j

    ii, jj2 = np.meshgrid(np.arange(nx), np.arange(m), indexing="ij")
    ii = ii.ravel()
    jj2 = jj2.ravel()
    cidv = np.vectorize(cid)
    q_c = np.stack([cidv(ii, jj2), cidv(ii + 1, jj2), cidv(ii + 1, jj2 + 1), cidv(ii, jj2 + 1)], axis=1)
    x = np.concatenate([xo, xc])
    cells = _split_quads(x, np.concatenate([q_o, q_c]))
    mesh = Mesh(cells, x, comm=comm, name="Grid")
    mid = mesh.facet_midpoints()
    tol = 1e-9
    marker = np.full(mesh.num_facets, 5, dtype=np.int32)  # obstacle by default
    marker[np.abs(mid[:, 0]) < tol] = 2
    marker[np.abs(mid[:, 0] - L) < tol] = 3
    marker[(np.abs(mid[:, 1]) < tol) | (np.abs(mid[:, 1] - H) < tol)] = 4
    ft = MeshTags(mesh, 1, np.arange(mesh.num_facets, dtype=np.int32), marker)
    return mesh, ft


def create_stenosis_channel(ny, L=138.0, R_in=1.57, R_out=1.2, x_sten=30.0, severity=0.567, slope=0.4, tension=0.5,
                            comm=None):
    """2-D stenosed channel of /root/reference/src/scenarios/stenosis.py:262-374: 0 <= x <= L, walls
    y = R_in +- R(x) with R the linear taper R_in -> R_out and, around x_sten, the narrowing to
    (1 - severity) R_taper(x_sten) drawn by two cubic Beziers per wall (junctions at x_sten -+ h_sten/slope, handle
    length tension * dist_x along the taper slope).  Defaults = the reference's defaults with grade "moderate"
    (stenosis.py:27-31,60-69).  The outline is the reference's; the triangulation is structured (gmsh is not
    available): ny cells across, columns spaced by the local cell height 2 R(x) / ny (aspect ratio ~1 everywhere,
    also in the throat), shorter-diagonal split.  Markers (stenosis.py:22-25): inlet=2 (x=0), outlet=3 (x=L), wall=4.
    """
    from .geom.shapes import StenosedChannel
    ny = int(ny)
    ch = StenosedChannel(L, R_in, R_out, x_sten, severity, slope, tension, yc=R_in, clamp_frac=None)
    # columns: x_{i+1} = x_i + 2 R(x_i) / ny, then rescaled so that the last one lands on L
    xs = [0.0]
    while xs[-1] < L:
        xs.append(xs[-1] + 2.0 * float(ch.radius(np.array([min(xs[-1], L)]))[0]) / ny)
    xs = np.array(xs)
    # keep the stenosis position exact under the rescaling: piecewise-linear map fixing 0, x_sten and L
    k = int(np.argmin(np.abs(xs - x_sten)))
    if 0 < k < len(xs) - 1:
        left = xs[: k + 1] * (x_sten / xs[k])
        right = x_sten + (xs[k:] - xs[k]) * ((L - x_sten) / (xs[-1] - xs[k]))
        xs = np.concatenate([left, right[1:]])
    else:
        xs = xs * (L / xs[-1])
    nx = len(xs) - 1
    Rx = ch.radius(xs)
    eta = np.linspace(-1.0, 1.0, ny + 1)
    X = np.repeat(xs[:, None], ny + 1, axis=1)
    Y = R_in + Rx[:, None] * eta[None, :]
    x = np.stack([X.ravel(), Y.ravel()], axis=1)
    i, j = np.meshgrid(np.arange(nx), np.arange(ny), indexing="ij")
    v0 = (i * (ny + 1) + j).ravel()
    quads = np.stack([v0, v0 + (ny + 1), v0 + (ny + 1) + 1, v0 + 1], axis=1)
    cells = _split_quads(x, quads)
    mesh = Mesh(cells, x, comm=comm, name="stenosis")
    mid = mesh.facet_midpoints()
    marker = np.full(mesh.num_facets, 4, dtype=np.int32)
    marker[np.abs(mid[:, 0]) < 1e-9 * L] = 2
    marker[np.abs(mid[:, 0] - L) < 1e-9 * L] = 3
    ft = MeshTags(mesh, 1, np.arange(mesh.num_facets, dtype=np.int32), marker)
    mesh.channel = ch
    return mesh, ft


def create_stenosis_tree(res, L=0.03, H=0.003, x_sten=0.01, severity=0.75, slope=0.3, tension=0.5, n_generations=3,
                         gamma=3.0, bifurcation_angle=35.0, length_ratio=8.0, asymmetry=0.5, coupling_slope=0.1,
                         comm=None):
    """Domain of /root/reference/src/scenarios/stenosis_with_tree.py:146-420: the stenosed channel [0,L] x [0,H]
    (Bezier walls, :256-310), a coupling trapezoid narrowing from H to 2 r_root over (H/2 - r_root) / coupling_slope
    (:312-341), and one channel of constant half-width per tree branch around a cubic-Bezier centreline that leaves
    its start node along the parent's direction (:343-420; 12 samples, handle 0.4), all fused.  r_root = 0.9 x the
    throat half-width (:236-238).  The tree itself comes from the pure-Python Murray-law generator
    (geom/vascular_tree.py = /root/reference/src/geom/tree/tree_2d.py) in place of the external VascuSynth binary
    the reference scenario shells out to (:166-195), which does not exist here.  Meshed by the implicit-domain
    mesher (geom/implicit_mesh.py) with cell size `res`.  Markers (:37-41): inlet=2 (x=0), outlet=3 (the end caps of
    the terminal branches), wall=4.  Returns (mesh, facet_tags); the tree is kept as `mesh.tree`."""
    from .geom.implicit_mesh import keep_largest_component, mesh_implicit_domain
    from .geom.shapes import Polygon, StenosedChannel, Union, branch_polygon
    from .geom.vascular_tree import VascularTree
    ch = StenosedChannel(L, H / 2.0, H / 2.0, x_sten, severity, slope, tension, yc=H / 2.0, clamp_frac=0.05)
    r_root = 0.9 * (H / 2.0) * (1.0 - severity)
    cl = (H / 2.0 - r_root) / coupling_slope
    trap = Polygon([(L, 0.0), (L + cl, H / 2.0 - r_root), (L + cl, H / 2.0 + r_root), (L, H)])
    tree = VascularTree(r_root, n_generations, gamma, bifurcation_angle, length_ratio, asymmetry).generate((L + cl, H / 2.0), 0.0)
    din = tree.incoming_direction()
    din[0] = (1.0, 0.0)  # the root leaves the coupling along +x (:355)
    parts, caps = [ch, trap], []
    term = set(tree.terminals)
    for (a, b), r in zip(tree.edges, tree.radius):
        poly, cap = branch_polygon(tree.nodes[a], tree.nodes[b], din[a], r)
        parts.append(Polygon(poly))
        if int(b) in term:
            caps.append((np.asarray(cap[0]), np.asarray(cap[1])))
    dom = Union(parts, pad=2.0 * res)
    x0, y0, x1, y1 = dom.bbox
    bbox = (x0 - 0.5 * res, y0 - 0.37 * res, x1 + res, y1 + res)  # off-grid offsets: no boundary line exactly on a grid line
    x, cells = mesh_implicit_domain(dom.phi, bbox, res, fill=dom.fill)
    x, cells = keep_largest_component(x, cells)
    mesh = Mesh(cells, x, comm=comm, name="stenosis_with_tree")
    mid = mesh.facet_midpoints()
    marker = np.full(mesh.num_facets, 4, dtype=np.int32)
    marker[mid[:, 0] < 0.25 * res] = 2
    for p0, p1 in caps:
        e = p1 - p0
        ln = np.linalg.norm(e)
        w = mid - p0
        s = (w @ e) / (ln * ln)
        dist = np.abs(w[:, 0] * e[1] - w[:, 1] * e[0]) / ln
        marker[(dist < 0.3 * res) & (s > -0.05) & (s < 1.05)] = 3
    ft = MeshTags(mesh, 1, np.arange(mesh.num_facets, dtype=np.int32), marker)
    mesh.tree, mesh.coupling_length, mesh.r_root, mesh.outlet_caps = tree, cl, r_root, caps
    return mesh, ft
