"""Affine tetrahedral meshes: the 3-D counterpart of mesh.py (same attribute surface: `topology.dim`,
`geometry.dim`, `geometry.x`, `comm`, `h`, exterior facets with (cell, local index of the opposite vertex), facet
markers) plus deterministic generators.  The reference reads its 3-D meshes from gmsh files
(/root/reference/src/scenarios/simple_bifurcation.py:71-75, scenario_factory.py:47-49); gmsh does not exist on the GPU
box, so the bifurcation is generated here (voxel tetrahedra of an implicit Y-shaped vessel)."""
from __future__ import annotations

import numpy as np

from .mesh import MeshTags, _SerialComm

# Kuhn subdivision of the unit cube into six tetrahedra sharing the diagonal 0-7 (corner k = (k&1, k>>1&1, k>>2&1)):
# conforming across neighbouring cubes, all of positive orientation
_KUHN = np.array([[0, 1, 3, 7], [0, 3, 2, 7], [0, 2, 6, 7], [0, 6, 4, 7], [0, 4, 5, 7], [0, 5, 1, 7]])


class _Topology3:
    dim = 3

    def cell_name(self):
        return "tetrahedron"

    def create_connectivity(self, d0, d1):
        return None


class _Geometry3:
    dim = 3

    def __init__(self, mesh):
        self._mesh = mesh

    @property
    def x(self):
        return self._mesh.x


class Mesh3D:
    """cells int32 [nc,4] (positively oriented), x float64 [nv,3].  Exterior facets = triangles with exactly one
    adjacent cell: facet_cells[f], facet_local[f] = local index of the vertex OPPOSITE to the facet, facet_vertices[f]."""

    def __init__(self, cells, x, comm=None, name="mesh"):
        cells = np.ascontiguousarray(cells, dtype=np.int32).copy()
        x = np.ascontiguousarray(x, dtype=np.float64)[:, :3].copy()
        assert cells.ndim == 2 and cells.shape[1] == 4
        p = x[cells]
        det = np.linalg.det(np.transpose(p[:, 1:] - p[:, :1], (0, 2, 1)))
        if np.any(det == 0.0):
            raise ValueError("degenerate cell in mesh")
        flip = det < 0
        if flip.any():
            cells[flip, 2], cells[flip, 3] = cells[flip, 3].copy(), cells[flip, 2].copy()
        self.cells, self.x, self.name = cells, x, name
        self.comm = comm or _SerialComm()
        self.topology = _Topology3()
        self.geometry = _Geometry3(self)
        self._build_facets()

    num_vertices = property(lambda self: self.x.shape[0])
    num_cells = property(lambda self: self.cells.shape[0])
    num_facets = property(lambda self: len(self.facet_cells))

    def _build_facets(self):
        c = self.cells.astype(np.int64)
        nc = len(c)
        # local facet i is opposite local vertex i
        loc = [[1, 2, 3], [0, 3, 2], [0, 1, 3], [0, 2, 1]]
        tri = np.stack([c[:, l] for l in loc], axis=1).reshape(-1, 3)
        srt = np.sort(tri, axis=1)
        nv1 = self.num_vertices + 1
        key = (srt[:, 0] * nv1 + srt[:, 1]) * nv1 + srt[:, 2]
        order = np.argsort(key, kind="stable")
        ks = key[order]
        first = np.ones(len(ks), bool)
        first[1:] = ks[1:] != ks[:-1]
        last = np.ones(len(ks), bool)
        last[:-1] = ks[1:] != ks[:-1]
        single = np.sort(order[first & last])
        self.facet_cells = (single // 4).astype(np.int32)
        self.facet_local = (single % 4).astype(np.int32)
        self.facet_vertices = tri[single].astype(np.int32)
        self.facet_marker = np.zeros(len(single), dtype=np.int32)
        assert nc > 0

    def facet_midpoints(self):
        return self.x[self.facet_vertices].mean(axis=1)

    def h(self, dim=3, entities=None):
        c = self.cells if entities is None else self.cells[np.asarray(entities)]
        p = self.x[c]
        out = np.zeros(len(c))
        for a in range(4):
            for b in range(a + 1, 4):
                out = np.maximum(out, np.linalg.norm(p[:, a] - p[:, b], axis=1))
        return out

    def cell_volumes(self):
        p = self.x[self.cells]
        return np.abs(np.linalg.det(np.transpose(p[:, 1:] - p[:, :1], (0, 2, 1)))) / 6.0

    def set_facet_markers(self, facets, values):
        self.facet_marker[np.asarray(facets, dtype=np.int64)] = np.asarray(values, dtype=np.int32)


def _voxel_tets(keep, xs, ys, zs):
    """Tetrahedra (Kuhn) of the kept voxels keep[i,j,k] of the tensor grid xs x ys x zs; unused vertices dropped."""
    nx, ny, nz = len(xs) - 1, len(ys) - 1, len(zs) - 1
    ii, jj, kk = np.nonzero(keep)
    sx, sy = (ny + 1) * (nz + 1), (nz + 1)
    base = ii * sx + jj * sy + kk
    corner = np.stack([base + (c & 1) * sx + ((c >> 1) & 1) * sy + ((c >> 2) & 1) for c in range(8)], axis=1)
    cells = corner[:, _KUHN].reshape(-1, 4)
    used = np.unique(cells)
    remap = np.full((nx + 1) * (ny + 1) * (nz + 1), -1, dtype=np.int64)
    remap[used] = np.arange(len(used))
    X, Y, Z = np.meshgrid(xs, ys, zs, indexing="ij")
    pts = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)[used]
    return remap[cells].astype(np.int32), pts


def create_unit_cube(n, comm=None):
    """[0,1]^3 with n^3 cubes of six tetrahedra each."""
    t = np.linspace(0.0, 1.0, n + 1)
    cells, pts = _voxel_tets(np.ones((n, n, n), bool), t, t, t)
    return Mesh3D(cells, pts, comm=comm, name="unit_cube")


def create_bifurcation(res, r_in=0.003918604, r_out=None, parent_len=None, daughter_len=None, half_angle=35.0, comm=None):
    """Y-shaped vessel: parent tube of radius r_in along +y from the inlet disc at y = 0 (the reference's inlet profile is
    `u_y = v (1 - (r / r_in)^2)`, r^2 = x^2 + z^2, simple_bifurcation.py:123-133), splitting in the x-y plane into two
    daughter tubes of radius r_out (default: Murray's law, r_in 2^(-1/3)) at +-half_angle degrees.  Voxel tetrahedra of
    size `res`: a cube is kept when its centre lies inside the union of the three capsule-shaped tubes; the daughters are cut
    by the plane y = y_end, so inlet and outlets are planar.  Facet markers as in the reference (:14-18): inlet 8,
    outlets 9 (x > 0) and 10 (x < 0), walls 11.  Returns (mesh, facet_tags)."""
    r_out = r_in * 2.0 ** (-1.0 / 3.0) if r_out is None else float(r_out)
    Lp = 4.0 * r_in if parent_len is None else float(parent_len)
    Ld = 5.0 * r_in if daughter_len is None else float(daughter_len)
    th = np.radians(half_angle)
    y_end = Lp + Ld * np.cos(th)
    xmax = Ld * np.sin(th) + r_out / np.cos(th) + res
    nxh = int(np.ceil(xmax / res))
    nzh = int(np.ceil((r_in + 0.5 * res) / res))
    ny = int(round(y_end / res))
    xs = res * np.arange(-nxh, nxh + 1)
    zs = res * np.arange(-nzh, nzh + 1)
    ys = np.linspace(0.0, ny * res, ny + 1)
    xc, yc, zc = 0.5 * (xs[1:] + xs[:-1]), 0.5 * (ys[1:] + ys[:-1]), 0.5 * (zs[1:] + zs[:-1])
    X, Y, Z = np.meshgrid(xc, yc, zc, indexing="ij")
    X = np.transpose(X, (0, 1, 2))

    def dist_to_segment(a, b):
        ab = b - a
        t = np.clip(((X - a[0]) * ab[0] + (Y - a[1]) * ab[1]) / (ab @ ab), 0.0, 1.0)
        return np.sqrt((X - a[0] - t * ab[0]) ** 2 + (Y - a[1] - t * ab[1]) ** 2 + Z ** 2)

    J = np.array([0.0, Lp])
    keep = dist_to_segment(np.array([0.0, -r_in]), J) < r_in
    for sgn in (1.0, -1.0):
        e = J + (Ld + 2.0 * r_out) * np.array([sgn * np.sin(th), np.cos(th)])  # runs past y_end: the plane cuts it
        keep |= dist_to_segment(J, e) < r_out
    # meshgrid(indexing="ij") over (xc, yc, zc) gives arrays [nx, ny, nz]
    cells, pts = _voxel_tets(keep, xs, ys, zs)
    mesh = Mesh3D(cells, pts, comm=comm, name="simple_bifurcation")
    mid = mesh.facet_midpoints()
    marker = np.full(mesh.num_facets, 11, dtype=np.int32)
    marker[np.abs(mid[:, 1]) < 1e-9 * max(y_end, 1.0) + 1e-12] = 8
    top = np.abs(mid[:, 1] - ys[-1]) < 1e-9 * y_end
    marker[top & (mid[:, 0] > 0)] = 9
    marker[top & (mid[:, 0] < 0)] = 10
    ft = MeshTags(mesh, 2, np.arange(mesh.num_facets, dtype=np.int32), marker)
    mesh.r_in, mesh.r_out, mesh.y_end = r_in, r_out, float(ys[-1])
    return mesh, ft
