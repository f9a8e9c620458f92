"""Mesh ingest: gmsh `.msh` (ASCII, format 4.1 and 2.2) -> `Mesh` + facet `MeshTags`.

The reference builds its meshes with the gmsh Python API and hands them to DOLFINx through
`gmshio.model_to_mesh` (/root/reference/src/scenarios/dfg_1.py:93-181): 2-D triangles carry the
fluid physical group, boundary lines carry the inlet / outlet / wall / obstacle markers that become
`facet_tags`.  gmsh is not part of this stack; a `.msh` file written by it (`gmsh.write("m.msh")`)
carries the same information and is read here.  Only what the P1/P1 path needs is kept: first-order
triangles (element type 2) and two-node lines (type 1) with their physical tag; for a 3-D file
(`gmshio.read_from_msh(..., gdim=3)`, simple_bifurcation.py:71-75) tetrahedra (type 4) and the physical triangles.

    mesh, facet_tags = read_msh("pipe_cylinder.msh")
    facet_tags.find(2)      # exterior facets of the physical line group 2
"""
from __future__ import annotations

import numpy as np

from .mesh import Mesh, MeshTags


def _sections(path):
    sec, name, buf = {}, None, []
    with open(path, "r") as f:
        for line in f:
            s = line.strip()
            if s.startswith("$End"):
                sec[name] = buf
                name, buf = None, []
            elif s.startswith("$"):
                name, buf = s[1:], []
            elif name is not None and s:
                buf.append(s)
    return sec


def _parse_v4(sec):
    # $Entities: curve / surface tag -> first physical tag
    phys = {1: {}, 2: {}, 3: {}}
    if "Entities" in sec:
        L = sec["Entities"]
        npnt, ncur, nsur, nvol = map(int, L[0].split())
        k = 1 + npnt
        for dim, cnt in ((1, ncur), (2, nsur), (3, nvol)):
            for _ in range(cnt):
                t = L[k].split()
                k += 1
                tag, nphys = int(t[0]), int(t[7])
                if nphys:
                    phys[dim][tag] = int(t[8])
    L = sec["Nodes"]
    nblocks, nnodes = int(L[0].split()[0]), int(L[0].split()[1])
    tags, xyz = np.empty(nnodes, np.int64), np.empty((nnodes, 3))
    k, n0 = 1, 0
    for _ in range(nblocks):
        _, _, parametric, nb = map(int, L[k].split())
        k += 1
        tags[n0:n0 + nb] = [int(L[k + i]) for i in range(nb)]
        k += nb
        for i in range(nb):
            xyz[n0 + i] = [float(v) for v in L[k + i].split()[:3]]
        k += nb
        n0 += nb
    L = sec["Elements"]
    nblocks = int(L[0].split()[0])
    tris, tri_phys, lines, line_phys, tets, tet_phys = [], [], [], [], [], []
    k = 1
    for _ in range(nblocks):
        edim, etag, etype, nb = map(int, L[k].split())
        k += 1
        if etype == 4:
            for i in range(nb):
                tets.append([int(v) for v in L[k + i].split()[1:5]])
            tet_phys += [phys[3].get(etag, 0)] * nb
        elif etype == 2:
            for i in range(nb):
                tris.append([int(v) for v in L[k + i].split()[1:4]])
            tri_phys += [phys[2].get(etag, 0)] * nb
        elif etype == 1:
            for i in range(nb):
                lines.append([int(v) for v in L[k + i].split()[1:3]])
            line_phys += [phys[1].get(etag, 0)] * nb
        k += nb
    return tags, xyz, tris, tri_phys, lines, line_phys, tets, tet_phys


def _parse_v2(sec):
    L = sec["Nodes"]
    nn = int(L[0])
    tags, xyz = np.empty(nn, np.int64), np.empty((nn, 3))
    for i in range(nn):
        t = L[1 + i].split()
        tags[i] = int(t[0])
        xyz[i] = [float(v) for v in t[1:4]]
    L = sec["Elements"]
    tris, tri_phys, lines, line_phys, tets, tet_phys = [], [], [], [], [], []
    for i in range(int(L[0])):
        t = [int(v) for v in L[1 + i].split()]
        etype, ntags = t[1], t[2]
        ph = t[3] if ntags > 0 else 0
        nodes = t[3 + ntags:]
        if etype == 4:
            tets.append(nodes[:4])
            tet_phys.append(ph)
        elif etype == 2:
            tris.append(nodes[:3])
            tri_phys.append(ph)
        elif etype == 1:
            lines.append(nodes[:2])
            line_phys.append(ph)
    return tags, xyz, tris, tri_phys, lines, line_phys, tets, tet_phys


def read_msh(path, comm=None, name=None):
    """Returns (Mesh, MeshTags of the exterior facets).  Nodes not referenced by a triangle (geometry
    points, arc centres) are dropped; `mesh.cell_tags` holds the physical tag of every triangle."""
    sec = _sections(path)
    if "MeshFormat" not in sec:
        raise ValueError("%s: not a gmsh .msh file" % path)
    fmt = sec["MeshFormat"][0].split()
    if int(fmt[1]) != 0:
        raise ValueError("%s: binary .msh is not supported, write it with Mesh.Binary = 0" % path)
    version = float(fmt[0])
    tags, xyz, tris, tri_phys, lines, line_phys, tets, tet_phys = (_parse_v4 if version >= 4.0 else _parse_v2)(sec)
    if tets:
        return _mesh3d_from_msh(tags, xyz, tets, tet_phys, tris, tri_phys, comm, name)
    if not tris:
        raise ValueError("%s: no first-order triangles (element type 2) or tetrahedra (type 4)" % path)
    tris = np.asarray(tris, dtype=np.int64)
    used = np.unique(tris)
    lut = -np.ones(int(tags.max()) + 1, dtype=np.int64)
    pos = -np.ones(int(tags.max()) + 1, dtype=np.int64)
    pos[tags] = np.arange(len(tags))
    lut[used] = np.arange(len(used))
    x = xyz[pos[used], :2]
    mesh = Mesh(lut[tris].astype(np.int32), x, comm=comm, name=name or "msh")
    mesh.cell_tags = np.asarray(tri_phys, dtype=np.int32)
    # physical lines -> exterior facets
    marker = np.zeros(mesh.num_facets, dtype=np.int32)
    if lines:
        ln = lut[np.asarray(lines, dtype=np.int64)]
        ok = (ln >= 0).all(axis=1)
        ln, lp = np.sort(ln[ok], axis=1), np.asarray(line_phys, dtype=np.int32)[ok]
        key = {(int(a), int(b)): int(p) for (a, b), p in zip(ln, lp)}
        fv = np.sort(mesh.facet_vertices, axis=1)
        for f in range(mesh.num_facets):
            marker[f] = key.get((int(fv[f, 0]), int(fv[f, 1])), 0)
    ft = MeshTags(mesh, 1, np.arange(mesh.num_facets, dtype=np.int32), marker)
    return mesh, ft


def _mesh3d_from_msh(tags, xyz, tets, tet_phys, tris, tri_phys, comm, name):
    """Tetrahedra (element type 4) with their physical volume tag; physical triangles become the facet markers
    (what `gmshio.read_from_msh(..., gdim=3)` hands the reference, simple_bifurcation.py:71-75)."""
    from .mesh3d import Mesh3D
    tets = np.asarray(tets, dtype=np.int64)
    used = np.unique(tets)
    lut = -np.ones(int(tags.max()) + 1, dtype=np.int64)
    pos = -np.ones(int(tags.max()) + 1, dtype=np.int64)
    pos[tags] = np.arange(len(tags))
    lut[used] = np.arange(len(used))
    mesh = Mesh3D(lut[tets].astype(np.int32), xyz[pos[used], :3], comm=comm, name=name or "msh")
    mesh.cell_tags = np.asarray(tet_phys, dtype=np.int32)
    marker = np.zeros(mesh.num_facets, dtype=np.int32)
    if tris:
        tr = lut[np.asarray(tris, dtype=np.int64)]
        ok = (tr >= 0).all(axis=1)
        tr, tp = np.sort(tr[ok], axis=1), np.asarray(tri_phys, dtype=np.int32)[ok]
        key = {(int(a), int(b), int(d)): int(q) for (a, b, d), q in zip(tr, tp)}
        fv = np.sort(mesh.facet_vertices, axis=1)
        for f in range(mesh.num_facets):
            marker[f] = key.get((int(fv[f, 0]), int(fv[f, 1]), int(fv[f, 2])), 0)
    mesh.facet_marker[:] = marker
    return mesh, MeshTags(mesh, 2, np.arange(mesh.num_facets, dtype=np.int32), marker)


def write_msh(path, mesh, facet_tags=None, cell_tag=1):
    """`.msh` 2.2 ASCII writer (round trips with read_msh; lets a mesh built here be opened in gmsh)."""
    nv, nc = mesh.num_vertices, mesh.num_cells
    fm = mesh.facet_marker if facet_tags is None else None
    if facet_tags is not None:
        fm = np.zeros(mesh.num_facets, dtype=np.int32)
        fm[facet_tags.indices] = facet_tags.values
    sel = np.nonzero(fm != 0)[0]
    if mesh.topology.dim == 3:
        with open(path, "w") as f:
            f.write("$MeshFormat\n2.2 0 8\n$EndMeshFormat\n$Nodes\n%d\n" % nv)
            for i in range(nv):
                f.write("%d %.17g %.17g %.17g\n" % (i + 1, mesh.x[i, 0], mesh.x[i, 1], mesh.x[i, 2]))
            f.write("$EndNodes\n$Elements\n%d\n" % (len(sel) + nc))
            e = 1
            for k in sel:
                a, b, d = mesh.facet_vertices[k]
                f.write("%d 2 2 %d %d %d %d %d\n" % (e, fm[k], fm[k], a + 1, b + 1, d + 1))
                e += 1
            for c in range(nc):
                a, b, d, g = mesh.cells[c]
                f.write("%d 4 2 %d %d %d %d %d %d\n" % (e, cell_tag, cell_tag, a + 1, b + 1, d + 1, g + 1))
                e += 1
            f.write("$EndElements\n")
        return
    with open(path, "w") as f:
        f.write("$MeshFormat\n2.2 0 8\n$EndMeshFormat\n$Nodes\n%d\n" % nv)
        for i in range(nv):
            f.write("%d %.17g %.17g 0\n" % (i + 1, mesh.x[i, 0], mesh.x[i, 1]))
        f.write("$EndNodes\n$Elements\n%d\n" % (len(sel) + nc))
        e = 1
        for k in sel:
            a, b = mesh.facet_vertices[k]
            f.write("%d 1 2 %d %d %d %d\n" % (e, fm[k], fm[k], a + 1, b + 1))
            e += 1
        for c in range(nc):
            a, b, d = mesh.cells[c]
            f.write("%d 2 2 %d %d %d %d %d\n" % (e, cell_tag, cell_tag, a + 1, b + 1, d + 1))
            e += 1
        f.write("$EndElements\n")
