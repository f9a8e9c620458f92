import numpy as np
import pytest

from cfd_hemodynamic_amd.mesh import (DFG_H, DFG_L, DFG_R, create_dfg_channel, create_stenosis_channel,
                                      create_unit_square, locate_entities_boundary)
from cfd_hemodynamic_amd.parallel import LocalPart, partition_vertices_rcb


def test_unit_square_right_diagonals():
    m = create_unit_square(3)
    assert m.num_vertices == 16 and m.num_cells == 18 and m.num_facets == 12
    assert np.isclose(m.cell_areas().sum(), 1.0)
    # first square split along (0,0)-(1,1): both cells contain vertices 0 and 5
    assert {0, 5} <= set(m.cells[0]) and {0, 5} <= set(m.cells[1])
    assert np.allclose(m.h(), np.sqrt(2) / 3)


@pytest.mark.parametrize("mm", [6, 18])
def test_dfg_channel(mm):
    m, ft = create_dfg_channel(mm)
    area = DFG_L * DFG_H - np.pi * DFG_R**2
    # polygonal cylinder: area error O(1/m^2)
    assert abs(m.cell_areas().sum() - area) < 0.02 / mm**2 + 1e-12 + np.pi * DFG_R**2 * (2 * np.pi / (4 * mm)) ** 2 / 6 * 1.5
    assert m.cell_areas().min() > 0
    assert set(np.unique(ft.values)) == {2, 3, 4, 5}
    obst = m.facet_vertices[ft.find(5)]
    r = np.linalg.norm(m.x[np.unique(obst)] - np.array([0.2, 0.2]), axis=1)
    assert np.allclose(r, DFG_R)
    assert len(ft.find(5)) == 4 * mm and len(ft.find(2)) == mm and len(ft.find(3)) == mm
    if mm == 18:
        assert 2500 < m.num_vertices < 3000  # ~8k DOF, the reference's coarse gmsh mesh


def test_dfg_refined_size():
    m, _ = create_dfg_channel(60)
    assert abs(m.num_vertices - 8.37 * 60**2) / (8.37 * 60**2) < 0.03


def test_locate_entities_all_vertices_rule():
    m = create_unit_square(4)
    lid = locate_entities_boundary(m, 1, lambda x: np.isclose(x[1], 1.0) & (x[0] > 1e-10) & (x[0] < 1 - 1e-10))
    # the two top facets touching the corners are excluded (lid_driven2D.py:65-67)
    assert len(lid) == 2


def test_stenosis_channel_markers():
    m, ft = create_stenosis_channel(8, L=20.0, x_sten=8.0)
    assert set(np.unique(ft.values)) == {2, 3, 4}
    assert m.cell_areas().min() > 0


@pytest.mark.parametrize("nparts", [2, 3, 4, 8])
def test_partition_and_halo_plan(nparts):
    m, _ = create_dfg_channel(8)
    owner = partition_vertices_rcb(m.x, nparts)
    counts = np.bincount(owner, minlength=nparts)
    assert counts.min() >= counts.max() - nparts  # balanced
    parts = [LocalPart(m, owner, r) for r in range(nparts)]
    assert sum(p.nvo for p in parts) == m.num_vertices
    seen_cells = np.zeros(m.num_cells, dtype=int)
    for p in parts:
        # owned rows are complete: every cell touching an owned vertex is local
        touch = (owner[m.cells] == p.rank).any(axis=1)
        assert np.array_equal(np.nonzero(touch)[0], p.cell_ids)
        seen_cells[p.cell_ids[owner[m.cells[p.cell_ids, 0]] == p.rank]] += 1
        # ghosts contiguous per neighbour, in neighbour order
        assert np.array_equal(p.recv_idx, np.arange(p.nvo, p.nv))
        assert np.all(np.diff(owner[p.ghost_global]) >= 0)
    assert np.all(seen_cells == 1)  # each cell integrated by exactly one rank (first-vertex rule)
    # send lists mirror the neighbours' receive lists
    for p in parts:
        for k, q in enumerate(p.nbr):
            sent = p.l2g[p.send_idx[p.send_ptr[k]:p.send_ptr[k + 1]]]
            qk = list(parts[q].nbr).index(p.rank)
            recv = parts[q].l2g[parts[q].recv_idx[parts[q].recv_ptr[qk]:parts[q].recv_ptr[qk + 1]]]
            assert np.array_equal(sent, recv)


def test_gmsh_reader_v41_and_roundtrip(tmp_path):
    """gmsh .msh ingest (the file form of dfg_1.py:93-181's model_to_mesh): physical line groups become
    facet tags, the physical surface the cell tag; geometry-only nodes are dropped; v2.2 round trip."""
    import os
    from cfd_hemodynamic_amd.meshio import read_msh, write_msh
    here = os.path.dirname(os.path.abspath(__file__))
    mesh, ft = read_msh(os.path.join(here, "golden", "two_triangles_v41.msh"))
    assert mesh.num_vertices == 4 and mesh.num_cells == 2 and mesh.num_facets == 4
    assert list(mesh.cell_tags) == [1, 1]
    mid = mesh.facet_midpoints()
    tag_at = {(round(float(a), 3), round(float(b), 3)): int(t) for (a, b), t in zip(mid, mesh.facet_marker)}
    # bottom and top are "walls" (4), left is "inlet" (2), right carries no physical group
    assert tag_at[(0.5, 0.0)] == 4 and tag_at[(0.5, 1.0)] == 4 and tag_at[(0.0, 0.5)] == 2 and tag_at[(1.0, 0.5)] == 0
    assert len(ft.find(4)) == 2 and len(ft.find(2)) == 1
    # a generated mesh survives write -> read (coordinates to the last bit, same tags)
    from cfd_hemodynamic_amd.mesh import create_dfg_channel
    m0, ft0 = create_dfg_channel(4)
    p = str(tmp_path / "dfg.msh")
    write_msh(p, m0, ft0)
    m1, ft1 = read_msh(p)
    assert np.array_equal(m1.x, m0.x) and np.array_equal(m1.cells, m0.cells)
    assert np.array_equal(m1.facet_marker, m0.facet_marker)
    for mk in (2, 3, 4, 5):
        assert np.array_equal(np.sort(ft1.find(mk)), np.sort(ft0.find(mk)))


def test_gmsh_reader_tetrahedra_v22_roundtrip_and_v41(tmp_path):
    """3-D `.msh` (gmshio.read_from_msh(..., gdim=3) of /root/reference/src/scenarios/simple_bifurcation.py:71-75):
    tetrahedra (type 4) + physical triangles -> Mesh3D + facet markers; 2.2 round trip and a hand-written 4.1 file."""
    from cfd_hemodynamic_amd.mesh3d import create_bifurcation
    from cfd_hemodynamic_amd.meshio import read_msh, write_msh
    mesh, ft = create_bifurcation(1.2e-3)
    p = str(tmp_path / "simple_bifurcation.msh")
    write_msh(p, mesh, ft, cell_tag=7)
    m2, ft2 = read_msh(p)
    assert m2.topology.dim == 3 and np.array_equal(m2.x, mesh.x) and m2.num_cells == mesh.num_cells
    assert (m2.cell_tags == 7).all()
    for tag in (8, 9, 10, 11):
        a = {tuple(sorted(v)) for v in mesh.facet_vertices[ft.find(tag)]}
        b = {tuple(sorted(v)) for v in m2.facet_vertices[ft2.find(tag)]}
        assert a == b and len(a) > 0
    # format 4.1: one tetrahedron, volume 1 in physical group 7, its four faces on surfaces 1..4 (physical 8, 9, 10, 11)
    q = str(tmp_path / "tet41.msh")
    with open(q, "w") as f:
        f.write("""$MeshFormat
4.1 0 8
$EndMeshFormat
$Entities
0 0 4 1
1 0 0 0 1 1 0 1 8 0
2 0 0 0 1 1 1 1 9 0
3 0 0 0 1 1 1 1 10 0
4 0 0 0 1 1 1 1 11 0
1 0 0 0 1 1 1 1 7 0
$EndEntities
$Nodes
1 4 1 4
3 1 0 4
1
2
3
4
0 0 0
1 0 0
0 1 0
0 0 1
$EndNodes
$Elements
5 5 1 5
2 1 2 1
1 1 2 3
2 2 2 1
2 1 2 4
2 3 2 1
3 1 3 4
2 4 2 1
4 2 3 4
3 1 4 1
5 1 2 3 4
$EndElements
""")
    m3, ft3 = read_msh(q)
    assert m3.num_cells == 1 and m3.num_vertices == 4 and m3.num_facets == 4 and (m3.cell_tags == 7).all()
    assert {tuple(sorted(v)) for v in m3.facet_vertices[ft3.find(8)]} == {(0, 1, 2)}
    assert {tuple(sorted(v)) for v in m3.facet_vertices[ft3.find(11)]} == {(1, 2, 3)}
    assert abs(m3.cell_volumes()[0] - 1.0 / 6.0) < 1e-15


@pytest.mark.parametrize("kind", ["P2_tri", "Q1_quad", "P2_tet", "Q1_hex"])
@pytest.mark.parametrize("nparts", [2, 3, 5])
def test_partition_of_node_meshes_of_the_generic_elements(kind, nparts):
    """Round 4: P2 / Q1 contexts take part in partitioned runs (cfdh_create_elem_part) -- the NODE mesh of the function space is
    partitioned like a vertex mesh.  Same invariants as for P1: owned rows complete, ghosts grouped by owner in receive order, every
    cell integrated by exactly one rank (first-node rule), send lists = the neighbours' receive lists."""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    if kind in ("P2_tri", "Q1_quad"):
        from gen_util import node_mesh
        m = node_mesh("P2" if kind == "P2_tri" else "Q1", 6)
    else:
        from gen3_util import node_mesh3
        m = node_mesh3("P2" if kind == "P2_tet" else "Q1", 3)
    assert m.cells.shape[1] == {"P2_tri": 6, "Q1_quad": 4, "P2_tet": 10, "Q1_hex": 8}[kind]
    owner = partition_vertices_rcb(m.x, nparts)
    parts = [LocalPart(m, owner, r) for r in range(nparts)]
    assert sum(p.nvo for p in parts) == m.num_vertices
    assert np.array_equal(np.sort(np.concatenate([p.owned_global for p in parts])), np.arange(m.num_vertices))
    seen_cells = np.zeros(m.num_cells, dtype=int)
    seen_facets = np.zeros(len(m.facet_cells), dtype=int)
    for p in parts:
        touch = (owner[m.cells] == p.rank).any(axis=1)
        assert np.array_equal(np.nonzero(touch)[0], p.cell_ids)
        assert p.cells.min() >= 0 and p.cells.max() < p.nv                      # every node of a local cell is local
        assert np.array_equal(p.l2g[p.cells], m.cells[p.cell_ids])               # local connectivity = global, renumbered
        own_cell = owner[m.cells[p.cell_ids, 0]] == p.rank
        seen_cells[p.cell_ids[own_cell]] += 1
        seen_facets[p.facet_ids[own_cell[p.facet_cells]]] += 1
        assert np.array_equal(p.recv_idx, np.arange(p.nvo, p.nv))
        assert np.all(np.diff(owner[p.ghost_global]) >= 0)
        assert np.allclose(p.x, m.x[p.l2g])
    assert np.all(seen_cells == 1) and np.all(seen_facets == 1)
    for p in parts:
        for k, q in enumerate(p.nbr):
            sent = p.l2g[p.send_idx[p.send_ptr[k]:p.send_ptr[k + 1]]]
            qk = list(parts[q].nbr).index(p.rank)
            recv = parts[q].l2g[parts[q].recv_idx[parts[q].recv_ptr[qk]:parts[q].recv_ptr[qk + 1]]]
            assert np.array_equal(sent, recv)


@pytest.mark.parametrize("layers", [2, 3])
@pytest.mark.parametrize("nparts", [2, 4, 5])
def test_partition_with_more_layers_of_overlap(nparts, layers):
    """Round 4: PartComm.make_part builds parts with TWO cell layers of overlap by default (the overlapping velocity cycle needs them on
    fine meshes).  The owned rows stay complete, every vertex of a local cell is local, the part contains the part with one layer less,
    ghosts are grouped by owner in receive order, and the send lists are the neighbours' receive lists."""
    m, _ = create_dfg_channel(8)
    owner = partition_vertices_rcb(m.x, nparts)
    parts = [LocalPart(m, owner, r, layers=layers) for r in range(nparts)]
    less = [LocalPart(m, owner, r, layers=layers - 1) for r in range(nparts)]
    assert sum(p.nvo for p in parts) == m.num_vertices
    for p, q in zip(parts, less):
        assert np.array_equal(p.owned_global, q.owned_global)
        assert set(q.cell_ids) <= set(p.cell_ids) and set(q.ghost_global) <= set(p.ghost_global)
        assert (owner[m.cells[p.cell_ids]] == p.rank).any(axis=1).sum() == (owner[m.cells] == p.rank).any(axis=1).sum()  # owned rows complete
        assert p.cells.min() >= 0 and p.cells.max() < p.nv and np.array_equal(p.l2g[p.cells], m.cells[p.cell_ids])
        assert np.array_equal(p.recv_idx, np.arange(p.nvo, p.nv)) and np.all(np.diff(owner[p.ghost_global]) >= 0)
        # exactly the cells within `layers` vertex-hops of an owned vertex
        reach = owner == p.rank
        for _ in range(layers - 1):
            reach = reach.copy()
            reach[np.unique(m.cells[reach[m.cells].any(axis=1)])] = True
        assert np.array_equal(np.nonzero(reach[m.cells].any(axis=1))[0], p.cell_ids)
    for p in parts:
        for k, q in enumerate(p.nbr):
            sent = p.l2g[p.send_idx[p.send_ptr[k]:p.send_ptr[k + 1]]]
            qk = list(parts[q].nbr).index(p.rank)
            recv = parts[q].l2g[parts[q].recv_idx[parts[q].recv_ptr[qk]:parts[q].recv_ptr[qk + 1]]]
            assert np.array_equal(sent, recv)
