"""XDMF/HDF5 mesh ingest (cfd_hemodynamic_amd/xdmf.py) in DOLFINx's layout: what the reference reads with
`XDMFFile.read_mesh(name="Grid")` / `read_meshtags(mesh, name="Facet markers")` (/root/reference/src/scenarios/dfg_1.py:43-48).
Heavy data go through libhdf5 itself (ctypes); the tests are skipped on an image without the library."""
import os

import numpy as np
import pytest

from cfd_hemodynamic_amd import xdmf
from cfd_hemodynamic_amd.mesh import MeshTags, create_dfg_channel

try:
    xdmf._h5()
except RuntimeError as e:  # pragma: no cover
    pytest.skip("libhdf5 not loadable: %s" % e, allow_module_level=True)


def test_hdf5_round_trip_of_nested_datasets(tmp_path):
    f = str(tmp_path / "a.h5")
    a = np.arange(12, dtype=np.float64).reshape(4, 3) * 0.1
    b = np.arange(10, dtype=np.int32).reshape(5, 2)
    xdmf.h5_write(f, {"/Mesh/Grid/geometry": a, "/Mesh/Grid/topology": b, "/top": np.array([3, 1, 2])})
    ra, rb = xdmf.h5_read(f, "/Mesh/Grid/geometry"), xdmf.h5_read(f, "/Mesh/Grid/topology")
    assert ra.dtype == np.float64 and np.array_equal(ra, a)
    assert rb.dtype == np.int64 and np.array_equal(rb, b)
    assert np.array_equal(xdmf.h5_read(f, "top"), [3, 1, 2])
    with pytest.raises(KeyError):
        xdmf.h5_read(f, "/Mesh/Grid/missing")
    with pytest.raises(OSError):
        xdmf.h5_read(str(tmp_path / "nope.h5"), "/x")


def test_triangle_mesh_and_facet_markers_round_trip(tmp_path):
    mesh, ft = create_dfg_channel(6)
    p = str(tmp_path / "pipe_cylinder.xdmf")
    xdmf.write_xdmf(p, mesh, ft)          # "Grid" + "Facet markers", as the reference's files are named
    assert os.path.exists(str(tmp_path / "pipe_cylinder.h5"))
    m2, ft2 = xdmf.read_xdmf(p, "Grid", "Facet markers")
    assert np.array_equal(m2.x, mesh.x)
    assert np.array_equal(np.sort(m2.cells, axis=1), np.sort(mesh.cells, axis=1))
    for tag in np.unique(ft.values):
        a = {tuple(sorted(v)) for v in mesh.facet_vertices[ft.find(tag)]}
        b = {tuple(sorted(v)) for v in m2.facet_vertices[ft2.find(tag)]}
        assert a == b and len(a) > 0
    # only some facets tagged: the others read back as 0
    some = ft.find(2)
    xdmf.write_xdmf(p, mesh, MeshTags(mesh, 1, some, np.full(len(some), 2, dtype=np.int32)))
    _, ft3 = xdmf.read_xdmf(p, "Grid", "Facet markers")
    assert set(np.unique(ft3.values)) == {0, 2} and len(ft3.find(2)) == len(some)
    with pytest.raises(KeyError):
        xdmf.read_xdmf(p, "mesh")
    # mesh alone
    m4, ft4 = xdmf.read_xdmf(p, "Grid")
    assert m4.num_cells == mesh.num_cells and not ft4.values.any()


def test_tetrahedron_mesh_round_trip(tmp_path):
    from cfd_hemodynamic_amd.mesh3d import create_bifurcation
    mesh, ft = create_bifurcation(1.2e-3)
    p = str(tmp_path / "bif.xdmf")
    xdmf.write_xdmf(p, mesh, ft, name="mesh", tags_name="mesh_tags")
    m2, ft2 = xdmf.read_xdmf(p, "mesh", "mesh_tags")
    assert m2.topology.dim == 3 and np.array_equal(m2.x, mesh.x) and m2.num_cells == mesh.num_cells
    for tag in (8, 9, 10, 11):
        assert len(ft2.find(tag)) == len(ft.find(tag)) > 0


def test_inline_xml_data_items(tmp_path):
    p = str(tmp_path / "tiny.xdmf")
    with open(p, "w") as f:
        f.write("""<?xml version="1.0"?>
<Xdmf Version="3.0"><Domain>
 <Grid Name="Grid" GridType="Uniform">
  <Topology TopologyType="Triangle" NumberOfElements="2" NodesPerElement="3">
   <DataItem Dimensions="2 3" NumberType="Int" Format="XML">0 1 2  0 2 3</DataItem></Topology>
  <Geometry GeometryType="XY"><DataItem Dimensions="4 2" Format="XML">0 0  1 0  1 1  0 1</DataItem></Geometry>
 </Grid>
 <Grid Name="Facet markers" GridType="Uniform">
  <Topology TopologyType="PolyLine" NumberOfElements="2" NodesPerElement="2">
   <DataItem Dimensions="2 2" NumberType="Int" Format="XML">0 1  2 3</DataItem></Topology>
  <Attribute Name="Facet markers" AttributeType="Scalar" Center="Cell">
   <DataItem Dimensions="2 1" NumberType="Int" Format="XML">7 9</DataItem></Attribute>
 </Grid>
</Domain></Xdmf>
""")
    mesh, ft = xdmf.read_xdmf(p, "Grid", "Facet markers")
    assert mesh.num_cells == 2 and mesh.num_vertices == 4 and mesh.num_facets == 4
    assert {tuple(sorted(v)) for v in mesh.facet_vertices[ft.find(7)]} == {(0, 1)}
    assert {tuple(sorted(v)) for v in mesh.facet_vertices[ft.find(9)]} == {(2, 3)}
    assert sorted(ft.values.tolist()) == [0, 0, 7, 9]


def test_dfg_scenario_accepts_an_xdmf_mesh_file(tmp_path, oracle_backend):
    """DFG1Benchmark(mesh_file="...xdmf") = the reference's `meshes/pipe_cylinder.xdmf` branch (dfg_1.py:42-48)."""
    from cfd_hemodynamic_amd.scenarios.dfg_1 import DFG1Benchmark
    mesh, ft = create_dfg_channel(6)
    p = str(tmp_path / "pipe_cylinder.xdmf")
    xdmf.write_xdmf(p, mesh, ft)
    a = DFG1Benchmark(oracle_backend, 0.01, 0.015, m=6, quiet=True)
    b = DFG1Benchmark(oracle_backend, 0.01, 0.015, mesh_file=p, quiet=True)
    a.solve(None); b.solve(None)
    assert abs(a.drag - b.drag) <= 1e-10 * abs(a.drag) and abs(a.norm_v - b.norm_v) <= 1e-12 * a.norm_v


@pytest.mark.gpu
def test_xdmf_ingest_on_the_device_path(tmp_path):
    """The reference's `XDMFFile.read_mesh(name="Grid")` / `read_meshtags(..., name="Facet markers")` route (dfg_1.py:43-48) on
    libcfdh.so: the DFG channel written by `write_xdmf` (XDMF + HDF5), read back and run on the GPU, against the same run on the
    generated mesh -- two steps, drag / lift / norms; and the tetrahedral bifurcation through the same reader."""
    from cfd_hemodynamic_amd.scenarios.dfg_1 import DFG1Benchmark
    from cfd_hemodynamic_amd.scenarios.simple_bifurcation import MicrovasculatureSimulation
    tight = dict(snes_rtol=1e-12, snes_stol=0.0, ksp_rtol=1e-10)
    mesh, ft = create_dfg_channel(24)
    p = str(tmp_path / "pipe_cylinder.xdmf")
    xdmf.write_xdmf(p, mesh, ft)
    a = DFG1Benchmark("stabilized_schur", 0.01, 0.015, m=24, quiet=True, options=tight)
    b = DFG1Benchmark("stabilized_schur", 0.01, 0.015, mesh_file=p, quiet=True, options=tight)
    assert b.mesh.num_vertices == a.mesh.num_vertices and b.solver.ctx.info(0) == a.mesh.num_vertices
    a.solve(None); b.solve(None)
    assert a.num_steps == b.num_steps == 2
    assert abs(a.drag - b.drag) <= 1e-9 * abs(a.drag) and abs(a.lift - b.lift) <= 1e-7 * abs(a.lift)
    assert abs(a.norm_v - b.norm_v) <= 1e-10 * a.norm_v and abs(a.norm_p - b.norm_p) <= 1e-9 * a.norm_p
    ua, ub = np.asarray(a.solver.u_sol.x.array), np.asarray(b.solver.u_sol.x.array)
    assert np.abs(ua - ub).max() <= 1e-9 * np.abs(ua).max()
    # tetrahedra (simple_bifurcation reads "mesh" / "mesh_tags")
    c3 = MicrovasculatureSimulation("stabilized_schur", 0.01, 0.015, res=8e-4, quiet=True, options=tight)
    p3 = str(tmp_path / "bif.xdmf")
    xdmf.write_xdmf(p3, c3.mesh, c3._ft, name="mesh", tags_name="mesh_tags")
    d3 = MicrovasculatureSimulation("stabilized_schur", 0.01, 0.015, mesh_file=p3, quiet=True, options=tight)
    c3.solve(None); d3.solve(None)
    assert abs(c3.norm_v - d3.norm_v) <= 1e-9 * c3.norm_v
