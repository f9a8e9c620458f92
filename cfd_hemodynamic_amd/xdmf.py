"""XDMF (+HDF5) mesh ingest in the layout DOLFINx writes and reads.

The reference loads its pre-built meshes with

    with XDMFFile(comm, "meshes/pipe_cylinder.xdmf", "r") as xdmf:
        mesh = xdmf.read_mesh(name="Grid")
        ft = xdmf.read_meshtags(mesh, name="Facet markers")

(/root/reference/src/scenarios/dfg_1.py:43-48, pipe_cylinder.py:39-44).  `read_xdmf(path, name, tags_name)` does the same
for P1 triangle and tetrahedron meshes: the `.xdmf` file is XML whose `DataItem`s point into a sibling `.h5` file
("mesh.h5:/Mesh/Grid/topology"; inline `Format="XML"` items are read too).  There is no h5py in this stack; the heavy data
are read through the HDF5 C library itself (libhdf5, bound with ctypes: H5Fopen / H5Dopen2 / H5Dread).  `write_xdmf`
produces files of the same layout (used by the tests and to hand meshes generated here to DOLFINx / ParaView).
"""
from __future__ import annotations

import ctypes as C
import ctypes.util
import glob
import os
import xml.etree.ElementTree as ET

import numpy as np

from .mesh import Mesh, MeshTags

_H5 = None
_CANDIDATES = ("/opt/conda/lib/libhdf5.so*", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so*",
               "/usr/lib/x86_64-linux-gnu/libhdf5*.so*", "/usr/lib64/libhdf5.so*")


def _h5():
    """The HDF5 C library (>= 1.10: 64-bit hid_t)."""
    global _H5
    if _H5 is not None:
        return _H5
    paths = []
    found = ctypes.util.find_library("hdf5")
    if found:
        paths.append(found)
    if os.environ.get("CFDH_HDF5_LIB"):
        paths.insert(0, os.environ["CFDH_HDF5_LIB"])
    for pat in _CANDIDATES:
        paths += sorted(p for p in glob.glob(pat) if "_hl" not in p and "fortran" not in p and "cpp" not in p)
    err = None
    for p in paths:
        try:
            L = C.CDLL(p)
            L.H5open()
            break
        except OSError as e:  # pragma: no cover - depends on the image
            err = e
    else:
        raise RuntimeError("XDMF heavy data need the HDF5 C library (libhdf5.so); none could be loaded"
                           " (set CFDH_HDF5_LIB). Last error: %s" % err)
    hid = C.c_int64
    L.H5Fopen.restype = hid; L.H5Fopen.argtypes = [C.c_char_p, C.c_uint, hid]
    L.H5Fcreate.restype = hid; L.H5Fcreate.argtypes = [C.c_char_p, C.c_uint, hid, hid]
    L.H5Fclose.argtypes = [hid]
    L.H5Dopen2.restype = hid; L.H5Dopen2.argtypes = [hid, C.c_char_p, hid]
    L.H5Dclose.argtypes = [hid]
    L.H5Dget_space.restype = hid; L.H5Dget_space.argtypes = [hid]
    L.H5Dget_type.restype = hid; L.H5Dget_type.argtypes = [hid]
    L.H5Sget_simple_extent_ndims.argtypes = [hid]
    L.H5Sget_simple_extent_dims.argtypes = [hid, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.H5Sclose.argtypes = [hid]
    L.H5Tget_class.argtypes = [hid]
    L.H5Tclose.argtypes = [hid]
    L.H5Dread.argtypes = [hid, hid, hid, hid, hid, C.c_void_p]
    L.H5Dwrite.argtypes = [hid, hid, hid, hid, hid, C.c_void_p]
    L.H5Screate_simple.restype = hid; L.H5Screate_simple.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.H5Dcreate2.restype = hid; L.H5Dcreate2.argtypes = [hid, C.c_char_p, hid, hid, hid, hid, hid]
    L.H5Gcreate2.restype = hid; L.H5Gcreate2.argtypes = [hid, C.c_char_p, hid, hid, hid]
    L.H5Gclose.argtypes = [hid]
    L.H5Lexists.argtypes = [hid, C.c_char_p, hid]
    L.H5Eset_auto2.argtypes = [hid, C.c_void_p, C.c_void_p]
    L.H5Eset_auto2(0, None, None)  # errors are reported through return codes / exceptions, not on stderr
    L._f64 = C.c_int64.in_dll(L, "H5T_NATIVE_DOUBLE_g").value
    L._i64 = C.c_int64.in_dll(L, "H5T_NATIVE_INT64_g").value
    _H5 = L
    return L


def h5_read(path, dataset):
    """One dataset of an HDF5 file as float64 (floating-point data) or int64 (integer data)."""
    L = _h5()
    f = L.H5Fopen(os.fsencode(path), 0, 0)
    if f < 0:
        raise OSError("cannot open HDF5 file %s" % path)
    try:
        d = L.H5Dopen2(f, dataset.encode(), 0)
        if d < 0:
            raise KeyError("%s: no dataset %s" % (path, dataset))
        try:
            sp, ty = L.H5Dget_space(d), L.H5Dget_type(d)
            nd = L.H5Sget_simple_extent_ndims(sp)
            dims = (C.c_uint64 * max(nd, 1))()
            L.H5Sget_simple_extent_dims(sp, dims, None)
            shape = tuple(int(dims[i]) for i in range(nd))
            cls = L.H5Tget_class(ty)  # 0: integer, 1: float
            L.H5Sclose(sp); L.H5Tclose(ty)
            if cls not in (0, 1):
                raise TypeError("%s:%s is neither integer nor floating point" % (path, dataset))
            out = np.empty(shape, dtype=np.float64 if cls == 1 else np.int64)
            if L.H5Dread(d, L._f64 if cls == 1 else L._i64, 0, 0, 0, out.ctypes.data_as(C.c_void_p)) < 0:
                raise OSError("H5Dread failed on %s:%s" % (path, dataset))
            return out
        finally:
            L.H5Dclose(d)
    finally:
        L.H5Fclose(f)


def h5_write(path, datasets):
    """Write {"/group/name": array} (float64 / int64, contiguous layout) into a new HDF5 file."""
    L = _h5()
    f = L.H5Fcreate(os.fsencode(path), 2, 0, 0)  # H5F_ACC_TRUNC
    if f < 0:
        raise OSError("cannot create HDF5 file %s" % path)
    try:
        for name, arr in datasets.items():
            parts = [q for q in name.split("/") if q]
            for k in range(1, len(parts)):
                g = "/" + "/".join(parts[:k])
                if L.H5Lexists(f, g.encode(), 0) <= 0:
                    L.H5Gclose(L.H5Gcreate2(f, g.encode(), 0, 0, 0))
            a = np.asarray(arr)
            isf = a.dtype.kind == "f"
            a = np.ascontiguousarray(a, dtype=np.float64 if isf else np.int64)
            dims = (C.c_uint64 * a.ndim)(*a.shape)
            sp = L.H5Screate_simple(a.ndim, dims, None)
            d = L.H5Dcreate2(f, ("/" + "/".join(parts)).encode(), L._f64 if isf else L._i64, sp, 0, 0, 0)
            if d < 0 or L.H5Dwrite(d, L._f64 if isf else L._i64, 0, 0, 0, a.ctypes.data_as(C.c_void_p)) < 0:
                raise OSError("cannot write %s:%s" % (path, name))
            L.H5Dclose(d); L.H5Sclose(sp)
    finally:
        L.H5Fclose(f)


def _local(tag):
    return tag.rsplit("}", 1)[-1]


def _data_item(node, base):
    item = next(ch for ch in node if _local(ch.tag) == "DataItem")
    fmt = item.get("Format", "XML").upper()
    dims = tuple(int(v) for v in item.get("Dimensions", "").split())
    text = (item.text or "").strip()
    if fmt == "HDF":
        fname, dset = text.split(":", 1)
        arr = h5_read(os.path.join(base, fname), dset)
    elif fmt == "XML":
        kind = item.get("NumberType", item.get("DataType", "Float")).lower()
        arr = np.array(text.split(), dtype=np.int64 if kind in ("int", "uint") else np.float64)
    else:
        raise ValueError("DataItem format %r is not supported (HDF or XML)" % fmt)
    return arr.reshape(dims) if dims and int(np.prod(dims)) == arr.size else arr


_CELL_NODES = {"triangle": 3, "tetrahedron": 4}
_FACET_NODES = {"polyline": 2, "triangle": 3}


def _grids(root):
    return [g for g in root.iter() if _local(g.tag) == "Grid" and g.get("GridType", "Uniform") == "Uniform"]


def read_xdmf(path, name="Grid", tags_name=None, comm=None):
    """`XDMFFile.read_mesh(name=...)` and, when `tags_name` is given, `read_meshtags(mesh, name=tags_name)`.
    Returns (mesh, facet MeshTags over the exterior facets); facets the tags grid does not list keep marker 0."""
    base = os.path.dirname(os.path.abspath(path))
    grids = {g.get("Name"): g for g in _grids(ET.parse(path).getroot())}
    if name not in grids:
        raise KeyError("%s: no grid named %r (found %s)" % (path, name, sorted(grids)))
    g = grids[name]
    topo = next(ch for ch in g if _local(ch.tag) == "Topology")
    geom = next(ch for ch in g if _local(ch.tag) == "Geometry")
    ctype = topo.get("TopologyType", "").lower()
    if ctype not in _CELL_NODES:
        raise ValueError("%s: cell type %r is not supported (first-order triangles and tetrahedra only)" % (path, ctype))
    cells = np.asarray(_data_item(topo, base), dtype=np.int64).reshape(-1, _CELL_NODES[ctype])
    x = np.asarray(_data_item(geom, base), dtype=np.float64)
    x = x.reshape(-1, x.shape[-1] if x.ndim == 2 else (3 if geom.get("GeometryType", "XYZ").upper() == "XYZ" else 2))
    if ctype == "triangle":
        mesh = Mesh(cells.astype(np.int32), x[:, :2], comm=comm, name=name)
    else:
        from .mesh3d import Mesh3D
        mesh = Mesh3D(cells.astype(np.int32), x[:, :3], comm=comm, name=name)
    marker = np.zeros(mesh.num_facets, dtype=np.int32)
    if tags_name is not None:
        if tags_name not in grids:
            raise KeyError("%s: no grid named %r (found %s)" % (path, tags_name, sorted(grids)))
        tg = grids[tags_name]
        ttopo = next(ch for ch in tg if _local(ch.tag) == "Topology")
        ftype = ttopo.get("TopologyType", "").lower()
        nn = _FACET_NODES.get(ftype)
        if nn != mesh.topology.dim:
            raise ValueError("%s: %r tags are not facet tags of a %s mesh" % (path, ftype, ctype))
        fverts = np.sort(np.asarray(_data_item(ttopo, base), dtype=np.int64).reshape(-1, nn), axis=1)
        attr = next(ch for ch in tg if _local(ch.tag) == "Attribute")
        vals = np.asarray(_data_item(attr, base)).reshape(-1).astype(np.int32)
        key = {tuple(int(q) for q in fv): int(v) for fv, v in zip(fverts, vals)}
        mine = np.sort(mesh.facet_vertices, axis=1)
        for k in range(mesh.num_facets):
            marker[k] = key.get(tuple(int(q) for q in mine[k]), 0)
    mesh.facet_marker[:] = marker
    ft = MeshTags(mesh, mesh.topology.dim - 1, np.arange(mesh.num_facets, dtype=np.int32), marker)
    return mesh, ft


def write_xdmf(path, mesh, facet_tags=None, name="Grid", tags_name="Facet markers"):
    """Mesh (+ facet tags) as `<path>` and `<path minus .xdmf>.h5` in DOLFINx's layout."""
    stem = os.path.splitext(path)[0]
    h5name = os.path.basename(stem) + ".h5"
    tdim = mesh.topology.dim
    gd = mesh.geometry.dim
    data = {"/Mesh/%s/topology" % name: np.asarray(mesh.cells, dtype=np.int64),
            "/Mesh/%s/geometry" % name: np.asarray(mesh.x, dtype=np.float64)[:, :gd]}
    cname = "Triangle" if tdim == 2 else "Tetrahedron"
    xml = ['<?xml version="1.0"?>', '<Xdmf Version="3.0" xmlns:xi="https://www.w3.org/2001/XInclude">', " <Domain>",
           '  <Grid Name="%s" GridType="Uniform">' % name,
           '   <Topology TopologyType="%s" NumberOfElements="%d" NodesPerElement="%d">' % (cname, mesh.num_cells, tdim + 1),
           '    <DataItem Dimensions="%d %d" NumberType="Int" Format="HDF">%s:/Mesh/%s/topology</DataItem>' % (
               mesh.num_cells, tdim + 1, h5name, name),
           "   </Topology>",
           '   <Geometry GeometryType="%s">' % ("XY" if gd == 2 else "XYZ"),
           '    <DataItem Dimensions="%d %d" Format="HDF">%s:/Mesh/%s/geometry</DataItem>' % (mesh.num_vertices, gd, h5name, name),
           "   </Geometry>", "  </Grid>"]
    if facet_tags is not None:
        idx = np.asarray(facet_tags.indices, dtype=np.int64)
        fv = np.asarray(mesh.facet_vertices, dtype=np.int64)[idx]
        data["/MeshTags/%s/topology" % tags_name] = fv
        data["/MeshTags/%s/Values" % tags_name] = np.asarray(facet_tags.values, dtype=np.int64)
        fname = "PolyLine" if tdim == 2 else "Triangle"
        xml += ['  <Grid Name="%s" GridType="Uniform">' % tags_name,
                '   <xi:include xpointer="xpointer(/Xdmf/Domain/Grid/Geometry)" />',
                '   <Topology TopologyType="%s" NumberOfElements="%d" NodesPerElement="%d">' % (fname, len(idx), tdim),
                '    <DataItem Dimensions="%d %d" NumberType="Int" Format="HDF">%s:/MeshTags/%s/topology</DataItem>' % (
                    len(idx), tdim, h5name, tags_name),
                "   </Topology>",
                '   <Attribute Name="%s" AttributeType="Scalar" Center="Cell">' % tags_name,
                '    <DataItem Dimensions="%d 1" NumberType="Int" Format="HDF">%s:/MeshTags/%s/Values</DataItem>' % (
                    len(idx), h5name, tags_name),
                "   </Attribute>", "  </Grid>"]
    xml += [" </Domain>", "</Xdmf>"]
    h5_write(stem + ".h5", data)
    with open(path, "w") as f:
        f.write("\n".join(xml) + "\n")
