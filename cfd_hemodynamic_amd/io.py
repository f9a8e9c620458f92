"""Time-series output readable by ParaView: the role of the ADIOS2 `VTXWriter`s of
/root/reference/src/scenario.py:208-228,258-263 (v.bp, p.bp, u_residual.bp, p_residual.bp, wss.bp).
ADIOS2 is not part of this stack; each writer produces `<name>.pvd` (the time index) plus one
`<name>_NNNNNN.vtu` per written step: P1 triangles, the field as point data, raw-appended binary.

Same call pattern as the reference: `w = VTUWriter(comm, path, function); w.write(t); w.close()`.
In a partitioned run rank 0 writes the global field (the state Functions hold it on every rank).
"""
from __future__ import annotations

import os
import struct

import numpy as np


class VTUWriter:
    def __init__(self, comm, filename: str, function, name: str = None):
        self.comm = comm
        base = filename[:-4] if filename.endswith((".pvd", ".vtu")) else filename
        if base.endswith(".bp"):
            base = base[:-3]  # accept the reference's file names
        self.base = base
        self.function = function
        self.name = name or getattr(function, "name", None) or os.path.basename(base)
        self.steps = []  # (t, file)
        self.mesh = function.function_space.mesh

    def write(self, t: float) -> None:
        vals = np.array(self.function.x.array, dtype=np.float64, copy=True)  # collective in partitioned runs
        if self.comm.rank != 0:
            return
        m = self.mesh
        nv, nc = m.num_vertices, len(m.cells)
        bs = vals.size // nv
        pts = np.zeros((nv, 3))
        pts[:, : m.x.shape[1]] = m.x
        data = vals.reshape(nv, bs)
        if bs == 2:  # ParaView wants 3-component vectors
            data = np.concatenate([data, np.zeros((nv, 1))], axis=1)
        npc = m.cells.shape[1]  # 3: VTK_TRIANGLE (5), 4: VTK_TETRA (10) / VTK_QUAD (9) in 2-D, 6: VTK_QUADRATIC_TRIANGLE (22)
        conn, ctype = np.ascontiguousarray(m.cells, dtype=np.int32), (5 if npc == 3 else 10)
        if npc == 4 and m.x.shape[1] == 2:
            conn, ctype = conn[:, [0, 1, 3, 2]], 9          # DOLFINx tensor-product order -> VTK's cyclic order
        elif npc == 6:
            conn, ctype = conn[:, [0, 1, 2, 5, 3, 4]], 22   # edge nodes: VTK wants (0-1), (1-2), (2-0); ours are opposite vertex 0, 1, 2
        elif npc == 8:
            conn, ctype = conn[:, [0, 1, 3, 2, 4, 5, 7, 6]], 12   # VTK_HEXAHEDRON: bottom face cyclic, then the top face
        elif npc == 10:
            # VTK_QUADRATIC_TETRA: edges (0-1) (1-2) (0-2) (0-3) (1-3) (2-3); ours (Basix): (2,3) (1,3) (1,2) (0,3) (0,2) (0,1)
            conn, ctype = conn[:, [0, 1, 2, 3, 9, 6, 8, 7, 5, 4]], 24
        arrays = [pts.ravel(), np.ascontiguousarray(conn).ravel(),
                  (npc * np.arange(1, nc + 1)).astype(np.int32), np.full(nc, ctype, dtype=np.uint8), data.ravel()]
        offs, blob = [], bytearray()
        for a in arrays:
            offs.append(len(blob))
            raw = np.ascontiguousarray(a).tobytes()
            blob += struct.pack("<Q", len(raw)) + raw
        ncomp = data.shape[1]
        fn = "%s_%06d.vtu" % (self.base, len(self.steps))
        head = (
            '<?xml version="1.0"?>\n'
            '<VTKFile type="UnstructuredGrid" version="1.0" byte_order="LittleEndian" header_type="UInt64">\n'
            '<UnstructuredGrid><Piece NumberOfPoints="%d" NumberOfCells="%d">\n'
            '<Points><DataArray type="Float64" NumberOfComponents="3" format="appended" offset="%d"/></Points>\n'
            '<Cells>\n<DataArray type="Int32" Name="connectivity" format="appended" offset="%d"/>\n'
            '<DataArray type="Int32" Name="offsets" format="appended" offset="%d"/>\n'
            '<DataArray type="UInt8" Name="types" format="appended" offset="%d"/>\n</Cells>\n'
            '<PointData><DataArray type="Float64" Name="%s" NumberOfComponents="%d" format="appended" offset="%d"/></PointData>\n'
            '</Piece></UnstructuredGrid>\n<AppendedData encoding="raw">\n_'
            % (nv, nc, offs[0], offs[1], offs[2], offs[3], self.name, ncomp, offs[4]))
        with open(fn, "wb") as f:
            f.write(head.encode())
            f.write(bytes(blob))
            f.write(b"\n</AppendedData>\n</VTKFile>\n")
        self.steps.append((float(t), os.path.basename(fn)))
        self._write_index()

    def _write_index(self):
        with open(self.base + ".pvd", "w") as f:
            f.write('<?xml version="1.0"?>\n<VTKFile type="Collection" version="0.1" byte_order="LittleEndian">\n<Collection>\n')
            for t, fn in self.steps:
                f.write('<DataSet timestep="%.12g" part="0" file="%s"/>\n' % (t, fn))
            f.write("</Collection>\n</VTKFile>\n")

    def close(self) -> None:
        if self.comm.rank == 0 and self.steps:
            self._write_index()


def read_vtu(path):
    """Minimal reader of the files written above (tests, post-processing scripts):
    returns dict(points, cells, name -> array)."""
    import re
    raw = open(path, "rb").read()
    k = raw.index(b'<AppendedData encoding="raw">')
    head = raw[:k].decode()
    start = raw.index(b"_", k) + 1
    out = {}
    np_, nc_ = map(int, re.search(r'NumberOfPoints="(\d+)" NumberOfCells="(\d+)"', head).groups())
    for m in re.finditer(r'<DataArray type="(\w+)"(?: Name="([^"]*)")?(?: NumberOfComponents="(\d+)")? format="appended" offset="(\d+)"/>', head):
        typ, name, ncomp, off = m.group(1), m.group(2), m.group(3), int(m.group(4))
        n = struct.unpack_from("<Q", raw, start + off)[0]
        dt = {"Float64": np.float64, "Int32": np.int32, "UInt8": np.uint8}[typ]
        a = np.frombuffer(raw, dtype=dt, count=n // np.dtype(dt).itemsize, offset=start + off + 8)
        if ncomp:
            a = a.reshape(-1, int(ncomp))
        out[name or "points"] = a
    out["cells"] = out.pop("connectivity").reshape(nc_, -1)
    assert len(out["points"]) == np_
    return out
