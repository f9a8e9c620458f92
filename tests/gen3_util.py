"""Problem builders for the 3-D P2 / Q1 tests (plain arrays fed identically to oracle/np_twin_gen3.py and to libcfdh.so)."""
import numpy as np

from cfd_hemodynamic_amd.elements import NodeMesh3D, create_box
from cfd_hemodynamic_amd.mesh3d import create_unit_cube
from oracle import np_twin_gen3 as G3

ETYPE3 = {"P1": G3.P1_TET, "P2": G3.P2_TET, "Q1": G3.Q1_HEX}
LIB_ETYPE3 = {"P1": 3, "P2": 1, "Q1": 2}  # CFDH_ELEM_P1_GENERIC, _P2 (tetrahedra for gdim 3), _Q1 (hexahedra for gdim 3)


def node_mesh3(kind, n=2, distort=0.0):
    """A small mesh of a box with its node set: P1/P2 on Kuhn tetrahedra, Q1 on (sheared) bricks."""
    if kind == "Q1":
        m = create_box((0.0, 0.0, 0.0), (1.0, 0.8, 0.6), (n, n + 1, n))
        m.x[:, 0] += distort * m.x[:, 1]   # parallelepipeds
        m.x[:, 1] += 0.5 * distort * m.x[:, 2]
        return m
    base = create_unit_cube(n)
    base.x[:, 0] += distort * base.x[:, 2]
    return base if kind == "P1" else NodeMesh3D(base)


def problem3(kind, m, prm):
    return G3.Problem(ETYPE3[kind], m.x, m.cells, m.facet_cells, m.facet_local, prm)


def facet_node_set3(m, facets):
    return np.unique(np.asarray(m.facet_vertices)[np.asarray(facets, dtype=np.int64)].ravel()).astype(np.int32)
