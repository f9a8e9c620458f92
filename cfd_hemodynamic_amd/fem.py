"""P1 function spaces and nodal functions: the attribute surface that the
reference's Scenario loop and scenario classes touch on DOLFINx objects
(SURVEY.md section 8b): `V.dofmap.index_map.size_global`,
`V.dofmap.index_map_bs`, `Function(V)`, `f.x.array`, `f.interpolate(fn)`,
`f.name`, `locate_dofs_topological`, `dirichletbc`.

P1 Lagrange on the mesh vertices; a vector space is blocked (`bs = gdim`),
dofs interleaved per vertex exactly as DOLFINx lays them out
(/root/reference/src/solvers/stabilized_schur.py:55-57).
"""
from __future__ import annotations

import numpy as np


class _IndexMap:
    def __init__(self, n):
        self.size_global = int(n)
        self.size_local = int(n)
        self.num_ghosts = 0
        self.local_range = (0, int(n))


class _DofMap:
    def __init__(self, n, bs):
        self.index_map = _IndexMap(n)
        self.index_map_bs = int(bs)


class FunctionSpace:
    """P1 space on `mesh`; `bs` = block size (gdim for velocity, 1 for pressure)."""

    def __init__(self, mesh, bs=1):
        self.mesh = mesh
        self.bs = int(bs)
        self.dofmap = _DofMap(mesh.num_vertices, bs)

    @property
    def num_dofs(self):
        return self.mesh.num_vertices * self.bs

    def tabulate_dof_coordinates(self):
        return self.mesh.geometry.x


def functionspace(mesh, element):
    """`functionspace(mesh, ("Lagrange", 1[, (gdim,)]))` mirror; only P1."""
    shape = None
    if isinstance(element, tuple):
        family, degree = element[0], element[1]
        shape = element[2] if len(element) > 2 else None
    else:
        family, degree, shape = element.family, element.degree, element.shape
    if int(degree) != 1 or str(family) not in ("Lagrange", "CG", "P"):
        raise ValueError("only P1 Lagrange spaces are supported")
    bs = 1 if not shape else int(shape[0])
    return FunctionSpace(mesh, bs)


class _Vector:
    """`Function.x`: owns the host array; `array` access can be observed by a
    device-resident solver (lazy download before a host read, host-dirty mark
    after the view was handed out)."""

    def __init__(self, n):
        self._array = np.zeros(n, dtype=np.float64)
        self._pre_access = None  # callable() -> None, run before handing out the view
        self._post_access = None

    @property
    def array(self):
        if self._pre_access is not None:
            self._pre_access()
        if self._post_access is not None:
            self._post_access()
        return self._array

    def scatter_forward(self):
        return None


class Function:
    def __init__(self, V, name="f"):
        self.function_space = V
        self.x = _Vector(V.num_dofs)
        self.name = name

    def interpolate(self, fn):
        """`fn` is a callable x[3,n] -> [bs,n] (or [n] for scalars), or a Function
        on the same space (copy), as in /root/reference/src/boundaryCondition.py:44."""
        V = self.function_space
        if isinstance(fn, Function):
            self.x.array[:] = fn.x.array
            return
        X = V.mesh.geometry.x.T
        vals = np.asarray(fn(X), dtype=np.float64)
        if V.bs == 1:
            self.x.array[:] = vals.reshape(-1)
        else:
            vals = vals.reshape(V.bs, -1)
            self.x.array[:] = vals.T.reshape(-1)

    def interpolate_at(self, fn, blocks):
        """`interpolate(fn)` restricted to the vertex blocks `blocks`: the same values there, nothing touched
        elsewhere.  Used for Dirichlet data, which are only ever read at the constrained dofs -- re-interpolating a
        whole 10^6-dof field before every step (boundaryCondition.py:48-51) costs ~0.5 ms per condition on the host."""
        V = self.function_space
        dst = self.x.array.reshape(-1, V.bs)
        if isinstance(fn, Function):
            dst[blocks] = fn.x.array.reshape(-1, V.bs)[blocks]
            return
        X = V.mesh.geometry.x.T[:, blocks]
        vals = np.asarray(fn(X), dtype=np.float64)
        dst[blocks] = vals.reshape(-1, 1) if V.bs == 1 else vals.reshape(V.bs, -1).T

    def vector_values(self):
        """[nv, bs] view."""
        return self.x._array.reshape(-1, self.function_space.bs)


class Constant:
    def __init__(self, mesh, value):
        self.value = np.asarray(value, dtype=np.float64)

    def __float__(self):
        return float(self.value)


def locate_dofs_topological(V, entity_dim, entities):
    """Vertex (block) indices of the given exterior facets."""
    mesh = V.mesh
    assert entity_dim == mesh.topology.dim - 1
    ent = np.asarray(entities, dtype=np.int64)
    return np.unique(mesh.facet_vertices[ent].ravel()).astype(np.int32)


def locate_dofs_geometrical(V, marker):
    X = V.mesh.geometry.x.T
    return np.nonzero(np.asarray(marker(X), dtype=bool))[0].astype(np.int32)


class DirichletBC:
    """Values taken from Function `g` at vertex blocks `dofs` (all components)."""

    def __init__(self, g, dofs):
        self.g = g
        self.dofs = np.asarray(dofs, dtype=np.int32)
        self.function_space = g.function_space

    def update(self):  # constant data; boundaryCondition._SourcedBC refreshes from its source
        return None


def dirichletbc(g, dofs):
    return DirichletBC(g, dofs)
