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


def _is_whole(key):
    return isinstance(key, slice) and key == slice(None, None, None) or key is Ellipsis


class _LazyArray(np.lib.mixins.NDArrayOperatorsMixin):
    """What `Function.x.array` hands out for a field that lives in HBM: an array-like that touches the host copy
    only when its data are really needed (then it downloads / marks the host copy dirty exactly like a plain
    access did before).  The one idiom it keeps on the device is the reference's state copy

        u_prev.x.array[:] = u_sol.x.array[:]          (/root/reference/src/scenario.py:306-307)

    a whole-array assignment from a sibling field, which becomes a device-to-device copy (cfdh_advance_field)
    instead of a 24 MB round trip over PCIe per field and step.  Everything else (indexing, ufuncs, numpy functions,
    ndarray methods) sees the synchronised host array."""

    __slots__ = ("_vec",)

    def __init__(self, vec):
        object.__setattr__(self, "_vec", vec)

    def _real(self):
        return self._vec._materialise()

    # --- cheap metadata: no synchronisation
    shape = property(lambda self: self._vec._array.shape)
    dtype = property(lambda self: self._vec._array.dtype)
    size = property(lambda self: self._vec._array.size)
    ndim = property(lambda self: 1)

    def __len__(self):
        return len(self._vec._array)

    # --- data access
    def __array__(self, dtype=None, copy=None):
        a = self._real()
        if dtype is not None and np.dtype(dtype) != a.dtype:
            return a.astype(dtype)
        return a.copy() if copy else a

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        conv = lambda v: v._real() if isinstance(v, _LazyArray) else v
        if "out" in kwargs:
            kwargs["out"] = tuple(conv(o) for o in kwargs["out"])
        return getattr(ufunc, method)(*(conv(v) for v in inputs), **kwargs)

    def __getitem__(self, key):
        if _is_whole(key):
            return _LazyArray(self._vec)  # `a[:]` of a whole field: still lazy
        return self._real()[key]

    def __setitem__(self, key, value):
        if _is_whole(key) and isinstance(value, _LazyArray) and self._vec._assign_on_device(value._vec):
            return
        self._real()[key] = value._real() if isinstance(value, _LazyArray) else value

    def __iter__(self):
        return iter(self._real())

    def __getattr__(self, name):  # ndarray methods / attributes: reshape, copy, max, ...
        return getattr(self._real(), name)

    def __repr__(self):
        return "lazy(%r)" % (self._real(),)


class _Vector:
    """`Function.x`: owns the host array.  A device-resident solver installs hooks: `_pre_access` (download before
    a host read), `_post_access` (host-dirty mark once the host array was handed out) and `_assign_hook`
    (whole-field assignment from another field, done on the device when possible)."""

    def __init__(self, n):
        self._array = np.zeros(n, dtype=np.float64)
        self._pre_access = None  # callable() -> None, run before handing out the view
        self._post_access = None
        self._assign_hook = None  # callable(src_vector) -> bool (True: handled on the device)

    def _materialise(self):
        if self._pre_access is not None:
            self._pre_access()
        if self._post_access is not None:
            self._post_access()
        return self._array

    def _assign_on_device(self, src):
        return bool(self._assign_hook(src)) if self._assign_hook is not None else False

    @property
    def array(self):
        if self._pre_access is None and self._post_access is None and self._assign_hook is None:
            return self._array
        return _LazyArray(self)

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
