"""Dirichlet data of a scenario, with the interface of the reference's `BoundaryCondition`
(/root/reference/src/boundaryCondition.py:13-52): a source Function `f`, a way of locating the constrained
vertex blocks -- `initTopological(entity_dim, entities)` or `initGeometrical(marker)`, exactly one of them --
and `getBC(V)`, which hands out a DirichletBC whose `update()` refreshes the values from the source (the solver
calls it before every residual evaluation, stabilized_schur.py:170; this is the hook a time-dependent inlet uses).

Built around one small strategy object per way of locating dofs; the values live in a Function on `V` owned by the
handed-out condition.  The refresh touches the constrained blocks only: Dirichlet values are never read
elsewhere, and re-interpolating a 10^6-dof field per condition and step is measurable on the host.
"""
from __future__ import annotations

from typing import Callable

import numpy as np

from .fem import DirichletBC, Function, FunctionSpace, locate_dofs_geometrical, locate_dofs_topological


class _ByEntities:
    """Vertex blocks of the given mesh entities (facets)."""

    def __init__(self, dim: int, entities: np.ndarray):
        self.dim, self.entities = int(dim), entities

    def locate(self, V: FunctionSpace) -> np.ndarray:
        return locate_dofs_topological(V, self.dim, self.entities)


class _ByCoordinates:
    """Vertex blocks whose coordinates satisfy `marker(x[3, n]) -> bool[n]`."""

    def __init__(self, marker: Callable[[np.ndarray], np.ndarray]):
        self.marker = marker

    def locate(self, V: FunctionSpace) -> np.ndarray:
        return locate_dofs_geometrical(V, self.marker)


class _SourcedBC(DirichletBC):
    """DirichletBC that remembers where its values come from."""

    def __init__(self, values: Function, dofs: np.ndarray, source):
        super().__init__(values, dofs)
        self._source = source

    def update(self) -> None:
        self.g.interpolate_at(self._source, self.dofs)


class BoundaryCondition:
    def __init__(self, f: Function):
        self.f = f            # source: a Function on the same space, or a callable x[3,n] -> values
        self._where = None    # _ByEntities | _ByCoordinates

    def _set_locator(self, loc) -> None:
        if self._where is not None:
            raise AssertionError("BoundaryCondition: the location of the constrained dofs was already given")
        self._where = loc

    def initTopological(self, entity_dim: int, entities: np.ndarray) -> None:
        self._set_locator(_ByEntities(entity_dim, entities))

    def initGeometrical(self, marker: Callable[[np.ndarray], np.ndarray]) -> None:
        self._set_locator(_ByCoordinates(marker))

    def getBC(self, V: FunctionSpace) -> DirichletBC:
        if self._where is None:
            raise AssertionError("BoundaryCondition: call initTopological or initGeometrical before getBC")
        values = Function(V)
        values.interpolate(self.f)
        return _SourcedBC(values, self._where.locate(V), self.f)
