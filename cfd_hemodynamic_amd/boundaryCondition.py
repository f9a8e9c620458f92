"""`BoundaryCondition` with the reference's interface
(/root/reference/src/boundaryCondition.py:13-55): wraps a source Function,
locates the constrained vertex blocks topologically or geometrically, and hands
out a DirichletBC whose `update()` re-interpolates the source (called by the
solver before every step, stabilized_schur.py:170)."""
from __future__ import annotations

from types import MethodType
from typing import Callable

from numpy import ndarray

from .fem import (DirichletBC, Function, FunctionSpace, dirichletbc, locate_dofs_geometrical,
                  locate_dofs_topological)


class BoundaryCondition:
    def __init__(self, f: Function):
        self._topological = False
        self._geometrical = False
        self.f = f

    def initTopological(self, entity_dim: int, entities: ndarray) -> None:
        assert not (self._topological or self._geometrical)
        self.entity_dim = entity_dim
        self.entities = entities
        self._topological = True

    def initGeometrical(self, marker: Callable) -> None:
        assert not (self._topological or self._geometrical)
        self.marker = marker
        self._geometrical = True

    def _getDofs(self, V: FunctionSpace) -> ndarray:
        assert self._topological or self._geometrical
        if self._topological:
            return locate_dofs_topological(V, self.entity_dim, self.entities)
        return locate_dofs_geometrical(V, self.marker)

    def getBC(self, V: FunctionSpace) -> DirichletBC:
        dofs = self._getDofs(V)
        self._f_V = Function(V)
        self._f_V.interpolate(self.f)
        bc = dirichletbc(self._f_V, dofs)

        def update(inner_self):
            # the reference re-interpolates the whole field (:48-51); the values are read at `dofs` only,
            # so the re-interpolation is restricted to them (identical Dirichlet data, O(boundary) work)
            self._f_V.interpolate_at(self.f, dofs)

        bc.update = MethodType(update, bc)
        return bc

    def updateBCValues(self, f: Function) -> None:
        assert self._f_V, "Boundary condition values have not been initialized."
