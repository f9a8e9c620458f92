"""Deterministic geometry for the tree-shaped BASELINE domain (config 5): the Murray-law vascular tree and an
implicit-domain triangle mesher (gmsh / OpenCASCADE do not exist on the GPU box)."""
