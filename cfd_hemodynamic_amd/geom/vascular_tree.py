"""Planar binary vascular tree obeying Murray's law -- the generator the reference's 2-D tree scenarios use
(/root/reference/src/geom/tree/tree_2d.py:32-198), restated as an explicit-stack walk over flat arrays.

Rules (tree_2d.py:118-176):
  * radii of the two children of a branch of radius r_p:  with q = (a / (1 - a))^(1/3) (a = `asymmetry`, the flow
    share of the left child; Poiseuille with L ~ r gives Q ~ r^3) and Murray's exponent g,
        r_left = r_p (1 + q^-g)^(-1/g),   r_right = r_left / q;
  * a branch of radius r is `length_ratio * r` long;
  * the children leave at  parent angle + half_angle * r_right / r_p  (left) and
    parent angle - half_angle * r_left / r_p  (right): the thinner child deflects more;
  * after `n_generations` bifurcations the branch ends are the terminals.
Node and edge numbering is the reference's depth-first pre-order (left subtree before right), so results can be
compared index by index (tests/test_geom.py against fixtures generated from the reference).
"""
from __future__ import annotations

import math

import numpy as np


class VascularTree:
    """nodes [n,2]; edges: int [m,2] (from, to) with `radius[m]`, `parent_radius[m]`; terminals: node ids."""

    def __init__(self, r_root=1.2, n_generations=3, gamma=3.0, bifurcation_angle=35.0, length_ratio=8.0,
                 asymmetry=0.5):
        if not 0.0 < asymmetry < 1.0:
            raise ValueError("asymmetry must lie strictly between 0 and 1")
        self.r_root, self.n_generations, self.gamma = float(r_root), int(n_generations), float(gamma)
        self.bifurcation_angle, self.length_ratio, self.asymmetry = float(bifurcation_angle), float(length_ratio), float(asymmetry)
        self.nodes = np.zeros((0, 2))
        self.edges = np.zeros((0, 2), dtype=np.int64)
        self.radius = np.zeros(0)
        self.parent_radius = np.zeros(0)
        self.terminals = []

    def child_radii(self, r_parent):
        q = (self.asymmetry / (1.0 - self.asymmetry)) ** (1.0 / 3.0)
        r_left = r_parent * (1.0 + q ** (-self.gamma)) ** (-1.0 / self.gamma)
        return r_left, r_left / q

    def generate(self, origin, direction=0.0):
        """Root branch from `origin` heading `direction` degrees (0 = +x), then the bifurcations."""
        pts = [np.array([float(origin[0]), float(origin[1])])]
        edges, rad, rpar, terminals = [], [], [], []

        def grow(start, radius, angle_deg, r_parent):
            th = math.radians(angle_deg)
            length = self.length_ratio * radius
            pts.append(pts[start] + length * np.array([math.cos(th), math.sin(th)]))
            edges.append((start, len(pts) - 1))
            rad.append(radius)
            rpar.append(r_parent)
            return len(pts) - 1

        tip = grow(0, self.r_root, direction, self.r_root)
        # pending bifurcations, last in first out; the right child is pushed first so that the left subtree is
        # numbered completely before the right sibling is even created (depth-first pre-order)
        todo = [("split", tip, self.r_root, float(direction), 1)]
        while todo:
            item = todo.pop()
            if item[0] == "branch":
                _, parent, r_child, angle, r_par, gen = item
                end = grow(parent, r_child, angle, r_par)
                todo.append(("split", end, r_child, angle, gen + 1))
                continue
            _, node, r_par, angle, gen = item
            if gen > self.n_generations:
                terminals.append(node)
                continue
            r_l, r_r = self.child_radii(r_par)
            a_l = angle + self.bifurcation_angle * (r_r / r_par)
            a_r = angle - self.bifurcation_angle * (r_l / r_par)
            todo.append(("branch", node, r_r, a_r, r_par, gen))
            todo.append(("branch", node, r_l, a_l, r_par, gen))
        self.nodes = np.array(pts)
        self.edges = np.array(edges, dtype=np.int64)
        self.radius = np.array(rad)
        self.parent_radius = np.array(rpar)
        self.terminals = terminals
        return self

    @property
    def bifurcations(self):
        """(node id, smallest child radius) of every node with two outgoing branches (tree_2d.py:178-198)."""
        out = []
        for n in np.unique(self.edges[:, 0]):
            sel = self.edges[:, 0] == n
            if sel.sum() >= 2:
                out.append((int(n), float(self.radius[sel].min())))
        return out

    def incoming_direction(self):
        """Unit direction of the branch ARRIVING at each node ((1,0)-rotated `direction` for the root start):
        the start tangent of the Bezier centreline of the branches leaving it (stenosis_with_tree.py:358-377)."""
        d = np.zeros_like(self.nodes)
        d[0] = self.nodes[self.edges[0, 1]] - self.nodes[0]
        for a, b in self.edges:
            d[b] = self.nodes[b] - self.nodes[a]
        n = np.linalg.norm(d, axis=1)
        return d / np.where(n > 0, n, 1.0)[:, None]
