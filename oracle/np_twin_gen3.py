"""NumPy twin for nodal equal-order elements in 3-D beyond P1 tetrahedra  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

SURVEY.md section 8f-4, the 3-D half: hexahedral cells (Q1/Q1, /root/reference/src/scenarios/unit_cube_pipe.py:103-109,
`create_box(..., cell_type=CellType.hexahedron)`) and `p_grade = 2` on tetrahedra (P2/P2,
/root/reference/src/solvers/stabilized_schur_backflow.py:84-87 with the 3-D meshes of scenario_factory.py:47-49).
PARITY UNPINNED like the rest of oracle/.  What pins this file: on P1 tetrahedra it reproduces oracle/np_twin_nd.py (closed-form
element integrals) to round-off; Jacobian = d(residual) by central differences; patch tests (tests/test_oracle_gen3.py).

The residual is the one of np_twin_gen.py / np_twin_nd.py (stabilized_schur.py:67-123, backflow variant :107,:158-176), evaluated
by quadrature on an affine cell:
  * tetrahedra: the 171-point degree-13 rule of the P1 path (np_twin_nd.quad_rule(3), include/cfdh_quad_tet.h);
  * hexahedra (parallelepipeds: `create_box`): 7 x 7 x 7 Gauss-Legendre.
The strong residual keeps its viscous part div(2 mu eps(u_mid)) = mu (lap u + grad div u): constant second derivatives on P2
tetrahedra, the mixed derivatives of the trilinear functions (linear in the third coordinate) on hexahedra -- evaluated at the
quadrature points.  Facet terms: 2 x 2 Gauss on the quadrilateral facets of a hexahedron, the 49-point collapsed rule on the
triangular facets of a P2 tetrahedron, the 6-point degree-3 rule of np_twin_nd.facet_rule(3) on P1 tetrahedra.

Local node order = DOLFINx / Basix.  Hexahedron: vertex v = i + 2 j + 4 k at (i, j, k); facets 0:(0,1,2,3) 1:(0,1,4,5)
2:(0,2,4,6) 3:(1,3,5,7) 4:(2,3,6,7) 5:(4,5,6,7).  P2 tetrahedron: vertices 0..3, then the edge nodes in Basix edge order
(2,3) (1,3) (1,2) (0,3) (0,2) (0,1); facet f opposite vertex f.  Element dofs: velocity (a, i) -> 3 a + i, then pressure 3 nloc + a.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
from scipy.special import roots_legendre

from . import np_twin_nd as ND
from .np_twin_gen import tau_pair
from .np_twin_nd import Params  # noqa: F401  (re-exported)

P1_TET, P2_TET, Q1_HEX = 3, 4, 5
TET_EDGES = [(2, 3), (1, 3), (1, 2), (0, 3), (0, 2), (0, 1)]
HEX_FACETS = [(0, 1, 2, 3), (0, 1, 4, 5), (0, 2, 4, 6), (1, 3, 5, 7), (2, 3, 6, 7), (4, 5, 6, 7)]
TET_FACETS = [(1, 2, 3), (0, 2, 3), (0, 1, 3), (0, 1, 2)]


class Element:
    def __init__(self, etype):
        self.etype = etype
        if etype == Q1_HEX:
            t, w = roots_legendre(7)
            t, w = 0.5 * (t + 1.0), 0.5 * w
            X, Y, Z = np.meshgrid(t, t, t, indexing="ij")
            self.xi = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
            self.w = np.einsum("i,j,k->ijk", w, w, w).ravel()
            self.meas, self.nvert, self.nloc, self.degree = 1.0, 8, 8, 1
            self.ref_vertices = np.array([[i, j, k] for k in (0, 1) for j in (0, 1) for i in (0, 1)], dtype=float)
            self.facets = HEX_FACETS
            t2, w2 = roots_legendre(2)
            t2, w2 = 0.5 * (t2 + 1.0), 0.5 * w2
            S, T = np.meshgrid(t2, t2, indexing="ij")
            self.fst = np.stack([S.ravel(), T.ravel()], axis=1)   # facet parameters (s, t) in [0, 1]^2
            self.fw = np.outer(w2, w2).ravel()                    # weights summing to 1: times the facet area
        else:
            L, W = ND.quad_rule(3)
            self.xi, self.w = L[:, 1:4], W
            self.meas, self.nvert = 1.0 / 6.0, 4
            self.nloc, self.degree = (4, 1) if etype == P1_TET else (10, 2)
            self.ref_vertices = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], dtype=float)
            self.facets = TET_FACETS
            if etype == P1_TET:
                P, Wf = ND.facet_rule(3)
                self.fbary, self.fw = np.asarray(P, dtype=float), np.asarray(Wf, dtype=float)
            else:
                Lf, Wf = ND.quad_rule(2)
                self.fbary, self.fw = Lf, Wf                       # barycentric w.r.t. the facet's vertices, weights sum to 1
        self.phi, self.dphi, self.d2phi = self.tabulate(self.xi)
        # per facet: reference points, basis values and reference gradients there
        self.fpts, self.fphi, self.fdphi = [], [], []
        for vs in self.facets:
            rv = self.ref_vertices[list(vs)]
            if etype == Q1_HEX:
                s, t = self.fst[:, 0:1], self.fst[:, 1:2]
                pts = (1 - s) * (1 - t) * rv[0] + s * (1 - t) * rv[1] + (1 - s) * t * rv[2] + s * t * rv[3]
            else:
                pts = self.fbary @ rv
            ph, dph, _ = self.tabulate(pts)
            self.fpts.append(pts); self.fphi.append(ph); self.fdphi.append(dph)

    def node_points(self):
        v = self.ref_vertices
        if self.etype == P2_TET:
            return np.vstack([v] + [0.5 * (v[i] + v[j])[None] for i, j in TET_EDGES])
        return v.copy()

    def tabulate(self, pts):
        """phi [n, nloc], reference gradients [n, nloc, 3], reference Hessians [n, nloc, 3, 3]"""
        n = len(pts)
        x, y, z = pts[:, 0], pts[:, 1], pts[:, 2]
        nl = self.nloc
        phi, d, H = np.zeros((n, nl)), np.zeros((n, nl, 3)), np.zeros((n, nl, 3, 3))
        if self.etype == Q1_HEX:
            f = [np.stack([1 - c, c], axis=1) for c in (x, y, z)]      # f[dir][:, bit]
            df = [-1.0, 1.0]
            for v in range(8):
                i, j, k = v & 1, (v >> 1) & 1, (v >> 2) & 1
                phi[:, v] = f[0][:, i] * f[1][:, j] * f[2][:, k]
                d[:, v, 0] = df[i] * f[1][:, j] * f[2][:, k]
                d[:, v, 1] = f[0][:, i] * df[j] * f[2][:, k]
                d[:, v, 2] = f[0][:, i] * f[1][:, j] * df[k]
                H[:, v, 0, 1] = H[:, v, 1, 0] = df[i] * df[j] * f[2][:, k]
                H[:, v, 0, 2] = H[:, v, 2, 0] = df[i] * f[1][:, j] * df[k]
                H[:, v, 1, 2] = H[:, v, 2, 1] = f[0][:, i] * df[j] * df[k]
            return phi, d, H
        l = np.stack([1.0 - x - y - z, x, y, z], axis=1)
        dl = np.array([[-1.0, -1.0, -1.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]])
        if self.etype == P1_TET:
            phi[:] = l
            d[:] = dl[None]
            return phi, d, H
        for i in range(4):
            phi[:, i] = l[:, i] * (2.0 * l[:, i] - 1.0)
            d[:, i] = (4.0 * l[:, i] - 1.0)[:, None] * dl[i]
            H[:, i] = 4.0 * np.outer(dl[i], dl[i])
        for e, (i, j) in enumerate(TET_EDGES):
            phi[:, 4 + e] = 4.0 * l[:, i] * l[:, j]
            d[:, 4 + e] = 4.0 * (l[:, i, None] * dl[j] + l[:, j, None] * dl[i])
            H[:, 4 + e] = 4.0 * (np.outer(dl[i], dl[j]) + np.outer(dl[j], dl[i]))
        return phi, d, H


_ELEMENTS = {}


def element(etype):
    if etype not in _ELEMENTS:
        _ELEMENTS[etype] = Element(etype)
    return _ELEMENTS[etype]


def cell_geometry(el, x, cells):
    """Jinv [nc,3,3] with (grad phi)_i = sum_k dphi_ref_k Jinv[k,i]; |det J|; h = largest vertex distance; det."""
    p = x[cells[:, : el.nvert]]
    x0 = p[:, 0]
    if el.etype == Q1_HEX:
        J = np.stack([p[:, 1] - x0, p[:, 2] - x0, p[:, 4] - x0], axis=2)   # J[:, i, k] = d x_i / d xi_k
        for v in range(8):
            i, j, k = v & 1, (v >> 1) & 1, (v >> 2) & 1
            assert np.abs(p[:, v] - (x0 + i * (p[:, 1] - x0) + j * (p[:, 2] - x0) + k * (p[:, 4] - x0))).max() <= 1e-10 * max(np.abs(p).max(), 1e-300), \
                "hexahedral cells must be parallelepipeds (affine map)"
    else:
        J = np.stack([p[:, 1] - x0, p[:, 2] - x0, p[:, 3] - x0], axis=2)
    det = np.linalg.det(J)
    Jinv = np.linalg.inv(J)
    h = np.zeros(len(cells))
    for a in range(el.nvert):
        for b in range(a + 1, el.nvert):
            h = np.maximum(h, np.linalg.norm(p[:, a] - p[:, b], axis=1))
    return Jinv, np.abs(det), h, det


def facet_geometry(el, x, cells, sel, f):
    """Outward unit normals [n,3] and facet measures [n] of local facet f of the cells sel (planar facets of affine cells)."""
    vs = el.facets[f]
    P = x[cells[sel][:, list(vs)]]
    e1, e2 = P[:, 1] - P[:, 0], P[:, 2] - P[:, 0]
    cr = np.cross(e1, e2)
    nrm = np.linalg.norm(cr, axis=1)
    area = nrm if el.etype == Q1_HEX else 0.5 * nrm
    n = cr / nrm[:, None]
    cen = x[cells[sel][:, : el.nvert]].mean(axis=1)
    fc = P.mean(axis=1)
    n *= np.sign(np.einsum("ci,ci->c", fc - cen, n))[:, None]
    return n, area


def element_tensors(etype, x, cells, u, un, p, prm, facet_flags=None, want_jac=True, un2=None):
    """Fe [nc, 4 nloc], Je [nc, 4 nloc, 4 nloc].  facet_flags uint16 [nc]: bit f exterior facet f, bit 8+f backflow facet f."""
    el = element(etype)
    nl, nc = el.nloc, len(cells)
    rho, mu, dt, muf, th, a0 = prm.rho, prm.mu, prm.dt, prm.mu_facet, prm.theta, prm.a0
    fvec = np.asarray(prm.f, dtype=float)
    Jinv, adet, h, det = cell_geometry(el, x, cells)
    ue, une, pe = u[cells], un[cells], p[cells]
    ubn = th * ue + (1.0 - th) * une
    wn = (a0 * ue + prm.a1 * une) / dt
    if prm.a2 != 0.0:
        wn = wn + prm.a2 * un2[cells] / dt
    wq = el.w * el.meas
    grad = np.einsum("qak,cki->cqai", el.dphi, Jinv)
    hess = np.einsum("qakl,cki,clj->cqaij", el.d2phi, Jinv, Jinv)
    lap = np.einsum("cqaii->cqa", hess)
    phi = el.phi
    ub = np.einsum("qa,cai->cqi", phi, ubn)
    w = np.einsum("qa,cai->cqi", phi, wn)
    unq = np.einsum("qa,cai->cqi", phi, une)
    G = np.einsum("cqai,caj->cqij", grad, ubn)
    divu = np.einsum("cqii->cq", G)
    C = np.einsum("cqi,cqij->cqj", ub, G)
    gp = np.einsum("cqai,ca->cqi", grad, pe)
    pq = np.einsum("qa,ca->cq", phi, pe)
    visc = mu * (np.einsum("cqa,cai->cqi", lap, ubn) + np.einsum("cqaij,caj->cqi", hess, ubn))
    R = rho * (w + C) - visc + gp - rho * fvec[None, None, :]
    tau, tauL = tau_pair(np.einsum("cqi,cqi->cq", unq, unq), h[:, None], prm)
    bgr = np.einsum("cqi,cqai->cqa", ub, grad)
    S = G + np.swapaxes(G, 2, 3)
    dv = adet[:, None] * wq[None, :]
    I3 = np.eye(3)

    Fu = np.einsum("cq,qa,cqi->cai", dv, phi, rho * (w + C - fvec[None, None, :]))
    Fu += np.einsum("cq,cqaj,cqij->cai", dv, grad, mu * S)
    Fu -= np.einsum("cq,cq,cqai->cai", dv, pq, grad)
    Fu += np.einsum("cq,cq,cqi,cqa->cai", dv, tau, R, bgr)
    Fu += np.einsum("cq,cq,cqai->cai", dv, tauL * rho * divu, grad)
    Fp = np.einsum("cq,qa,cq->ca", dv, phi, divu) + np.einsum("cq,cq,cqi,cqai->ca", dv, tau / rho, R, grad)

    Je = None
    if want_jac:
        Je = np.zeros((nc, 4 * nl, 4 * nl))
        dWC = rho * (a0 / dt * np.einsum("qb,ij->qbij", phi, I3)[None] + th * (np.einsum("qb,cqji->cqbij", phi, G) + np.einsum("cqb,ij->cqbij", bgr, I3)))
        dR = dWC - mu * th * (np.einsum("cqb,ij->cqbij", lap, I3) + hess)
        Juu = np.einsum("cq,qa,cqbij->caibj", dv, phi, dWC)
        Juu += mu * th * (np.einsum("cq,cqaj,cqbi->caibj", dv, grad, grad) + np.einsum("cq,cqak,cqbk,ij->caibj", dv, grad, grad, I3))
        Juu += np.einsum("cq,cq,cqbij,cqa->caibj", dv, tau, dR, bgr)
        Juu += th * np.einsum("cq,cq,cqi,qb,cqaj->caibj", dv, tau, R, phi, grad)
        Juu += rho * th * np.einsum("cq,cq,cqbj,cqai->caibj", dv, tauL, grad, grad)
        Jup = -np.einsum("cq,qb,cqai->caib", dv, phi, grad) + np.einsum("cq,cq,cqbi,cqa->caib", dv, tau, grad, bgr)
        Jpu = th * np.einsum("cq,qa,cqbj->cabj", dv, phi, grad) + np.einsum("cq,cq,cqbij,cqai->cabj", dv, tau / rho, dR, grad)
        Jpp = np.einsum("cq,cq,cqbi,cqai->cab", dv, tau / rho, grad, grad)
        Je[:, : 3 * nl, : 3 * nl] = Juu.reshape(nc, 3 * nl, 3 * nl)
        Je[:, : 3 * nl, 3 * nl:] = Jup.reshape(nc, 3 * nl, nl)
        Je[:, 3 * nl:, : 3 * nl] = Jpu.reshape(nc, nl, 3 * nl)
        Je[:, 3 * nl:, 3 * nl:] = Jpp

    if facet_flags is not None:
        for f in range(len(el.facets)):
            ext = (facet_flags >> f) & 1 if prm.ds_terms else np.zeros(nc, dtype=np.int64)
            bf = (facet_flags >> (8 + f)) & 1 if prm.beta_backflow != 0.0 else np.zeros(nc, dtype=np.int64)
            sel = np.nonzero(ext | bf)[0]
            if len(sel) == 0:
                continue
            n, area = facet_geometry(el, x, cells, sel, f)
            fphi = el.fphi[f]
            fgrad = np.einsum("qak,cki->cqai", el.fdphi[f], Jinv[sel])
            fw = area[:, None] * el.fw[None, :]
            ubf = np.einsum("qa,cai->cqi", fphi, ubn[sel])
            if prm.ds_terms:
                m = ext[sel].astype(np.float64)[:, None] * fw
                pf = np.einsum("qa,ca->cq", fphi, pe[sel])
                Gf = np.einsum("cqai,caj->cqij", fgrad, ubn[sel])
                Fu[sel] += np.einsum("cq,qa,cqi->cai", m, fphi, pf[:, :, None] * n[:, None, :] - muf * np.einsum("cqij,cj->cqi", Gf, n))
                if want_jac:
                    Je[sel, : 3 * nl, 3 * nl:] += np.einsum("cq,qa,qb,ci->caib", m, fphi, fphi, n).reshape(len(sel), 3 * nl, nl)
                    Je[sel, : 3 * nl, : 3 * nl] -= muf * th * np.einsum("cq,qa,cqbi,cj->caibj", m, fphi, fgrad, n).reshape(len(sel), 3 * nl, 3 * nl)
            if prm.beta_backflow != 0.0:
                m = bf[sel].astype(np.float64)[:, None] * fw
                sq = np.einsum("qa,cai,ci->cq", fphi, une[sel], n)
                cq = prm.beta_backflow * rho * 0.5 * (sq - np.abs(sq)) * m
                Fu[sel] -= np.einsum("cq,qa,cqi->cai", cq, fphi, ubf)
                if want_jac:
                    Je[sel, : 3 * nl, : 3 * nl] -= th * np.einsum("cq,qa,qb,ij->caibj", cq, fphi, fphi, I3).reshape(len(sel), 3 * nl, 3 * nl)
    Fe = np.concatenate([Fu.reshape(nc, 3 * nl), Fp], axis=1)
    return Fe, Je


class Problem:
    """Mesh (node coordinates [nn,3], cells [nc, nloc]) + parameters + Dirichlet data; same interface as np_twin_gen.Problem."""

    def __init__(self, etype, x, cells, facet_cells, facet_local, prm):
        self.etype = etype
        self.el = element(etype)
        self.x = np.ascontiguousarray(x, dtype=np.float64)
        self.cells = np.ascontiguousarray(cells, dtype=np.int64)
        self.nv, self.nc, nl = len(self.x), len(self.cells), self.el.nloc
        self.prm = prm
        self.facet_cells = np.asarray(facet_cells, dtype=np.int64)
        self.facet_local = np.asarray(facet_local, dtype=np.int64)
        ff = np.zeros(self.nc, dtype=np.uint16)
        np.bitwise_or.at(ff, self.facet_cells, (1 << self.facet_local).astype(np.uint16))
        self.facet_flags = ff
        self._ext_flags = ff.copy()
        self.ndof = 4 * self.nv
        self.nu = 3 * self.nv
        ld = np.empty((self.nc, 4 * nl), dtype=np.int64)
        for a in range(nl):
            for i in range(3):
                ld[:, 3 * a + i] = 3 * self.cells[:, a] + i
            ld[:, 3 * nl + a] = 3 * self.nv + self.cells[:, a]
        self.ldofs = ld
        self.isbc = np.zeros(self.ndof, dtype=bool)
        self.bcval = np.zeros(self.ndof)
        self.bcmult = np.zeros(self.ndof)

    def set_boundary_terms(self, ds_terms, backflow_facets=None, beta=0.0):
        self.prm.ds_terms = bool(ds_terms)
        self.prm.beta_backflow = float(beta)
        ff = self._ext_flags.copy()
        if backflow_facets is not None and len(backflow_facets):
            k = np.asarray(backflow_facets, dtype=np.int64)
            np.bitwise_or.at(ff, self.facet_cells[k], (256 << self.facet_local[k]).astype(np.uint16))
        self.facet_flags = ff

    def facet_nodes(self, k):
        """Global node ids of exterior facet k (4 for a hexahedron, 3 or 6 for a tetrahedron: edge nodes last)."""
        c, f = self.facet_cells[k], self.facet_local[k]
        vs = list(self.el.facets[f])
        if self.etype == P2_TET:
            vs = vs + [4 + e for e, (i, j) in enumerate(TET_EDGES) if i in vs and j in vs]
        return self.cells[c, vs]

    def clear_bcs(self):
        self.isbc[:] = False
        self.bcval[:] = 0.0
        self.bcmult[:] = 0.0

    def add_bc_u(self, nodes, values):
        nodes = np.asarray(nodes, dtype=np.int64)
        values = np.asarray(values, dtype=np.float64).reshape(len(nodes), 3)
        for i in range(3):
            d = 3 * nodes + i
            self.isbc[d] = True
            self.bcval[d] = values[:, i]
            np.add.at(self.bcmult, d, 1.0)

    def add_bc_p(self, nodes, values):
        nodes = np.asarray(nodes, dtype=np.int64)
        d = self.nu + nodes
        self.isbc[d] = True
        self.bcval[d] = np.broadcast_to(np.asarray(values, dtype=np.float64), (len(nodes),))
        np.add.at(self.bcmult, d, 1.0)

    def split(self, xvec):
        return xvec[: self.nu].reshape(self.nv, 3), xvec[self.nu:]

    def assemble(self, xvec, un, want_jac=True, apply_bc=True, un2=None):
        """Dirichlet handling of stabilized_schur.py:157-175, as np_twin.Problem.assemble."""
        u, p = self.split(xvec)
        need_j = want_jac
        lift = None
        if apply_bc and self.isbc.any():
            lift = np.where(self.isbc, self.bcval - xvec, 0.0)
            if np.any(lift != 0.0):
                need_j = True
        Fe, Je = element_tensors(self.etype, self.x, self.cells, u, np.asarray(un).reshape(-1, 3), p, self.prm, self.facet_flags,
                                 want_jac=need_j, un2=None if un2 is None else np.asarray(un2).reshape(-1, 3))
        ld = self.ldofs
        nd = ld.shape[1]
        if apply_bc and self.isbc.any():
            bce = self.isbc[ld]
            if lift is not None and np.any(lift != 0.0):
                Fe = Fe + np.einsum("crk,ck->cr", Je, lift[ld])
            Fe = np.where(bce, 0.0, Fe)
            if Je is not None:
                Je = Je * (~bce)[:, :, None] * (~bce)[:, None, :]
        F = np.zeros(self.ndof)
        np.add.at(F, ld.ravel(), Fe.ravel())
        J = None
        if want_jac:
            rows = np.repeat(ld, nd, axis=1).ravel()
            cols = np.tile(ld, (1, nd)).ravel()
            J = sp.coo_matrix((Je.ravel(), (rows, cols)), shape=(self.ndof, self.ndof)).tocsr()
            J.sum_duplicates()
        if apply_bc and self.isbc.any():
            F[self.isbc] = (xvec - self.bcval)[self.isbc]
            if J is not None:
                J = (J + sp.diags(np.where(self.isbc, self.bcmult, 0.0))).tocsr()
        return F, J

    def newton(self, x0, un, rtol=1e-12, atol=1e-14, max_it=25, un2=None):
        x = x0.copy()
        hist = []
        singular = None
        for it in range(max_it + 1):
            F, J = self.assemble(x, un, want_jac=True, un2=un2)
            if singular is None:
                e = np.zeros(self.ndof)
                e[self.nu:] = 1.0 / np.sqrt(self.nv)
                jn = np.linalg.norm(J @ e)
                singular = bool(jn < 1e-7 and jn <= 1e-6 * np.linalg.norm(abs(J) @ e))
            fn = np.linalg.norm(F)
            hist.append(fn)
            if fn <= atol or (it > 0 and fn <= rtol * hist[0]):
                break
            if it == max_it:
                raise RuntimeError("twin newton did not converge: %r" % hist)
            if singular:
                e = np.zeros(self.ndof)
                e[self.nu:] = 1.0
                A = sp.bmat([[J, sp.csr_matrix(e[:, None])], [sp.csr_matrix(e[None, :]), None]]).tocsc()
                d = spla.splu(A).solve(np.concatenate([F, [0.0]]))[:-1]
            else:
                d = spla.splu(J.tocsc()).solve(F)
            x -= d
        return x, hist

    def l2_norms(self, xvec):
        u, p = self.split(xvec)
        el = self.el
        _, adet, _, _ = cell_geometry(el, self.x, self.cells)
        m = np.einsum("q,qa,qb->ab", el.w * el.meas, el.phi, el.phi)
        ue, pe = u[self.cells], p[self.cells]
        return (np.sqrt(np.einsum("c,ab,cai,cbi->", adet, m, ue, ue)), np.sqrt(np.einsum("c,ab,ca,cb->", adet, m, pe, pe)))

    def flux(self, xvec, facets):
        """int u.n over the given exterior facets (outward normal)."""
        u, _ = self.split(xvec)
        el = self.el
        tot = 0.0
        facets = np.asarray(facets, dtype=np.int64)
        for f in range(len(el.facets)):
            k = facets[self.facet_local[facets] == f]
            if len(k) == 0:
                continue
            sel = self.facet_cells[k]
            n, area = facet_geometry(el, self.x, self.cells, sel, f)
            uq = np.einsum("qa,cai->cqi", el.fphi[f], u[self.cells[sel]])
            tot += float(np.einsum("c,q,cqi,ci->", area, el.fw, uq, n))
        return tot


def p2_from_p1_tets(x, cells):
    """P2 nodes of a straight-sided tetrahedral mesh: vertices first, then one node per edge (midpoint).
    Returns (node coordinates [nn,3], cells [nc,10] in DOLFINx local order, edges [ne,2])."""
    cells = np.asarray(cells, dtype=np.int64)
    nv = len(x)
    e = np.concatenate([np.sort(cells[:, list(l)], axis=1) for l in TET_EDGES])
    ue, inv = np.unique(e, axis=0, return_inverse=True)
    inv = inv.reshape(6, len(cells)).T
    xn = np.vstack([x, 0.5 * (x[ue[:, 0]] + x[ue[:, 1]])])
    return xn, np.hstack([cells, nv + inv]), ue
