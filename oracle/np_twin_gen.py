"""NumPy twin for nodal equal-order elements beyond P1 simplices  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

SURVEY.md section 8f-4: `p_grade = 2` (P2/P2 triangles, /root/reference/src/solvers/stabilized_schur_backflow.py:84-87)
and quadrilateral cells (Q1/Q1, /root/reference/src/scenarios/unit_square_pipe.py:101-105).  PARITY UNPINNED like the
rest of oracle/ (no FEniCSx here, no golden vector in the reference).  What pins this file: on P1 triangles it reproduces
oracle/np_twin.py (closed-form element integrals) to round-off; Jacobian = d(residual) by central differences; patch
tests (tests/test_oracle_gen.py).

Same residual as np_twin.py (stabilized_schur.py:67-123, backflow variant :107,:158-176), written for any nodal element
with an affine geometry map and evaluated by quadrature:
  * P1 / P2 triangles: the 49-point degree-13 collapsed Gauss rule of the P1 path for all cell terms;
  * Q1 quadrilaterals (parallelograms: `create_rectangle(..., CellType.quadrilateral)`): 7 x 7 Gauss-Legendre.
For degree-2 elements the strong residual keeps its viscous part, div(2 mu eps(u_mid)) = mu (lap u + grad div u)
(second derivatives of the basis are cell constants under an affine map); the mixed derivative of Q1 likewise.
Facet terms: Gauss-Legendre on the facet, 2 points for degree-1 elements (UFL's estimate for the backflow term, as in
np_twin.py), 4 points for P2.

Local node order = DOLFINx / Basix: triangle vertices 0,1,2 then the edge nodes opposite to them; quadrilateral
(0,0),(1,0),(0,1),(1,1).  Local facets: triangle facet f opposite vertex f; quadrilateral 0:(0,1) 1:(0,2) 2:(1,3) 3:(2,3).
Local dof order of the element tensors: velocity (a,i) -> 2a+i, then pressure a -> 2 nloc + a.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
from scipy.special import roots_legendre

from .np_twin import EPS_VNORM, Params, quad_rule  # noqa: F401  (Params re-exported)

P1_TRI, P2_TRI, Q1_QUAD = 0, 1, 2


class Element:
    """Reference data of one element type: basis values / reference gradients at the cell quadrature points, constant
    reference Hessians, facet node lists and facet quadrature."""

    def __init__(self, etype):
        self.etype = etype
        if etype in (P1_TRI, P2_TRI):
            L, W = quad_rule()
            xi = L[:, 1:3]                       # (xi, eta) = (lambda_1, lambda_2)
            self.meas = 0.5                      # reference measure: integral = |det J| * meas * sum w f
            self.nvert = 3
            self.facets = [(1, 2), (0, 2), (0, 1)]
            self.ref_vertices = np.array([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0]])
        else:
            t, w = roots_legendre(7)
            t, w = 0.5 * (t + 1.0), 0.5 * w
            X, Y = np.meshgrid(t, t, indexing="ij")
            xi = np.stack([X.ravel(), Y.ravel()], axis=1)
            W = np.outer(w, w).ravel()
            self.meas = 1.0
            self.nvert = 4
            self.facets = [(0, 1), (0, 2), (1, 3), (2, 3)]
            self.ref_vertices = np.array([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0], [1.0, 1.0]])
        self.xi, self.w = xi, W
        self.nloc = {P1_TRI: 3, P2_TRI: 6, Q1_QUAD: 4}[etype]
        self.degree = 2 if etype == P2_TRI else 1
        self.phi, self.dphi = self.tabulate(xi)
        self.d2phi = self.hessians()
        nqf = 4 if self.degree == 2 else 2
        t, w = roots_legendre(nqf)
        self.ft, self.fw = 0.5 * (t + 1.0), 0.5 * w
        # per facet: reference points, basis values and reference gradients there
        self.fphi, self.fdphi = [], []
        for (a, b) in self.facets:
            pts = (1.0 - self.ft)[:, None] * self.ref_vertices[a] + self.ft[:, None] * self.ref_vertices[b]
            ph, dph = self.tabulate(pts)
            self.fphi.append(ph)
            self.fdphi.append(dph)
        self.ref_nodes = self.node_points()

    def node_points(self):
        v = self.ref_vertices
        if self.etype == P2_TRI:
            return np.vstack([v, 0.5 * (v[1] + v[2]), 0.5 * (v[0] + v[2]), 0.5 * (v[0] + v[1])])
        return v.copy()

    def tabulate(self, pts):
        x, y = pts[:, 0], pts[:, 1]
        n = len(pts)
        if self.etype == P1_TRI:
            phi = np.stack([1.0 - x - y, x, y], axis=1)
            d = np.zeros((n, 3, 2))
            d[:, 0] = (-1.0, -1.0); d[:, 1] = (1.0, 0.0); d[:, 2] = (0.0, 1.0)
            return phi, d
        if self.etype == P2_TRI:
            l = np.stack([1.0 - x - y, x, y], axis=1)
            dl = np.array([[-1.0, -1.0], [1.0, 0.0], [0.0, 1.0]])
            phi = np.empty((n, 6))
            d = np.empty((n, 6, 2))
            for i in range(3):
                phi[:, i] = l[:, i] * (2.0 * l[:, i] - 1.0)
                d[:, i] = (4.0 * l[:, i] - 1.0)[:, None] * dl[i]
            for e, (i, j) in enumerate([(1, 2), (0, 2), (0, 1)]):
                phi[:, 3 + e] = 4.0 * l[:, i] * l[:, j]
                d[:, 3 + e] = 4.0 * (l[:, i, None] * dl[j] + l[:, j, None] * dl[i])
            return phi, d
        phi = np.stack([(1 - x) * (1 - y), x * (1 - y), (1 - x) * y, x * y], axis=1)
        d = np.empty((n, 4, 2))
        d[:, 0, 0], d[:, 0, 1] = -(1 - y), -(1 - x)
        d[:, 1, 0], d[:, 1, 1] = (1 - y), -x
        d[:, 2, 0], d[:, 2, 1] = -y, (1 - x)
        d[:, 3, 0], d[:, 3, 1] = y, x
        return phi, d

    def hessians(self):
        H = np.zeros((self.nloc, 2, 2))
        if self.etype == P2_TRI:
            dl = np.array([[-1.0, -1.0], [1.0, 0.0], [0.0, 1.0]])
            for i in range(3):
                H[i] = 4.0 * np.outer(dl[i], dl[i])
            for e, (i, j) in enumerate([(1, 2), (0, 2), (0, 1)]):
                H[3 + e] = 4.0 * (np.outer(dl[i], dl[j]) + np.outer(dl[j], dl[i]))
        elif self.etype == Q1_QUAD:
            for a, s in enumerate([1.0, -1.0, -1.0, 1.0]):
                H[a, 0, 1] = H[a, 1, 0] = s
        return H


_ELEMENTS = {}


def element(etype):
    if etype not in _ELEMENTS:
        _ELEMENTS[etype] = Element(etype)
    return _ELEMENTS[etype]


def tau_pair(s, h, prm):
    """tau (SUPG/PSPG) and tau_L (LSIC) from |u_n|^2 = s and the cell size h (stabilized_schur.py:100-118)."""
    nu = prm.mu / prm.rho
    t1 = np.maximum(4.0 * s, EPS_VNORM**2) / (h * h)
    tau = 1.0 / np.sqrt(t1 + 4.0 / (prm.dt * prm.dt) + 16.0 * nu * nu / h**4)
    vn = np.sqrt(s)
    Re = vn * h / (2.0 * nu)
    z = np.where(Re <= 3.0, Re / 3.0, 1.0)
    return tau, vn * h * z / 2.0


def cell_geometry(el, x, cells):
    """Jinv [nc,2,2] with (grad phi)_i = sum_k dphi_ref_k Jinv[k,i]; |det J|; h = largest vertex distance."""
    p = x[cells[:, : el.nvert]]
    x0, x1, x2 = p[:, 0], p[:, 1], p[:, 2]
    J = np.stack([x1 - x0, x2 - x0], axis=2)  # J[:, i, k] = d x_i / d xi_k
    det = J[:, 0, 0] * J[:, 1, 1] - J[:, 0, 1] * J[:, 1, 0]
    Jinv = np.empty_like(J)
    Jinv[:, 0, 0], Jinv[:, 0, 1] = J[:, 1, 1] / det, -J[:, 0, 1] / det
    Jinv[:, 1, 0], Jinv[:, 1, 1] = -J[:, 1, 0] / det, J[:, 0, 0] / det
    if el.etype == Q1_QUAD:
        assert np.abs(p[:, 3] - (x1 + x2 - x0)).max() <= 1e-10 * np.abs(p).max(), "Q1 cells must be parallelograms (affine map)"
    h = np.zeros(len(cells))
    for a in range(el.nvert):
        for b in range(a + 1, el.nvert):
            h = np.maximum(h, np.linalg.norm(p[:, a] - p[:, b], axis=1))
    return Jinv, np.abs(det), h, det


def element_tensors(etype, x, cells, u, un, p, prm, facet_flags=None, want_jac=True, un2=None):
    """Fe [nc, 3 nloc], Je [nc, 3 nloc, 3 nloc].  facet_flags uint16 [nc]: bit f exterior facet f, bit 8+f backflow facet f."""
    el = element(etype)
    nl, nc = el.nloc, len(cells)
    rho, mu, dt, muf, th, a0 = prm.rho, prm.mu, prm.dt, prm.mu_facet, prm.theta, prm.a0
    Jinv, adet, h, det = cell_geometry(el, x, cells)
    ue, une, pe = u[cells], un[cells], p[cells]
    ubn = th * ue + (1.0 - th) * une
    wn = (a0 * ue + prm.a1 * une) / dt
    if prm.a2 != 0.0:
        wn = wn + prm.a2 * un2[cells] / dt
    wq = el.w * el.meas                                          # [q]
    grad = np.einsum("qak,cki->cqai", el.dphi, Jinv)             # [c,q,a,i]
    hess = np.einsum("akl,cki,clj->caij", el.d2phi, Jinv, Jinv)  # [c,a,i,j]
    lap = hess[:, :, 0, 0] + hess[:, :, 1, 1]
    phi = el.phi
    ub = np.einsum("qa,cai->cqi", phi, ubn)
    w = np.einsum("qa,cai->cqi", phi, wn)
    unq = np.einsum("qa,cai->cqi", phi, une)
    G = np.einsum("cqai,caj->cqij", grad, ubn)                   # G_ij = d_i ubar_j
    divu = G[..., 0, 0] + G[..., 1, 1]
    C = np.einsum("cqi,cqij->cqj", ub, G)
    gp = np.einsum("cqai,ca->cqi", grad, pe)
    pq = np.einsum("qa,ca->cq", phi, pe)
    visc = mu * (np.einsum("ca,cai->ci", lap, ubn) + np.einsum("caij,caj->ci", hess, ubn))  # div(2 mu eps(ubar)), cell constant
    R = rho * (w + C) - visc[:, None, :] + gp - rho * prm.f[None, None, :]
    tau, tauL = tau_pair(np.einsum("cqi,cqi->cq", unq, unq), h[:, None], prm)
    bgr = np.einsum("cqi,cqai->cqa", ub, grad)                   # ubar . grad phi_a
    S = G + np.swapaxes(G, 2, 3)
    dv = adet[:, None] * wq[None, :]                             # [c,q]

    Fu = np.einsum("cq,qa,cqi->cai", dv, phi, rho * (w + C - prm.f[None, None, :]))
    Fu += np.einsum("cq,cqaj,cqij->cai", dv, grad, mu * S)
    Fu -= np.einsum("cq,cq,cqai->cai", dv, pq, grad)
    Fu += np.einsum("cq,cq,cqi,cqa->cai", dv, tau, R, bgr)
    Fu += np.einsum("cq,cq,cqai->cai", dv, tauL * rho * divu, grad)
    Fp = np.einsum("cq,qa,cq->ca", dv, phi, divu) + np.einsum("cq,cq,cqi,cqai->ca", dv, tau / rho, R, grad)

    Je = None
    if want_jac:
        Je = np.zeros((nc, 3 * nl, 3 * nl))
        I2 = np.eye(2)
        # dR_i / d u_(b,j) = rho (a0/dt phi_b d_ij + th (phi_b G_ji + d_ij bgr_b)) - mu th (lap_b d_ij + hess_b[i,j])
        dR = rho * (a0 / dt * np.einsum("qb,ij->qbij", phi, I2)[None] + th * (np.einsum("qb,cqji->cqbij", phi, G) + np.einsum("cqb,ij->cqbij", bgr, I2)))
        dR = dR - mu * th * (np.einsum("cb,ij->cbij", lap, I2) + hess)[:, None]
        dWC = rho * (a0 / dt * np.einsum("qb,ij->qbij", phi, I2)[None] + th * (np.einsum("qb,cqji->cqbij", phi, G) + np.einsum("cqb,ij->cqbij", bgr, I2)))
        Juu = np.einsum("cq,qa,cqbij->caibj", dv, phi, dWC)
        Juu += mu * th * (np.einsum("cq,cqaj,cqbi->caibj", dv, grad, grad) + np.einsum("cq,cqak,cqbk,ij->caibj", dv, grad, grad, I2))
        Juu += np.einsum("cq,cq,cqbij,cqa->caibj", dv, tau, dR, bgr)
        Juu += th * np.einsum("cq,cq,cqi,qb,cqaj->caibj", dv, tau, R, phi, grad)
        Juu += rho * th * np.einsum("cq,cq,cqbj,cqai->caibj", dv, tauL, grad, grad)
        Jup = -np.einsum("cq,qb,cqai->caib", dv, phi, grad) + np.einsum("cq,cq,cqbi,cqa->caib", dv, tau, grad, bgr)
        Jpu = th * np.einsum("cq,qa,cqbj->cabj", dv, phi, grad) + np.einsum("cq,cq,cqbij,cqai->cabj", dv, tau / rho, dR, grad)
        Jpp = np.einsum("cq,cq,cqbi,cqai->cab", dv, tau / rho, grad, grad)
        Je[:, : 2 * nl, : 2 * nl] = Juu.reshape(nc, 2 * nl, 2 * nl)
        Je[:, : 2 * nl, 2 * nl:] = Jup.reshape(nc, 2 * nl, nl)
        Je[:, 2 * nl:, : 2 * nl] = Jpu.reshape(nc, nl, 2 * nl)
        Je[:, 2 * nl:, 2 * nl:] = Jpp

    if facet_flags is not None:
        cen = x[cells[:, : el.nvert]].mean(axis=1)
        for f, (va, vb) in enumerate(el.facets):
            ext = (facet_flags >> f) & 1 if prm.ds_terms else np.zeros(nc, dtype=np.int64)
            bf = (facet_flags >> (8 + f)) & 1 if prm.beta_backflow != 0.0 else np.zeros(nc, dtype=np.int64)
            sel = np.nonzero(ext | bf)[0]
            if len(sel) == 0:
                continue
            xa, xb = x[cells[sel, va]], x[cells[sel, vb]]
            t = xb - xa
            elen = np.linalg.norm(t, axis=1)
            n = np.stack([t[:, 1], -t[:, 0]], axis=1) / elen[:, None]
            n *= np.sign(np.einsum("ci,ci->c", 0.5 * (xa + xb) - cen[sel], n))[:, None]   # outward
            fphi = el.fphi[f]                                                # [qf, a]
            fgrad = np.einsum("qak,cki->cqai", el.fdphi[f], Jinv[sel])       # [c,qf,a,i]
            fw = elen[:, None] * el.fw[None, :]                              # [c,qf]
            ubf = np.einsum("qa,cai->cqi", fphi, ubn[sel])
            if prm.ds_terms:
                m = ext[sel].astype(np.float64)[:, None] * fw
                pf = np.einsum("qa,ca->cq", fphi, pe[sel])
                Gf = np.einsum("cqai,caj->cqij", fgrad, ubn[sel])
                # + p n.v - mu_f (v . (nabla_grad ubar) n) = phi_a (p n_i - mu_f sum_j d_i ubar_j n_j)
                Fu[sel] += np.einsum("cq,qa,cqi->cai", m, fphi, pf[:, :, None] * n[:, None, :] - muf * np.einsum("cqij,cj->cqi", Gf, n))
                if want_jac:
                    Je[sel, : 2 * nl, 2 * nl:] += np.einsum("cq,qa,qb,ci->caib", m, fphi, fphi, n).reshape(len(sel), 2 * nl, nl)
                    Je[sel, : 2 * nl, : 2 * nl] -= muf * th * np.einsum("cq,qa,cqbi,cj->caibj", m, fphi, fgrad, n).reshape(len(sel), 2 * nl, 2 * nl)
            if prm.beta_backflow != 0.0:
                m = bf[sel].astype(np.float64)[:, None] * fw
                sq = np.einsum("qa,cai,ci->cq", fphi, une[sel], n)
                cq = prm.beta_backflow * rho * 0.5 * (sq - np.abs(sq)) * m
                Fu[sel] -= np.einsum("cq,qa,cqi->cai", cq, fphi, ubf)
                if want_jac:
                    Je[sel, : 2 * nl, : 2 * nl] -= th * np.einsum("cq,qa,qb,ij->caibj", cq, fphi, fphi, np.eye(2)).reshape(len(sel), 2 * nl, 2 * nl)
    Fe = np.concatenate([Fu.reshape(nc, 2 * nl), Fp], axis=1)
    return Fe, Je


class Problem:
    """Mesh (node coordinates, cells [nc, nloc]) + parameters + Dirichlet data; same interface as np_twin.Problem."""

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
        self.ndof = 3 * self.nv
        self.nu = 2 * self.nv
        ld = np.empty((self.nc, 3 * nl), dtype=np.int64)
        for a in range(nl):
            ld[:, 2 * a] = 2 * self.cells[:, a]
            ld[:, 2 * a + 1] = 2 * self.cells[:, a] + 1
            ld[:, 2 * nl + a] = 2 * self.nv + self.cells[:, a]
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
        """Global node ids of exterior facet k (2 for degree 1, 3 for P2: the edge node last)."""
        c, f = self.facet_cells[k], self.facet_local[k]
        va, vb = self.el.facets[f]
        loc = [va, vb] + ([3 + f] if self.etype == P2_TRI else [])
        return self.cells[c, loc]

    def clear_bcs(self):
        self.isbc[:] = False
        self.bcval[:] = 0.0
        self.bcmult[:] = 0.0

    def add_bc_u(self, nodes, values):
        nodes = np.asarray(nodes, dtype=np.int64)
        values = np.asarray(values, dtype=np.float64).reshape(-1, 2)
        for i in range(2):
            d = 2 * nodes + i
            self.isbc[d] = True
            self.bcval[d] = values[:, i]
            self.bcmult[d] += 1.0

    def add_bc_p(self, nodes, values):
        nodes = np.asarray(nodes, dtype=np.int64)
        d = 2 * self.nv + nodes
        self.isbc[d] = True
        self.bcval[d] = np.asarray(values, dtype=np.float64).reshape(-1)
        self.bcmult[d] += 1.0

    def split(self, xvec):
        return xvec[: 2 * self.nv].reshape(-1, 2), xvec[2 * self.nv:]

    def assemble(self, xvec, un, want_jac=True, apply_bc=True, un2=None):
        """Dirichlet handling of stabilized_schur.py:157-175, as np_twin.Problem.assemble."""
        u, p = self.split(xvec)
        need_j = want_jac
        lift = None
        if apply_bc and self.isbc.any():
            lift = np.where(self.isbc, self.bcval - xvec, 0.0)
            if np.any(lift != 0.0):
                need_j = True
        Fe, Je = element_tensors(self.etype, self.x, self.cells, u, np.asarray(un).reshape(-1, 2), p, self.prm, self.facet_flags,
                                 want_jac=need_j, un2=None if un2 is None else np.asarray(un2).reshape(-1, 2))
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
        """sqrt(int u.u), sqrt(int p^2) with the element's own mass matrix (scenario.py:315-324)."""
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
        cen = self.x[self.cells[:, : el.nvert]].mean(axis=1)
        for k in np.asarray(facets, dtype=np.int64):
            c, f = self.facet_cells[k], self.facet_local[k]
            va, vb = el.facets[f]
            xa, xb = self.x[self.cells[c, va]], self.x[self.cells[c, vb]]
            t = xb - xa
            n = np.array([t[1], -t[0]])
            n *= np.sign((0.5 * (xa + xb) - cen[c]) @ n)
            uq = el.fphi[f] @ u[self.cells[c]]
            tot += float((el.fw[:, None] * uq).sum(axis=0) @ n)
        return tot


def p2_from_p1(x, cells):
    """P2 nodes of a straight-sided triangle mesh: vertices first, then one node per edge (midpoint).
    Returns (node coordinates [nn,2], cells [nc,6] in DOLFINx local order, edges [ne,2])."""
    cells = np.asarray(cells, dtype=np.int64)
    nv = len(x)
    loc = [(1, 2), (0, 2), (0, 1)]
    e = np.concatenate([np.sort(cells[:, l], axis=1) for l in loc])
    ue, inv = np.unique(e, axis=0, return_inverse=True)
    inv = inv.reshape(3, len(cells)).T
    xn = np.vstack([x, 0.5 * (x[ue[:, 0]] + x[ue[:, 1]])])
    return xn, np.hstack([cells, nv + inv]), ue
