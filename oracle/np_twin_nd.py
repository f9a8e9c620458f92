"""Dimension-generic NumPy/SciPy twin (affine P1/P1 simplices, d = 2 or 3)  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import anything under oracle/.
PARITY UNPINNED, as for oracle/np_twin.py (same header applies).  For d = 2 this file must reproduce np_twin.py
to round-off (tests/test_oracle_nd.py) -- that is what pins the d = 3 instance, which restates the same residual
(/root/reference/src/solvers/stabilized_schur.py:67-123), exact Jacobian (:185-189) and block assembly / Dirichlet
semantics (:144-175) on tetrahedra: 12 + 4 element dofs, the case of /root/reference/src/scenarios/simple_bifurcation.py.

Algebra (SURVEY.md Appendix A with general d): gradients of the barycentrics are cell constants;
  int_K l_a l_b = |K| (1 + d_ab) / ((d+1)(d+2)),   int_K l_a = |K| / (d+1);
  exterior facet f (opposite local vertex f): |f| = d |K| |grad l_f|, outward normal n = -grad l_f / |grad l_f|,
  oint_f l_a l_b = |f| (1 + d_ab) / (d (d+1)),   oint_f l_a = |f| / d   (a, b on the facet);
only M_ab = int_K tau l_a l_b and L = int_K tau_L need quadrature (collapsed Gauss, 7 points per direction: degree 13).
Monolithic ordering: all velocity dofs (vertex-major, component-minor), then all pressure dofs.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
from scipy.special import roots_jacobi, roots_legendre

EPS_VNORM = 1e-15


def quad_rule(d, n=7):
    """Barycentric points [nq, d+1] and weights (sum 1) on the reference simplex; d = 2 is the rule of np_twin.quad_rule."""
    if d == 2:
        tj, wj = roots_jacobi(n, 1.0, 0.0)
        tl, wl = roots_legendre(n)
        u, v = 0.5 * (tj + 1.0), 0.5 * (tl + 1.0)
        U, V = np.meshgrid(u, v, indexing="ij")
        W = np.outer(0.25 * wj, 0.5 * wl) * 2.0
        x, y = U.ravel(), (V * (1.0 - U)).ravel()
        return np.stack([1.0 - x - y, x, y], axis=1), W.ravel()
    # d = 3 (round 4): the fully symmetric 171-point degree-13 rule of include/cfdh_quad_tet.h, the table the C oracle and the kernels
    # compile in (generators and their provenance: tools/gen_quadrature_tet.py).  n != 7 asks for the collapsed Gauss rule below,
    # which tests/test_oracle_nd.py uses as the independent reference for the table's exactness.
    if n == 7:
        return tet_rule_table()
    return collapsed_tet_rule(n)


def collapsed_tet_rule(n):
    """x = u, y = v (1-u), z = w (1-u)(1-v); Jacobian (1-u)^2 (1-v): n^3 points, exact to degree 2n - 1 (rounds 2-3 used n = 7)."""
    t2, w2 = roots_jacobi(n, 2.0, 0.0)
    t1, w1 = roots_jacobi(n, 1.0, 0.0)
    t0, w0 = roots_legendre(n)
    u, v, w = 0.5 * (t2 + 1.0), 0.5 * (t1 + 1.0), 0.5 * (t0 + 1.0)
    U, V, Wc = np.meshgrid(u, v, w, indexing="ij")
    wt = np.einsum("i,j,k->ijk", w2 / 8.0, w1 / 4.0, w0 / 2.0) * 6.0  # reference volume 1/6 -> weights sum to 1
    x = U.ravel()
    y = (V * (1.0 - U)).ravel()
    z = (Wc * (1.0 - U) * (1.0 - V)).ravel()
    return np.stack([1.0 - x - y - z, x, y, z], axis=1), wt.ravel()


def tet_rule_table():
    """Barycentric points and weights of include/cfdh_quad_tet.h (CFDH3_QL, CFDH3_QW)."""
    import os
    import re
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "cfdh_quad_tet.h")
    text = open(path).read()
    nq = int(re.search(r"#define CFDH3_NQ (\d+)", text).group(1))
    num = r"[-+]?\d\.\d+e[-+]\d+"
    wtxt = text[text.index("CFDH3_QW[CFDH3_NQ]"):text.index("CFDH3_QL[CFDH3_NQ]")]
    ltxt = text[text.index("CFDH3_QL[CFDH3_NQ]"):]
    W = np.array([float(v) for v in re.findall(num, wtxt)])
    L = np.array([float(v) for v in re.findall(num, ltxt)]).reshape(-1, 4)
    assert len(W) == nq and L.shape == (nq, 4)
    return L, W


_RULES = {}


def facet_rule(d):
    """Degree-3 rule on the reference facet: barycentric points (w.r.t. the facet's vertices in increasing local index) and
    weights summing to 1.  Edge: 2-point Gauss-Legendre.  Triangle: the 6-point rule of Strang and Fix (the rule Basix would
    hand FFCx for degree 3 is a 6-point Xiao-Gimbutas rule; the integrand is not polynomial, so the two differ at the
    level the cell rules do)."""
    if d == 2:
        gq = 0.5 / np.sqrt(3.0)
        return [(0.5 + gq, 0.5 - gq), (0.5 - gq, 0.5 + gq)], [0.5, 0.5]
    a, b, c = 0.659027622374092, 0.231933368553031, 0.109039009072877
    pts = [(a, b, c), (a, c, b), (b, a, c), (b, c, a), (c, a, b), (c, b, a)]
    return pts, [1.0 / 6.0] * 6


def _rule(d):
    if d not in _RULES:
        _RULES[d] = quad_rule(d)
    return _RULES[d]


class Params:
    def __init__(self, dt, rho, mu, f, mu_facet=None, theta=0.5, a0=1.0, a1=-1.0, a2=0.0, ds_terms=True, beta_backflow=0.0):
        self.dt, self.rho, self.mu = float(dt), float(rho), float(mu)
        self.f = np.asarray(f, dtype=np.float64)
        self.mu_facet = float(mu if mu_facet is None else mu_facet)
        self.theta, self.a0, self.a1, self.a2 = float(theta), float(a0), float(a1), float(a2)
        self.ds_terms = bool(ds_terms)
        self.beta_backflow = float(beta_backflow)


def geometry(x, cells):
    """grad(lambda_a) [nc,d+1,d], |K| [nc], h [nc] (greatest vertex distance, stabilized_schur.py:85-88)."""
    d = x.shape[1]
    p = x[cells]                                   # [nc, d+1, d]
    Jm = np.transpose(p[:, 1:] - p[:, :1], (0, 2, 1))  # columns x_a - x_0
    det = np.linalg.det(Jm)
    Jinv = np.linalg.inv(Jm)                       # rows = grad l_1 .. grad l_d
    g = np.empty((len(cells), d + 1, d))
    g[:, 1:] = Jinv
    g[:, 0] = -Jinv.sum(axis=1)
    vol = np.abs(det) / (2.0 if d == 2 else 6.0)
    h = np.zeros(len(cells))
    for a in range(d + 1):
        for b in range(a + 1, d + 1):
            h = np.maximum(h, np.linalg.norm(p[:, a] - p[:, b], axis=1))
    return g, vol, h


def tau_moments(un, vol, h, prm):
    d = un.shape[2]
    QL, QW = _rule(d)
    nu = prm.mu / prm.rho
    uq = np.einsum("qa,caj->cqj", QL, un)
    s = np.einsum("cqj,cqj->cq", uq, uq)
    hh = h[:, None]
    t1 = np.maximum(4.0 * s, EPS_VNORM**2) / (hh * hh)
    t2 = 4.0 / (prm.dt * prm.dt)
    t3 = 16.0 * nu * nu / (hh**4)
    tau = 1.0 / np.sqrt(t1 + t2 + t3)
    vn = np.sqrt(s)
    Re = vn * hh / (2.0 * nu)
    z = np.where(Re <= 3.0, Re / 3.0, 1.0)
    tauL = vn * hh * z / 2.0
    M = vol[:, None, None] * np.einsum("q,cq,qa,qb->cab", QW, tau, QL, QL)
    Lm = vol * np.einsum("q,cq->c", QW, tauL)
    return M, Lm


def element_tensors(x, cells, u, un, p, prm, facet_flags=None, want_jac=True, un2=None):
    """Fe [nc, (d+1)^2 ... ] -> [nc, nd], Je [nc, nd, nd], nd = (d+1)(d+1); local order: velocity (a,i) -> d a + i, then
    pressure a -> d (d+1) + a.  facet_flags bit f: the facet opposite local vertex f is exterior."""
    d = x.shape[1]
    n1 = d + 1
    nd = n1 * n1
    rho, mu, dt, muf = prm.rho, prm.mu, prm.dt, prm.mu_facet
    g, vol, h = geometry(x, cells)
    nc = len(cells)
    ue, une, pe = u[cells], un[cells], p[cells]
    th, a0 = prm.theta, prm.a0
    ub = th * ue + (1.0 - th) * une
    w = (a0 * ue + prm.a1 * une) / dt
    if prm.a2 != 0.0:
        w = w + prm.a2 * un2[cells] / dt
    G = np.einsum("cai,caj->cij", g, ub)
    divu = np.einsum("cii->c", G)
    Cn = np.einsum("cai,cij->caj", ub, G)
    gp = np.einsum("ca,cai->ci", pe, g)
    ff = prm.f[:d]
    R = rho * (w + Cn) + gp[:, None, :] - rho * ff[None, None, :]
    beta = np.einsum("cbi,cai->cba", ub, g)
    mab = vol[:, None, None] * (1.0 + np.eye(n1))[None] / (n1 * (n1 + 1.0))
    M, Lm = tau_moments(une, vol, h, prm)
    mt = M.sum(axis=2)
    T = mt.sum(axis=1)
    Q = np.einsum("cbd,cbi->cdi", M, R)
    E = 0.5 * (G + np.transpose(G, (0, 2, 1)))
    pbar = pe.mean(axis=1)
    Fu = rho * np.einsum("cab,cbi->cai", mab, w + Cn)
    Fu -= rho * ff[None, None, :] * (vol / n1)[:, None, None]
    Fu += vol[:, None, None] * (2.0 * mu * np.einsum("cik,cak->cai", E, g) - pbar[:, None, None] * g)
    Fu += np.einsum("cda,cdi->cai", beta, Q)
    Fu += (rho * Lm * divu)[:, None, None] * g
    Fp = (vol / n1 * divu)[:, None] + (1.0 / rho) * np.einsum("cb,cbi,cai->ca", mt, R, g)
    Je = None
    if want_jac:
        Je = np.zeros((nc, nd, nd))
        MB = np.einsum("cbd,cda->cba", M, beta)
        BMB = np.einsum("cdb,cda->cba", beta, MB)
        mB = np.einsum("cad,cdb->cab", mab, beta)
        mtB = np.einsum("cd,cda->ca", mt, beta)
        gg = np.einsum("cai,cbi->cab", g, g)
        po = d * n1
        for a in range(n1):
            for b in range(n1):
                for i in range(d):
                    for j in range(d):
                        dij = 1.0 if i == j else 0.0
                        v = rho * mab[:, a, b] * dij * a0 / dt
                        v = v + rho * th * (mab[:, a, b] * G[:, j, i] + dij * mB[:, a, b])
                        v = v + vol * mu * th * (g[:, b, i] * g[:, a, j] + gg[:, a, b] * dij)
                        v = v + rho * ((dij * a0 / dt + th * G[:, j, i]) * MB[:, b, a] + th * dij * BMB[:, b, a])
                        v = v + th * g[:, a, j] * Q[:, b, i]
                        v = v + rho * Lm * th * g[:, b, j] * g[:, a, i]
                        Je[:, d * a + i, d * b + j] = v
                    Je[:, d * a + i, po + b] = -vol / n1 * g[:, a, i] + g[:, b, i] * mtB[:, a]
                for j in range(d):
                    Gg = np.einsum("ck,ck->c", G[:, j, :], g[:, a, :])
                    v = vol / n1 * th * g[:, b, j]
                    v = v + mt[:, b] * (g[:, a, j] * a0 / dt + th * Gg)
                    v = v + th * g[:, a, j] * mtB[:, b]
                    Je[:, po + a, d * b + j] = v
                Je[:, po + a, po + b] = T * gg[:, a, b] / rho
    if facet_flags is not None and prm.beta_backflow != 0.0 and np.any(facet_flags.astype(np.int64) >> n1):
        # backflow stabilisation (stabilized_schur_backflow.py:165-176): F -= beta rho int_out (u_prev.n)_- (ubar . v) ds,
        # (s)_- = (s - |s|)/2, on the facets flagged in bits n1 .. 2 n1 - 1.  UFL's estimated degree is 3 (abs keeps the
        # degree): 2-point Gauss-Legendre on an edge, a 6-point degree-3 rule on a triangle (facet_rule).
        QF, WF = facet_rule(d)
        for f in range(n1):
            sel = np.nonzero((facet_flags.astype(np.int64) >> (n1 + f)) & 1)[0]
            if len(sel) == 0:
                continue
            gf = g[sel, f]
            gl = np.linalg.norm(gf, axis=1)
            n = -gf / gl[:, None]
            fm = d * vol[sel] * gl
            ev = [a for a in range(n1) if a != f]
            sv = [np.einsum("ci,ci->c", une[sel, a], n) for a in ev]
            for lam, wq in zip(QF, WF):
                sq = sum(l * s_ for l, s_ in zip(lam, sv))
                cq = prm.beta_backflow * rho * 0.5 * (sq - np.abs(sq)) * wq * fm
                uq = sum(l * ub[sel, a] for l, a in zip(lam, ev))
                for la, a in zip(lam, ev):
                    Fu[sel, a, :] -= (cq * la)[:, None] * uq
                    if want_jac:
                        for lb, b in zip(lam, ev):
                            for i in range(d):
                                Je[sel, d * a + i, d * b + i] -= th * cq * la * lb
    if facet_flags is not None and prm.ds_terms and np.any(facet_flags & ((1 << n1) - 1)):
        po = d * n1
        for f in range(n1):
            sel = np.nonzero((facet_flags >> f) & 1)[0]
            if len(sel) == 0:
                continue
            gf = g[sel, f]
            gl = np.linalg.norm(gf, axis=1)
            n = -gf / gl[:, None]
            fm = d * vol[sel] * gl                       # facet measure
            Gn = np.einsum("cij,cj->ci", G[sel], n)      # sum_j d_i ubar_j n_j
            ev = [a for a in range(n1) if a != f]
            for a in ev:
                pint = sum(pe[sel, b] * (2.0 if a == b else 1.0) for b in ev) / (d * (d + 1.0))
                Fu[sel, a, :] += n * (fm * pint)[:, None] - muf * Gn * (fm / d)[:, None]
                if want_jac:
                    for i in range(d):
                        for b in ev:
                            Je[sel, d * a + i, po + b] += n[:, i] * fm * (2.0 if a == b else 1.0) / (d * (d + 1.0))
                        for b in range(n1):
                            for j in range(d):
                                Je[sel, d * a + i, d * b + j] -= muf * th * g[sel, b, i] * n[:, j] * fm / d
    Fe = np.concatenate([Fu.reshape(nc, d * n1), Fp], axis=1)
    return Fe, Je


class Problem:
    """Mesh + parameters + Dirichlet data in plain arrays (independent of the product); d from x.shape[1]."""

    def __init__(self, x, cells, facet_cells, facet_local, prm):
        self.x = np.ascontiguousarray(x, dtype=np.float64)
        self.d = self.x.shape[1]
        self.cells = np.ascontiguousarray(cells, dtype=np.int64)
        self.nv, self.nc = len(self.x), len(self.cells)
        self.prm = prm
        d, n1 = self.d, self.d + 1
        self.facet_cells = np.asarray(facet_cells, dtype=np.int64)
        self.facet_local = np.asarray(facet_local, dtype=np.int64)
        ff = np.zeros(self.nc, dtype=np.uint8)
        np.bitwise_or.at(ff, self.facet_cells, (1 << self.facet_local).astype(np.uint8))
        self.facet_flags = ff
        self._ext_flags = ff.copy()
        self.ndof = n1 * self.nv
        self.nu = d * self.nv
        c = self.cells
        ld = np.empty((self.nc, n1 * n1), dtype=np.int64)
        for a in range(n1):
            for i in range(d):
                ld[:, d * a + i] = d * c[:, a] + i
            ld[:, d * n1 + a] = self.nu + c[:, a]
        self.ldofs = ld
        self.isbc = np.zeros(self.ndof, dtype=bool)
        self.bcval = np.zeros(self.ndof)
        self.bcmult = np.zeros(self.ndof)

    def set_boundary_terms(self, ds_terms, backflow_facets=None, beta=0.0):
        """backflow_facets: indices into the exterior-facet arrays (the facets tagged `outlet`)."""
        self.prm.ds_terms = bool(ds_terms)
        self.prm.beta_backflow = float(beta)
        n1 = self.d + 1
        ff = self._ext_flags.copy()
        if backflow_facets is not None and len(backflow_facets):
            k = np.asarray(backflow_facets, dtype=np.int64)
            np.bitwise_or.at(ff, self.facet_cells[k], ((1 << n1) << self.facet_local[k]).astype(np.uint8))
        self.facet_flags = ff

    def clear_bcs(self):
        self.isbc[:] = False
        self.bcval[:] = 0.0
        self.bcmult[:] = 0.0

    def add_bc_u(self, nodes, values):
        nodes = np.asarray(nodes, dtype=np.int64)
        values = np.asarray(values, dtype=np.float64).reshape(-1, self.d)
        for i in range(self.d):
            k = self.d * nodes + i
            self.isbc[k] = True
            self.bcval[k] = values[:, i]
            self.bcmult[k] += 1.0

    def add_bc_p(self, nodes, values):
        k = self.nu + np.asarray(nodes, dtype=np.int64)
        self.isbc[k] = True
        self.bcval[k] = np.asarray(values, dtype=np.float64).reshape(-1)
        self.bcmult[k] += 1.0

    def split(self, xvec):
        return xvec[: self.nu].reshape(-1, self.d), xvec[self.nu:]

    def assemble(self, xvec, un, want_jac=True, apply_bc=True, un2=None):
        u, p = self.split(xvec)
        need_j = want_jac
        lift = None
        if apply_bc and self.isbc.any():
            lift = np.where(self.isbc, self.bcval - xvec, 0.0)
            if np.any(lift != 0.0):
                need_j = True
        Fe, Je = element_tensors(self.x, self.cells, u, np.asarray(un).reshape(-1, self.d), p, self.prm, self.facet_flags,
                                 want_jac=need_j, un2=None if un2 is None else np.asarray(un2).reshape(-1, self.d))
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
                # constant-pressure null vector?  Not with a pressure condition, and not with a do-nothing outlet either (the
                # weak form without the ds pair fixes the pressure level): decided on the matrix, relative to |J| e
                # (the product's criterion, csrc/cfdh_solver.cpp: PETSc's absolute bound on the normalised vector AND a relative one)
                e = np.zeros(self.ndof)
                e[self.nu:] = 1.0 / np.sqrt(self.ndof - self.nu)
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
                dx = spla.splu(A).solve(np.concatenate([F, [0.0]]))[:-1]
            else:
                dx = spla.splu(J.tocsc()).solve(F)
            x -= dx
        return x, hist

    def wall_shear_stress(self, xvec, mu=None):
        """assemble_wss of solverBase.py:163-195: (1/|f|) oint w . Tt ds with T = -sigma(u, p) n, Tt = T - (T.n) n, as a P1 vector
        field: every exterior facet gives Tt / d to each of its d vertices (the pressure part of T is normal and drops out)."""
        d, n1 = self.d, self.d + 1
        u, _ = self.split(xvec)
        mu = self.prm.mu if mu is None else mu
        g, vol, _ = geometry(self.x, self.cells)
        out = np.zeros((self.nv, d))
        for e, fl in zip(self.facet_cells, self.facet_local):
            gf = g[e, fl]
            n = -gf / np.linalg.norm(gf)
            G = np.einsum("ai,aj->ij", g[e], u[self.cells[e]])          # G_ij = d_i u_j
            T = -mu * (G + G.T) @ n
            Tt = T - (T @ n) * n
            for a in range(n1):
                if a != fl:
                    out[self.cells[e, a]] += Tt / d
        return out.ravel()

    def l2_norms(self, xvec):
        u, p = self.split(xvec)
        n1 = self.d + 1
        _, vol, _ = geometry(self.x, self.cells)
        mab = vol[:, None, None] * (1.0 + np.eye(n1))[None] / (n1 * (n1 + 1.0))
        ue, pe = u[self.cells], p[self.cells]
        return (np.sqrt(np.einsum("cab,cai,cbi->", mab, ue, ue)), np.sqrt(np.einsum("cab,ca,cb->", mab, pe, pe)))

    def flux(self, xvec, facets):
        """Outward volume flux through the given exterior facets (indices into facet_cells/facet_local)."""
        u, _ = self.split(xvec)
        fc, fl = self.facet_cells[facets], self.facet_local[facets]
        cells = self.cells[fc]
        g, vol, _ = geometry(self.x, cells)
        idx = np.arange(len(fc))
        gf = g[idx, fl]
        gl = np.linalg.norm(gf, axis=1)
        n = -gf / gl[:, None]
        fm = self.d * vol * gl
        usum = np.zeros((len(fc), self.d))
        for a in range(self.d + 1):
            usum += np.where((fl != a)[:, None], u[cells[:, a]], 0.0)
        return float(np.sum(fm * np.einsum("ci,ci->c", usum / self.d, n)))
