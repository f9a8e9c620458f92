"""NumPy/SciPy twin of the CPU oracle  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
anything under oracle/.  PARITY UNPINNED: the reference (FEniCSx 0.9 + PETSc,
/root/reference/singularity.def:2) cannot be built or imported in this
container and holds no golden vector for this path; what
pins this restatement instead is listed in DESIGN.md ("Oracle").

Restates, for affine P1/P1 triangles, the residual of
/root/reference/src/solvers/stabilized_schur.py:67-123 (Galerkin + exterior
facet terms + SUPG + PSPG + LSIC), its exact Gateaux derivative (:185-189), and
the block assembly / Dirichlet semantics of :144-175
(`assemble_vector_block(..., x0=x, alpha=-1)`), then solves each Newton step
with a DIRECT sparse factorisation (scipy splu), so the discrete solution is
known to round-off.  Algebra: SURVEY.md Appendix A.

Monolithic ordering (stabilized_schur.py:194-196,237-252): all velocity dofs,
vertex-major / component-minor, then all pressure dofs.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
from scipy.special import roots_jacobi, roots_legendre

EPS_VNORM = 1e-15  # np.finfo(float64).resolution, stabilized_schur.py:100


def quad_rule(n=7):
    """Same rule as tools/gen_quadrature.py (collapsed Gauss, weights sum to 1)."""
    tj, wj = roots_jacobi(n, 1.0, 0.0)
    tl, wl = roots_legendre(n)
    u = 0.5 * (tj + 1.0)
    v = 0.5 * (tl + 1.0)
    U, V = np.meshgrid(u, v, indexing="ij")
    W = np.outer(0.25 * wj, 0.5 * wl) * 2.0
    x = U.ravel()
    y = (V * (1.0 - U)).ravel()
    return np.stack([1.0 - x - y, x, y], axis=1), W.ravel()


_QL, _QW = quad_rule()


class Params:
    """theta, a0, a1, a2: time scheme.  Base solver (stabilized_schur.py:72-80): spatial terms at
    u_mid = (u + u_n)/2 -> theta = 1/2, time term (u - u_n)/dt -> (a0,a1,a2) = (1,-1,0).
    stabilized_schur_bdf2.py:79-110: fully implicit u_mid = u -> theta = 1, time term
    (a0 u + a1 u_n + a2 u_nm1)/dt with BDF1 (1,-1,0) on the first step, BDF2 (1.5,-2,0.5) after."""

    def __init__(self, dt, rho, mu, f=(0.0, 0.0), mu_facet=None, theta=0.5, a0=1.0, a1=-1.0, a2=0.0,
                 ds_terms=True, beta_backflow=0.0):
        self.theta, self.a0, self.a1, self.a2 = float(theta), float(a0), float(a1), float(a2)
        # ds_terms: the `dot(p n, v) ds - dot(mu grad(u_mid) n, v) ds` pair of stabilized_schur.py:79 on ALL
        # exterior facets; stabilized_schur_backflow.py:107 drops it (do-nothing outlet) and adds
        # -beta rho (u_prev.n)_- (u_mid . v) on the outlet facets (:165-176).
        self.ds_terms = bool(ds_terms)
        self.beta_backflow = float(beta_backflow)
        self.dt = float(dt)
        self.rho = float(rho)
        self.mu = float(mu)
        self.f = np.asarray(f, dtype=np.float64)[:2]
        # stabilized_schur.py:79 uses the raw python float `mu` in the ds term
        self.mu_facet = float(mu if mu_facet is None else mu_facet)


def geometry(x, cells):
    """grad(lambda_a) [nc,3,2], area [nc], h [nc] (max vertex distance, :85-88)."""
    p = x[cells]
    x0, x1, x2 = p[:, 0], p[:, 1], p[:, 2]
    det = (x1[:, 0] - x0[:, 0]) * (x2[:, 1] - x0[:, 1]) - (x1[:, 1] - x0[:, 1]) * (x2[:, 0] - x0[:, 0])
    g = np.empty((len(cells), 3, 2))
    g[:, 0, 0] = (x1[:, 1] - x2[:, 1]) / det
    g[:, 0, 1] = (x2[:, 0] - x1[:, 0]) / det
    g[:, 1, 0] = (x2[:, 1] - x0[:, 1]) / det
    g[:, 1, 1] = (x0[:, 0] - x2[:, 0]) / det
    g[:, 2, 0] = (x0[:, 1] - x1[:, 1]) / det
    g[:, 2, 1] = (x1[:, 0] - x0[:, 0]) / det
    area = 0.5 * np.abs(det)
    d01 = np.linalg.norm(x0 - x1, axis=1)
    d12 = np.linalg.norm(x1 - x2, axis=1)
    d20 = np.linalg.norm(x2 - x0, axis=1)
    h = np.maximum(d01, np.maximum(d12, d20))
    return g, area, h


def tau_moments(un, area, h, prm):
    """M_ab = int_K tau l_a l_b  [nc,3,3],  L = int_K tau_L [nc].

    tau   : stabilized_schur.py:100-108   (depends on u_prev only)
    tau_L : stabilized_schur.py:116-118
    """
    nu = prm.mu / prm.rho
    uq = np.einsum("qa,caj->cqj", _QL, un)  # u_n at quadrature points
    s = np.einsum("cqj,cqj->cq", uq, uq)
    hh = h[:, None]
    t1 = np.maximum(4.0 * s, EPS_VNORM**2) / (hh * hh)  # 1/tau1^2
    t2 = 4.0 / (prm.dt * prm.dt)
    t3 = 16.0 * nu * nu / (hh**4)
    tau = 1.0 / np.sqrt(t1 + t2 + t3)
    vn = np.sqrt(s)
    Re = vn * hh / (2.0 * nu)
    z = np.where(Re <= 3.0, Re / 3.0, 1.0)
    tauL = vn * hh * z / 2.0
    M = area[:, None, None] * np.einsum("q,cq,qa,qb->cab", _QW, tau, _QL, _QL)
    Lm = area * np.einsum("q,cq->c", _QW, tauL)
    return M, Lm


def element_tensors(x, cells, u, un, p, prm, facet_flags=None, want_jac=True, un2=None):
    """Element residual Fe [nc,9] and Jacobian Je [nc,9,9].

    Local dof order: velocity (a,i) -> 2a+i (0..5), pressure a -> 6+a.
    u, un: [nv,2]; p: [nv]; facet_flags: uint8 [nc], bit f set when the facet
    opposite local vertex f is an exterior facet.
    """
    rho, mu, dt, muf = prm.rho, prm.mu, prm.dt, prm.mu_facet
    g, area, h = geometry(x, cells)
    nc = len(cells)
    ue, une, pe = u[cells], un[cells], p[cells]
    th, a0 = prm.theta, prm.a0
    ub = th * ue + (1.0 - th) * une
    w = (a0 * ue + prm.a1 * une) / dt
    if prm.a2 != 0.0:
        w = w + prm.a2 * un2[cells] / dt
    G = np.einsum("cai,caj->cij", g, ub)  # G_ij = d_i ubar_j
    divu = G[:, 0, 0] + G[:, 1, 1]
    Cn = np.einsum("cai,cij->caj", ub, G)
    gp = np.einsum("ca,cai->ci", pe, g)
    R = rho * (w + Cn) + gp[:, None, :] - rho * prm.f[None, None, :]
    beta = np.einsum("cbi,cai->cba", ub, g)  # beta[c,a] = ubar_c . g_a
    mab = area[:, None, None] * (1.0 + np.eye(3))[None] / 12.0
    M, Lm = tau_moments(une, area, h, prm)
    mt = M.sum(axis=2)
    T = mt.sum(axis=1)
    Q = np.einsum("cbd,cbi->cdi", M, R)  # Q[d,i] = sum_b M[b,d] R[b,i]
    E = 0.5 * (G + np.transpose(G, (0, 2, 1)))
    pbar = pe.mean(axis=1)

    Fe = np.zeros((nc, 9))
    Fu = rho * np.einsum("cab,cbi->cai", mab, w + Cn)
    Fu -= rho * prm.f[None, None, :] * (area / 3.0)[:, None, None]
    Fu += area[:, None, None] * (2.0 * mu * np.einsum("cik,cak->cai", E, g) - pbar[:, None, None] * g)
    Fu += np.einsum("cda,cdi->cai", beta, Q)
    Fu += (rho * Lm * divu)[:, None, None] * g
    Fp = (area / 3.0 * divu)[:, None] + (1.0 / rho) * np.einsum("cb,cbi,cai->ca", mt, R, g)

    Je = np.zeros((nc, 9, 9)) if want_jac else None
    if want_jac:
        MB = np.einsum("cbd,cda->cba", M, beta)  # sum_d M[b,d] beta[d,a]
        BMB = np.einsum("cdb,cda->cba", beta, MB)  # [b,a] = sum_cd M[c,d] beta[c,b] beta[d,a]
        mB = np.einsum("cad,cdb->cab", mab, beta)  # sum_c m[a,c] beta[c,b]
        mtB = np.einsum("cd,cda->ca", mt, beta)  # sum_d mt[d] beta[d,a]
        gg = np.einsum("cai,cbi->cab", g, g)
        for a in range(3):
            for b in range(3):
                for i in range(2):
                    for j in range(2):
                        dij = 1.0 if i == j else 0.0
                        v = rho * mab[:, a, b] * dij * a0 / dt
                        v = v + rho * th * (mab[:, a, b] * G[:, j, i] + dij * mB[:, a, b])
                        v = v + area * mu * th * (g[:, b, i] * g[:, a, j] + gg[:, a, b] * dij)
                        v = v + rho * ((dij * a0 / dt + th * G[:, j, i]) * MB[:, b, a] + th * dij * BMB[:, b, a])
                        v = v + th * g[:, a, j] * Q[:, b, i]
                        v = v + rho * Lm * th * g[:, b, j] * g[:, a, i]
                        Je[:, 2 * a + i, 2 * b + j] = v
                    # J_up
                    Je[:, 2 * a + i, 6 + b] = -area / 3.0 * g[:, a, i] + g[:, b, i] * mtB[:, a]
                for j in range(2):
                    Gg = G[:, j, 0] * g[:, a, 0] + G[:, j, 1] * g[:, a, 1]
                    v = area / 3.0 * th * g[:, b, j]
                    v = v + mt[:, b] * (g[:, a, j] * a0 / dt + th * Gg)
                    v = v + th * g[:, a, j] * np.einsum("cd,cd->c", mt, beta[:, :, b])
                    Je[:, 6 + a, 2 * b + j] = v
                Je[:, 6 + a, 6 + b] = T * gg[:, a, b] / rho

    if facet_flags is not None and prm.beta_backflow != 0.0 and np.any(facet_flags >> 3):
        # backflow stabilisation (stabilized_schur_backflow.py:165-176; Moghadam et al. 2011 eq. 10):
        #   F -= beta rho int_out (u_prev.n)_- (ubar . v) ds,  (s)_- = (s - |s|)/2.
        # UFL estimates degree(abs(x)) = degree(x): 1 + 1 + 1 = 3 -> 2-point Gauss-Legendre per facet.
        gq = 0.5 / np.sqrt(3.0)
        for f in range(3):
            sel = np.nonzero((facet_flags >> (3 + f)) & 1)[0]
            if len(sel) == 0:
                continue
            gf = g[sel, f]
            gl = np.linalg.norm(gf, axis=1)
            n = -gf / gl[:, None]
            elen = 2.0 * area[sel] * gl
            a1, a2 = (f + 1) % 3, (f + 2) % 3
            s1 = np.einsum("ci,ci->c", une[sel, a1], n)
            s2 = np.einsum("ci,ci->c", une[sel, a2], n)
            for t in (0.5 - gq, 0.5 + gq):
                lam = {a1: 1.0 - t, a2: t}
                sq = (1.0 - t) * s1 + t * s2
                cq = prm.beta_backflow * rho * 0.5 * (sq - np.abs(sq)) * 0.5 * elen
                uq = (1.0 - t) * ub[sel, a1] + t * ub[sel, a2]
                for a in (a1, a2):
                    Fu[sel, a, :] -= (cq * lam[a])[:, None] * uq
                    if want_jac:
                        for b in (a1, a2):
                            for i in range(2):
                                Je[sel, 2 * a + i, 2 * b + i] -= th * cq * lam[a] * lam[b]
    if facet_flags is not None and prm.ds_terms and np.any(facet_flags & 7):
        for f in range(3):
            sel = np.nonzero((facet_flags >> f) & 1)[0]
            if len(sel) == 0:
                continue
            gf = g[sel, f]
            gl = np.linalg.norm(gf, axis=1)
            n = -gf / gl[:, None]
            elen = 2.0 * area[sel] * gl
            Gn = np.einsum("cij,cj->ci", G[sel], n)  # sum_j d_i ubar_j n_j
            ev = [(f + 1) % 3, (f + 2) % 3]
            for a in ev:
                pint = sum(pe[sel, b] * (2.0 if a == b else 1.0) for b in ev) / 6.0
                Fu[sel, a, :] += n * (elen * pint)[:, None] - muf * Gn * (elen / 2.0)[:, None]
                if want_jac:
                    for i in range(2):
                        for b in ev:
                            Je[sel, 2 * a + i, 6 + b] += n[:, i] * elen * (2.0 if a == b else 1.0) / 6.0
                        for b in range(3):
                            for j in range(2):
                                Je[sel, 2 * a + i, 2 * b + j] -= muf * th * g[sel, b, i] * n[:, j] * elen / 2.0
    Fe[:, 0:6] = Fu.reshape(nc, 6)
    Fe[:, 6:9] = Fp
    return Fe, Je


class Problem:
    """Mesh + parameters + Dirichlet data in plain arrays (independent of the product)."""

    def set_boundary_terms(self, ds_terms, backflow_facets=None, beta=0.0):
        """backflow_facets: indices into the exterior-facet arrays (the facets tagged `outlet`)."""
        self.prm.ds_terms = bool(ds_terms)
        self.prm.beta_backflow = float(beta)
        ff = self._ext_flags.copy()
        if backflow_facets is not None and len(backflow_facets):
            k = np.asarray(backflow_facets, dtype=np.int64)
            np.bitwise_or.at(ff, self.facet_cells[k], (8 << self.facet_local[k]).astype(np.uint8))
        self.facet_flags = ff

    def __init__(self, x, cells, facet_cells, facet_local, prm):
        self.x = np.ascontiguousarray(x, dtype=np.float64)
        self.cells = np.ascontiguousarray(cells, dtype=np.int64)
        self.nv = len(self.x)
        self.nc = len(self.cells)
        self.prm = prm
        self.facet_cells = np.asarray(facet_cells, dtype=np.int64)
        self.facet_local = np.asarray(facet_local, dtype=np.int64)
        ff = np.zeros(self.nc, dtype=np.uint8)
        np.bitwise_or.at(ff, self.facet_cells, (1 << self.facet_local).astype(np.uint8))
        self.facet_flags = ff
        self._ext_flags = ff.copy()
        self.ndof = 3 * self.nv
        c = self.cells
        ld = np.empty((self.nc, 9), dtype=np.int64)
        for a in range(3):
            ld[:, 2 * a] = 2 * c[:, a]
            ld[:, 2 * a + 1] = 2 * c[:, a] + 1
            ld[:, 6 + a] = 2 * self.nv + c[:, a]
        self.ldofs = ld
        self.isbc = np.zeros(self.ndof, dtype=bool)
        self.bcval = np.zeros(self.ndof)
        self.bcmult = np.zeros(self.ndof)

    def clear_bcs(self):
        self.isbc[:] = False
        self.bcval[:] = 0.0
        self.bcmult[:] = 0.0

    def add_bc_u(self, nodes, values):
        """One DirichletBC object on velocity: vertex ids + [n,2] values.  Later
        objects overwrite the value, the diagonal counts the objects
        (DOLFINx block assembly adds 1.0 per bc object)."""
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

    # -- assembly ------------------------------------------------------------
    def split(self, xvec):
        return xvec[: 2 * self.nv].reshape(-1, 2), xvec[2 * self.nv :]

    def assemble(self, xvec, un, want_jac=True, apply_bc=True, un2=None):
        """F (and J as scipy CSR) at the monolithic state `xvec`, previous
        velocity `un` [nv,2]; Dirichlet handling of stabilized_schur.py:157-175."""
        u, p = self.split(xvec)
        need_j = want_jac
        lift = None
        if apply_bc and self.isbc.any():
            lift = np.where(self.isbc, self.bcval - xvec, 0.0)
            if np.any(lift != 0.0):
                need_j = True
        Fe, Je = element_tensors(self.x, self.cells, u, np.asarray(un).reshape(-1, 2), p, self.prm,
                                 self.facet_flags, want_jac=need_j,
                                 un2=None if un2 is None else np.asarray(un2).reshape(-1, 2))
        ld = self.ldofs
        if apply_bc and self.isbc.any():
            bce = self.isbc[ld]  # [nc,9]
            if lift is not None and np.any(lift != 0.0):
                Fe = Fe + np.einsum("crk,ck->cr", Je, lift[ld])
            Fe = np.where(bce, 0.0, Fe)
            if Je is not None:
                Je = Je * (~bce)[:, :, None] * (~bce)[:, None, :]
        F = np.zeros(self.ndof)
        np.add.at(F, ld.ravel(), Fe.ravel())
        J = None
        if want_jac:
            rows = np.repeat(ld, 9, axis=1).ravel()
            cols = np.tile(ld, (1, 9)).ravel()
            J = sp.coo_matrix((Je.ravel(), (rows, cols)), shape=(self.ndof, self.ndof)).tocsr()
            J.sum_duplicates()
        if apply_bc and self.isbc.any():
            F[self.isbc] = (xvec - self.bcval)[self.isbc]
            if J is not None:
                J = J + sp.diags(np.where(self.isbc, self.bcmult, 0.0))
                J = J.tocsr()
        return F, J

    # -- Newton with a direct solve -----------------------------------------
    def newton(self, x0, un, rtol=1e-12, atol=1e-14, max_it=25, remove_p_mean=False, verbose=False, un2=None):
        x = x0.copy()
        hist = []
        singular = not self.isbc[2 * self.nv :].any()
        for it in range(max_it + 1):
            F, J = self.assemble(x, un, want_jac=True, un2=un2)
            fn = np.linalg.norm(F)
            hist.append(fn)
            if verbose:
                print("  twin newton %d |F| = %.3e" % (it, fn))
            if fn <= atol or (it > 0 and fn <= rtol * hist[0]):
                break
            if it == max_it:
                raise RuntimeError("twin newton did not converge: %r" % hist)
            if singular:
                # pin the constant-pressure mode with a Lagrange multiplier
                e = np.zeros(self.ndof)
                e[2 * self.nv :] = 1.0
                A = sp.bmat([[J, sp.csr_matrix(e[:, None])], [sp.csr_matrix(e[None, :]), None]]).tocsc()
                d = spla.splu(A).solve(np.concatenate([F, [0.0]]))[:-1]
            else:
                d = spla.splu(J.tocsc()).solve(F)
            x -= d
        return x, hist

    # -- functionals ----------------------------------------------------------
    def l2_norms(self, xvec):
        """sqrt(int u.u), sqrt(int p^2)  (/root/reference/src/scenario.py:315-324)."""
        u, p = self.split(xvec)
        _, area, _ = geometry(self.x, self.cells)
        mab = area[:, None, None] * (1.0 + np.eye(3))[None] / 12.0
        ue = u[self.cells]
        pe = p[self.cells]
        nu2 = np.einsum("cab,cai,cbi->", mab, ue, ue)
        np2 = np.einsum("cab,ca,cb->", mab, pe, pe)
        return np.sqrt(nu2), np.sqrt(np2)

    def drag_lift(self, xvec, facets, mu):
        """F_D, F_L of /root/reference/src/scenarios/dfg_1.py:183-202 over the
        given exterior facets (indices into facet_cells/facet_local);
        the scenario reports 500*F_D, 500*F_L."""
        u, p = self.split(xvec)
        fc = self.facet_cells[facets]
        fl = self.facet_local[facets]
        cells = self.cells[fc]
        g, area, _ = geometry(self.x, cells)
        idx = np.arange(len(fc))
        gf = g[idx, fl]
        gl = np.linalg.norm(gf, axis=1)
        nout = -gf / gl[:, None]
        n = -nout  # n = -FacetNormal
        elen = 2.0 * area * gl
        t = np.stack([n[:, 1], -n[:, 0]], axis=1)
        ue = u[cells]
        ut = np.einsum("cai,ci->ca", ue, t)
        gut = np.einsum("ca,cai->ci", ut, g)
        dn = np.einsum("ci,ci->c", gut, n)
        a1 = cells[idx, (fl + 1) % 3]
        a2 = cells[idx, (fl + 2) % 3]
        pm = 0.5 * (p[a1] + p[a2])
        FD = np.sum(elen * (mu * dn * n[:, 1] - pm * n[:, 0]))
        FL = -np.sum(elen * (mu * dn * n[:, 0] + pm * n[:, 1]))
        return FD, FL


def time_loop(prob, un0, nsteps, x_guess=None, remove_p_mean=True, newton_kw=None):
    """`Scenario.solve` + `solveStep` skeleton (scenario.py:221-307,
    stabilized_schur.py:313-321): initial guess carried over, its pressure mean
    removed every step, u_prev <- u_sol after the step."""
    newton_kw = newton_kw or {}
    nv = prob.nv
    un = np.asarray(un0, dtype=np.float64).reshape(-1, 2).copy()
    x = np.zeros(prob.ndof) if x_guess is None else x_guess.copy()
    hists = []
    for _ in range(nsteps):
        if remove_p_mean:
            x[2 * nv :] -= x[2 * nv :].mean()
        x, hist = prob.newton(x, un, **newton_kw)
        hists.append(hist)
        un = x[: 2 * nv].reshape(-1, 2).copy()
    return x, hists
