"""CPU study (numpy twin + scipy, exact sub-solves): why does the Cahouet-Chabard Schur approximation need 10x more
FGMRES iterations on P2/P2 than on P1/P1 at the same node count?  Builds the backflow stenosis on a short channel both ways,
takes the Jacobian of the second time step and counts right-preconditioned GMRES iterations (rtol 1e-5) for variants of the
block preconditioner with EXACT sub-solves, so that only the Schur form is measured.
Usage: python tools/p2_schur_study.py [ny_p2=8] [L=20] [v_max=20]"""
import os
import sys

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gen_util import facet_node_set, problem, stenosis_nodes  # noqa: E402
from oracle import np_twin as T, np_twin_gen as G, orcg  # noqa: E402

G.element_tensors = orcg.element_tensors  # C element routine: fast


def gmres_right(J, b, pc, rtol=1e-5, maxit=400):
    n = len(b)
    V = [b / np.linalg.norm(b)]
    Z = []
    H = np.zeros((maxit + 1, maxit))
    bn = np.linalg.norm(b)
    for j in range(maxit):
        z = pc(V[j])
        Z.append(z)
        w = J @ z
        for i in range(j + 1):
            H[i, j] = V[i] @ w
            w = w - H[i, j] * V[i]
        for i in range(j + 1):  # second pass
            d = V[i] @ w
            H[i, j] += d
            w = w - d * V[i]
        H[j + 1, j] = np.linalg.norm(w)
        V.append(w / H[j + 1, j])
        e = np.zeros(j + 2)
        e[0] = bn
        y, res, *_ = np.linalg.lstsq(H[: j + 2, : j + 1], e, rcond=None)
        r = np.linalg.norm(H[: j + 2, : j + 1] @ y - e)
        if r <= rtol * bn:
            return j + 1
    return maxit


def build(kind, ny, L, xs, vmax):
    m, ft = stenosis_nodes(kind, ny, L, xs)
    nv = m.num_vertices
    prm = T.Params(0.01, 1.06e-3, 3.5e-3, (0.0, 0.0), ds_terms=False, beta_backflow=0.2)
    pb = problem(kind, m, prm)
    pb.set_boundary_terms(False, ft.find(3), 0.2)
    wn = facet_node_set(m, ft.find(4))
    inn = facet_node_set(m, ft.find(2))
    outn = facet_node_set(m, ft.find(3))
    pb.add_bc_u(wn, np.zeros((len(wn), 2)))
    y = m.x[inn, 1]
    pb.add_bc_u(inn, np.stack([vmax * (1.0 - ((y - 1.57) / 1.57) ** 2), 0 * y], 1))
    x = np.zeros(3 * nv)
    un = np.zeros((nv, 2))
    for _ in range(2):
        x, _ = pb.newton(x, un)
        un = x[: 2 * nv].reshape(-1, 2).copy()
    F, J = pb.assemble(x, un)
    return m, pb, J.tocsr(), F, outn


def main():
    ny2 = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    L = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
    vmax = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
    for kind, ny in (("P1", 2 * ny2), ("P2", ny2)):
        m, pb, J, F, outn = build(kind, ny, L, 8.0, vmax)
        study(kind, m, pb, J, F, outn)


def study(kind, m, pb, J, F, outn):
    nv = pb.nv
    nu = 2 * nv
    A00, A01, A10, A11 = J[:nu, :nu].tocsc(), J[:nu, nu:].tocsc(), J[nu:, :nu].tocsc(), J[nu:, nu:].tocsc()
    prm = pb.prm
    # scalar stiffness / mass from the Jacobian structure itself: assemble a pure-diffusion, pure-mass problem with the twin
    Ls, Ms = laplace_mass(pb)
    isbc_p = pb.isbc[nu:]
    pd = np.zeros(nv, dtype=bool)
    pd[outn] = True  # outflow nodes: Dirichlet in L
    free = ~pd
    Lf = Ls.tolil()
    Ld = Ls.tocsr()[free][:, free].tocsc()
    Lsolve = spl.splu(Ld)
    Mdiag = Ms.diagonal()
    Ml_rowsum = np.asarray(Ms.sum(axis=1)).ravel()
    Ml_scaled = Mdiag * (Ms.sum() / Mdiag.sum())
    a_, b_ = prm.rho * prm.a0 / (prm.theta * prm.dt), prm.mu
    A00s = spl.splu(A00)
    Msolve = spl.splu(Ms.tocsc())
    S_exact = (A11 - A10 @ sp.csc_matrix(A00s.solve(A01.toarray()))).toarray() if nv < 6000 and not os.environ.get("QUICK") else None

    def make_pc(schur):
        def pc(r):
            zp = schur(r[nu:])
            zu = A00s.solve(r[:nu] - A01 @ zp)
            return np.concatenate([zu, zp])
        return pc

    def linv(v):
        out = np.zeros(nv)
        out[free] = Lsolve.solve(v[free])
        return out

    def cc(Ml, consistent=False, with_h=True):
        T_ = A11.diagonal() / Ls.diagonal()
        Hm = (sp.diags((1.0 + a_ * T_) * Ml) + b_ * A11).tocsc()
        Hs = spl.splu(Hm)

        def schur(rp):
            y = Hs.solve(rp) if with_h else rp / Ml
            if consistent:
                # (a' L^-1 + b' M^-1) M y  -> with H = (I + a'T) M + b' A11 consistent
                return a_ * linv(Ms @ y) + b_ * y
            return a_ * linv(Ml * y) + b_ * y
        return schur

    def cc_consistent():
        T_ = A11.diagonal() / Ls.diagonal()
        Hm = (sp.diags(np.sqrt(1.0 + a_ * T_)) @ Ms @ sp.diags(np.sqrt(1.0 + a_ * T_)) + b_ * A11).tocsc()
        Hs = spl.splu(Hm)
        return lambda rp: a_ * linv(Ms @ Hs.solve(rp)) + b_ * Hs.solve(rp)

    def exact(rp):
        return np.linalg.solve(S_exact, rp)

    def selfp():
        Sp = (A11 - A10 @ sp.diags(1.0 / A00.diagonal()) @ A01).tocsc()
        s = spl.splu(Sp)
        return s.solve

    def k_plus_a11(Ml, consistent):
        """S ~ K + A11 with K^-1 = a' L^-1 + b' M^-1 applied exactly: solve (K + A11) z = r by dense algebra."""
        Linv = np.zeros((nv, nv))
        Linv[np.ix_(free, free)] = np.linalg.inv(Ld.toarray())
        Minv = np.linalg.inv(Ms.toarray()) if consistent else np.diag(1.0 / Ml)
        Kinv = a_ * Linv + b_ * Minv
        K = np.linalg.pinv(Kinv)
        Sd = K + A11.toarray()
        return lambda rp: np.linalg.solve(Sd, rp)

    # ---- inexact pieces, one at a time (the product's: scalar proxy of A00, Jacobi V-cycles, Chebyshev(2) on H)
    proxy = (0.5 * (A00[0::2, 0::2] + A00[1::2, 1::2])).tocsc()
    proxys = spl.splu(proxy)

    def a00_proxy(ru):
        out = np.empty(nu)
        out[0::2] = proxys.solve(ru[0::2])
        out[1::2] = proxys.solve(ru[1::2])
        return out

    def jacobi_weights(A, nit=30):
        d = A.diagonal()
        v = np.random.default_rng(0).standard_normal(A.shape[0])
        for _ in range(nit):
            v = (A @ v) / d
            lam = np.linalg.norm(v)
            v /= lam
        return d, 1.0 / (1.1 * lam) * 4.0 / 3.0 * 1.0  # omega = 4/3 / rho(D^-1 A)

    def two_grid(A, Pc, ncyc=1, coarse_exact=True):
        """V(1,1) damped Jacobi + exact coarse solve on span(Pc) (None: smoothing only)."""
        A = A.tocsr()
        d, om = jacobi_weights(A)
        W = om / d
        Acs = spl.splu((Pc.T @ A @ Pc).tocsc()) if Pc is not None else None

        def apply(bv):
            x = np.zeros_like(bv)
            for _ in range(ncyc):
                x = x + W * (bv - A @ x)
                if Acs is not None:
                    x = x + Pc @ Acs.solve(Pc.T @ (bv - A @ x))
                x = x + W * (bv - A @ x)
            return x
        return apply

    def cheb(A, deg):
        A = A.tocsr()
        d = A.diagonal()
        v = np.random.default_rng(1).standard_normal(A.shape[0])
        for _ in range(30):
            v = (A @ v) / d
            lam = np.linalg.norm(v)
            v /= lam
        lmax, lmin = 1.1 * lam, 1.1 * lam / 8.0
        th, dl = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)

        def apply(bv):
            x = np.zeros_like(bv)
            r = bv.copy()
            sig = th / dl
            rho_ = 1.0 / sig
            dvec = r / d / th
            for k in range(deg):
                x = x + dvec
                if k == deg - 1:
                    break
                r = bv - A @ x
                rho_n = 1.0 / (2.0 * sig - rho_)
                dvec = rho_n * rho_ * dvec + 2.0 * rho_n / dl * (r / d)
                rho_ = rho_n
            return x
        return apply

    # P1 subspace of the P2 node set (vertices first, then edge nodes: interpolation 1 at the vertex, 1/2 + 1/2 on edges)
    Pc_full = None
    if kind == "P2":
        nvert = int(pb.cells[:, :3].max()) + 1
        rows, cols, vals = list(range(nvert)), list(range(nvert)), [1.0] * nvert
        seen = set()
        for c in pb.cells:
            for e, (i, j) in enumerate([(1, 2), (0, 2), (0, 1)]):
                n_ = int(c[3 + e])
                if n_ in seen:
                    continue
                seen.add(n_)
                rows += [n_, n_]; cols += [int(c[i]), int(c[j])]; vals += [0.5, 0.5]
        Pc_full = sp.csr_matrix((vals, (rows, cols)), shape=(nv, nvert))

    def restrict_cols(Pc, keep_rows):
        """Coarse space for a matrix with Dirichlet rows removed: drop coarse functions attached to removed vertices."""
        return Pc

    T_ = A11.diagonal() / Ls.diagonal()
    Hm = (sp.diags((1.0 + a_ * T_) * Ml_scaled) + b_ * A11).tocsc()
    Hs = spl.splu(Hm)

    def variant(h_solve, l_solve, a_solve):
        def pc(r):
            y = h_solve(r[nu:])
            zp = a_ * l_solve(Ml_scaled * y) + b_ * y
            zu = a_solve(r[:nu] - A01 @ zp)
            return np.concatenate([zu, zp])
        return pc

    def l_tg(ncyc, pmg):
        Pc = None
        if pmg and Pc_full is not None:
            Pc = Pc_full[free]
        tg = two_grid(Ld, Pc, ncyc)

        def f(v):
            out = np.zeros(nv)
            out[free] = tg(v[free])
            return out
        return f

    def a_tg(ncyc, pmg):
        Pc = Pc_full if (pmg and Pc_full is not None) else None
        tg = two_grid(proxy, Pc, ncyc)

        def f(ru):
            out = np.empty(nu)
            out[0::2] = tg(ru[0::2]); out[1::2] = tg(ru[1::2])
            return out
        return f

    b = -F
    res = {}
    # PSPG cancellation carried explicitly: A11' = A11 - C diag(A00)^-1 A01, C = A10 + A01^T (the stabilisation part of A10)
    Cst = (A10 + A01.T).tocsr()
    for dname, Dinv in (("diag(A00)", sp.diags(1.0 / A00.diagonal())), ("0.25 diag(A00)", sp.diags(0.25 / A00.diagonal())),
                        ("0.5 diag(A00)", sp.diags(0.5 / A00.diagonal())), ("0.1 diag(A00)", sp.diags(0.1 / A00.diagonal()))):
        A11c = (A11 - Cst @ Dinv @ A01).tocsc()
        for kt in (1.0,):
            Hk = (sp.diags((1.0 + kt * a_ * T_) * Ml_scaled) + b_ * A11c).tocsc()
            res["CC exact, H'' = (1 + %g a'T) M + b' (A11 - C %s^-1 A01)" % (kt, dname)] = gmres_right(J, b, variant(spl.splu(Hk).solve, linv, A00s.solve))
    if kind == "P2":
        nvert_ = Pc_full.shape[1]
        Rinj = sp.csr_matrix((np.ones(nvert_), (np.arange(nvert_), np.arange(nvert_))), shape=(nvert_, nv))
        E = (sp.identity(nv) - Pc_full @ Rinj).tocsr()          # hierarchical surplus: 0 at vertices, value - mean of the edge ends
        A11hp = (E.T @ A11 @ E).tocsc()
        for c_all, c_sur in ((0.25, 0.6), (0.2, 0.7), (0.3, 0.5), (0.15, 0.8)):
            for kt in (1.0, 0.0):
                Hk = (sp.diags((1.0 + kt * a_ * T_) * Ml_scaled) + b_ * (c_all * A11 + c_sur * A11hp)).tocsc()
                res["CC exact, H = (1 + %g a'T) M + b' (%g A11 + %g E^T A11 E)" % (kt, c_all, c_sur)] = gmres_right(J, b, variant(spl.splu(Hk).solve, linv, A00s.solve))
        for kt in ():
            for kap in (1.0, 0.8, 0.6):
                Hk = (sp.diags((1.0 + kt * a_ * T_) * Ml_scaled) + kap * b_ * A11hp).tocsc()
                res["CC exact, H' = (1 + %g a'T) M + %g b' E^T A11 E" % (kt, kap)] = gmres_right(J, b, variant(spl.splu(Hk).solve, linv, A00s.solve))
    for kap in ((2.0, 4.0) if not os.environ.get("QUICK") else ()):
        for kt in (1.0,):
            Hk = (sp.diags((1.0 + kt * a_ * T_) * Ml_scaled) + kap * b_ * A11).tocsc()
            res["CC exact, H = (1 + %g a'T) M + %g b' A11" % (kt, kap)] = gmres_right(J, b, variant(spl.splu(Hk).solve, linv, A00s.solve))
    res["proxy exact for A00, rest exact"] = gmres_right(J, b, variant(Hs.solve, linv, a00_proxy))
    res["H by Chebyshev(2), rest exact"] = gmres_right(J, b, variant(cheb(Hm, 2), linv, A00s.solve))
    res["H by Chebyshev(4), rest exact"] = gmres_right(J, b, variant(cheb(Hm, 4), linv, A00s.solve))
    if kind == "P2":
        res["L: Jacobi V(1,1) + exact P1 coarse, rest exact"] = gmres_right(J, b, variant(Hs.solve, l_tg(1, True), A00s.solve))
        res["A00: proxy Jacobi V(1,1) + exact P1 coarse, rest exact"] = gmres_right(J, b, variant(Hs.solve, linv, a_tg(1, True)))
        res["all three inexact (P1 coarse exact)"] = gmres_right(J, b, variant(cheb(Hm, 2), l_tg(1, True), a_tg(1, True)))
    if S_exact is not None:
        res["exact S"] = gmres_right(J, b, make_pc(exact))
    res["CC, M_l scaled diag (product)"] = gmres_right(J, b, make_pc(cc(Ml_scaled)))
    res["CC, M_l row sums"] = gmres_right(J, b, make_pc(cc(np.where(np.abs(Ml_rowsum) > 1e-14, Ml_rowsum, 1e-14)))) if kind == "P1" else None
    res["CC, consistent mass in H and in L rhs"] = gmres_right(J, b, make_pc(cc_consistent()))
    if not os.environ.get("QUICK"):
        res["SELFP exact"] = gmres_right(J, b, make_pc(selfp()))
    if nv < 4000 and not os.environ.get("QUICK"):
        res["(K + A11)^-1, K^-1 = a'L^-1 + b'Ml^-1"] = gmres_right(J, b, make_pc(k_plus_a11(Ml_scaled, False)))
        res["(K + A11)^-1, consistent M"] = gmres_right(J, b, make_pc(k_plus_a11(Ml_scaled, True)))
    print(kind, "nodes", nv, "ndof", 3 * nv)
    for k, v in res.items():
        print("   %-45s %s" % (k, v))
    if S_exact is not None and nv < 4000:
        # spectrum of S_cc^-1 S
        sc = cc(Ml_scaled)
        Pm = np.column_stack([sc(e) for e in np.eye(nv)])
        ev = np.linalg.eigvals(Pm @ S_exact)
        ev = ev[np.argsort(np.abs(ev))]
        print("   |eig(S_cc^-1 S)|: min %.3g max %.3g; #<0.3: %d, #>3: %d of %d" % (np.abs(ev).min(), np.abs(ev).max(), (np.abs(ev) < 0.3).sum(), (np.abs(ev) > 3).sum(), nv))


def laplace_mass(pb):
    """Scalar stiffness / consistent mass on the node graph by the twin's tabulation (affine cells)."""
    el = pb.el
    Jinv, adet, h, det = G.cell_geometry(el, pb.x, pb.cells)
    g = np.einsum("qak,cki->cqai", el.dphi, Jinv)              # physical gradients [nc, nq, nl, 2]
    wq = el.meas * el.w                                          # [nq]
    Le = np.einsum("q,c,cqai,cqbi->cab", wq, adet, g, g)
    Me = np.einsum("q,c,qa,qb->cab", wq, adet, el.phi, el.phi)
    nl = el.nloc
    rows = np.repeat(pb.cells, nl, axis=1).ravel()
    cols = np.tile(pb.cells, (1, nl)).ravel()
    Ls = sp.coo_matrix((Le.ravel(), (rows, cols)), shape=(pb.nv, pb.nv)).tocsr()
    Ms = sp.coo_matrix((Me.ravel(), (rows, cols)), shape=(pb.nv, pb.nv)).tocsr()
    return Ls, Ms


if __name__ == "__main__":
    main()
