"""Triangle mesh of an implicitly given planar domain {phi < 0}: a uniform background triangulation is cut by the
zero level set, with vertices close to it snapped onto it first ("isosurface stuffing" reduced to 2-D).  Stands
where the reference calls gmsh on an OpenCASCADE boolean union (/root/reference/src/scenarios/stenosis_with_tree.py:
421-497); gmsh is not available on the GPU box, and a deterministic generator is needed so that the CPU oracle and the
HIP path see the same mesh.

Algorithm (all steps vectorised NumPy, deterministic):
  1. background grid of squares of size h over the bounding box, squares split in a union-jack pattern;
  2. phi at the grid vertices; every grid edge with a sign change gets its crossing point (regula falsi on phi);
  3. a vertex with a crossing closer than alpha * |edge| is moved onto the nearest such crossing (phi := 0);
  4. the remaining cut edges (crossing in the middle part of the edge) receive a new vertex at the crossing;
  5. every triangle is clipped against {phi <= 0}: kept, dropped, or replaced by the one or two triangles of its
     inside part -- neighbouring triangles share the crossing vertices, so the result is conforming;
  6. unused vertices are dropped.
With alpha = 0.3 the smallest angle stays above ~10 degrees on the shapes used here (checked by `mesh_quality`).
"""
from __future__ import annotations

import numpy as np


def _grid(bbox, h):
    x0, y0, x1, y1 = bbox
    nx = max(1, int(np.ceil((x1 - x0) / h - 1e-9)))
    ny = max(1, int(np.ceil((y1 - y0) / h - 1e-9)))
    xs = x0 + h * np.arange(nx + 1)
    ys = y0 + h * np.arange(ny + 1)
    return nx, ny, xs, ys


def _refine_crossings(phi, a, b, fa, fb, its):
    """Points on the segments a->b (phi(a) < 0 < phi(b)) where phi vanishes: Illinois regula falsi."""
    ta, tb = np.zeros(len(a)), np.ones(len(a))
    fa, fb = fa.copy(), fb.copy()
    t = fa / (fa - fb)
    side = np.zeros(len(a), dtype=np.int8)
    for _ in range(its):
        f = phi(a + t[:, None] * (b - a))
        neg = f < 0
        # shrink the bracket; Illinois: halve the stale end's function value when the same end moved twice
        stale_b = neg & (side == -1)
        stale_a = (~neg) & (side == 1)
        fb = np.where(stale_b, 0.5 * fb, fb)
        fa = np.where(stale_a, 0.5 * fa, fa)
        ta = np.where(neg, t, ta)
        fa = np.where(neg, f, fa)
        tb = np.where(neg, tb, t)
        fb = np.where(neg, fb, f)
        side = np.where(neg, -1, 1).astype(np.int8)
        den = fa - fb
        t = np.where(den != 0, ta + (tb - ta) * fa / np.where(den != 0, den, 1.0), 0.5 * (ta + tb))
        t = np.clip(t, ta, tb)
    return t


def mesh_implicit_domain(phi, bbox, h, alpha=0.3, refine_its=6, fill=None):
    """phi: callable points[n,2] -> float[n] (negative inside).  bbox = (x0, y0, x1, y1).  `fill(xs, ys)` may return
    phi on the whole grid [ny+1, nx+1] faster than phi on all points (optional).  Returns x [nv,2], cells [nc,3]
    (counter-clockwise)."""
    nx, ny, xs, ys = _grid(bbox, h)
    NX = nx + 1
    if fill is not None:
        F = np.asarray(fill(xs, ys), dtype=np.float64)
    else:
        X, Y = np.meshgrid(xs, ys, indexing="xy")
        F = phi(np.stack([X.ravel(), Y.ravel()], 1)).reshape(ny + 1, nx + 1)
    # squares with at least one inside corner
    ins = F < 0
    act = ins[:-1, :-1] | ins[:-1, 1:] | ins[1:, :-1] | ins[1:, 1:]
    jj, ii = np.nonzero(act)
    v00 = jj * NX + ii
    v10, v01, v11 = v00 + 1, v00 + NX, v00 + NX + 1
    even = ((ii + jj) & 1) == 0
    t1 = np.where(even[:, None], np.stack([v00, v10, v11], 1), np.stack([v00, v10, v01], 1))
    t2 = np.where(even[:, None], np.stack([v00, v11, v01], 1), np.stack([v10, v11, v01], 1))
    T = np.concatenate([t1, t2]).astype(np.int64)
    nvg = (nx + 1) * (ny + 1)
    P = np.empty((nvg, 2))
    P[:, 0] = np.tile(xs, ny + 1)
    P[:, 1] = np.repeat(ys, nx + 1)
    f = F.ravel().copy()
    # ---- unique edges of the active triangles; edge k of a triangle is opposite to its corner k
    ea = np.concatenate([T[:, 1], T[:, 2], T[:, 0]])
    eb = np.concatenate([T[:, 2], T[:, 0], T[:, 1]])
    lo, hi = np.minimum(ea, eb), np.maximum(ea, eb)
    key, inv = np.unique(lo * nvg + hi, return_inverse=True)
    E = np.stack([key // nvg, key % nvg], 1)
    tri_edge = inv.reshape(3, -1).T  # [nt,3]
    # ---- crossings
    fl, fh = f[E[:, 0]], f[E[:, 1]]
    cut = (fl * fh) < 0
    ce = np.nonzero(cut)[0]
    a_in = np.where(fl[ce] < 0, E[ce, 0], E[ce, 1])  # inside endpoint
    b_out = np.where(fl[ce] < 0, E[ce, 1], E[ce, 0])
    t = _refine_crossings(phi, P[a_in], P[b_out], f[a_in], f[b_out], refine_its)
    Xc = P[a_in] + t[:, None] * (P[b_out] - P[a_in])
    elen = np.linalg.norm(P[b_out] - P[a_in], axis=1)
    # ---- snapping: each vertex moves to its nearest close crossing
    cand_v = np.concatenate([a_in[t < alpha], b_out[t > 1.0 - alpha]])
    cand_c = np.concatenate([np.nonzero(t < alpha)[0], np.nonzero(t > 1.0 - alpha)[0]])
    if len(cand_v):
        dist = np.concatenate([(t * elen)[t < alpha], ((1.0 - t) * elen)[t > 1.0 - alpha]])
        order = np.lexsort((cand_c, dist, cand_v))
        cv, cc = cand_v[order], cand_c[order]
        first = np.ones(len(cv), bool)
        first[1:] = cv[1:] != cv[:-1]
        P[cv[first]] = Xc[cc[first]]
        f[cv[first]] = 0.0
    # ---- remaining cut edges -> new vertices
    still = (f[a_in] * f[b_out]) < 0
    new_id = np.full(len(E), -1, dtype=np.int64)
    new_id[ce[still]] = nvg + np.arange(int(still.sum()))
    P = np.concatenate([P, Xc[still]])
    f = np.concatenate([f, np.zeros(int(still.sum()))])
    # ---- clip the triangles
    s = np.sign(f[T]).astype(np.int8)  # [nt,3]
    npl, nmi = (s > 0).sum(1), (s < 0).sum(1)
    out = []
    keep = (npl == 0) & (nmi >= 1)
    out.append(T[keep])
    allz = (npl == 0) & (nmi == 0)
    if allz.any():
        cen = P[T[allz]].mean(axis=1)
        out.append(T[allz][phi(cen) < 0])
    ar = np.arange(3)

    def rotated(mask, special):
        """Triangles of `mask` rotated so that corner `special[mask]` comes LAST (index 2); also the crossing vertex
        on the edge (1,2) [opposite the rotated corner 0] and on the edge (2,0) [opposite the rotated corner 1]."""
        idx = np.nonzero(mask)[0]
        r = (special[idx] + 1) % 3  # original index of the corner that becomes corner 0
        cols = (r[:, None] + ar[None, :]) % 3
        V = T[idx[:, None], cols]
        S = s[idx[:, None], cols]
        e12 = new_id[tri_edge[idx, cols[:, 0]]]
        e20 = new_id[tri_edge[idx, cols[:, 1]]]
        return V, S, e12, e20

    # one outside corner
    one = npl == 1
    if one.any():
        sp = np.argmax(s > 0, axis=1)
        V, S, e12, e20 = rotated(one, sp)
        a, b = V[:, 0], V[:, 1]
        both = (S[:, 0] < 0) & (S[:, 1] < 0)
        if both.any():
            A, B, C1, C2 = a[both], b[both], e12[both], e20[both]
            d1 = np.linalg.norm(P[A] - P[C1], axis=1)
            d2 = np.linalg.norm(P[B] - P[C2], axis=1)
            use1 = d1 <= d2
            out.append(np.stack([A, B, C1], 1)[use1])
            out.append(np.stack([A, C1, C2], 1)[use1])
            out.append(np.stack([A, B, C2], 1)[~use1])
            out.append(np.stack([B, C1, C2], 1)[~use1])
        m = (S[:, 0] < 0) & (S[:, 1] == 0)
        out.append(np.stack([a[m], b[m], e20[m]], 1))
        m = (S[:, 0] == 0) & (S[:, 1] < 0)
        out.append(np.stack([a[m], b[m], e12[m]], 1))
    # two outside corners, one inside
    two = (npl == 2) & (nmi == 1)
    if two.any():
        sp = np.argmax(s < 0, axis=1)  # the inside corner
        idx = np.nonzero(two)[0]
        r = sp[idx]
        cols = (r[:, None] + ar[None, :]) % 3  # inside corner first
        V = T[idx[:, None], cols]
        e01 = new_id[tri_edge[idx, cols[:, 2]]]  # edge (0,1) is opposite corner 2
        e20 = new_id[tri_edge[idx, cols[:, 1]]]
        out.append(np.stack([V[:, 0], e01, e20], 1))
    C = np.concatenate([o for o in out if len(o)])
    assert (C >= 0).all(), "clipping referenced an edge without a crossing vertex"
    # ---- drop degenerate triangles, compact the numbering
    p0, p1, p2 = P[C[:, 0]], P[C[:, 1]], P[C[:, 2]]
    area2 = (p1[:, 0] - p0[:, 0]) * (p2[:, 1] - p0[:, 1]) - (p1[:, 1] - p0[:, 1]) * (p2[:, 0] - p0[:, 0])
    C = C[area2 > 1e-9 * h * h]
    used = np.unique(C)
    remap = np.full(len(P), -1, dtype=np.int64)
    remap[used] = np.arange(len(used))
    return P[used], remap[C].astype(np.int32)


def keep_largest_component(x, cells):
    """Drops cells that are not edge-connected to the largest piece (slivers of the union that touch it in a point)."""
    import scipy.sparse as sp
    import scipy.sparse.csgraph as csg
    nc = len(cells)
    e = np.concatenate([cells[:, [1, 2]], cells[:, [2, 0]], cells[:, [0, 1]]])
    e.sort(axis=1)
    key = e[:, 0].astype(np.int64) * (len(x) + 1) + e[:, 1]
    order = np.argsort(key, kind="stable")
    ks = key[order]
    same = np.nonzero(ks[1:] == ks[:-1])[0]
    ca, cb = order[same] % nc, order[same + 1] % nc
    g = sp.coo_matrix((np.ones(len(ca)), (ca, cb)), shape=(nc, nc))
    ncomp, lab = csg.connected_components(g, directed=False)
    if ncomp == 1:
        return x, cells
    big = np.argmax(np.bincount(lab))
    cells = cells[lab == big]
    used = np.unique(cells)
    remap = np.full(len(x), -1, dtype=np.int64)
    remap[used] = np.arange(len(used))
    return x[used], remap[cells].astype(np.int32)


def mesh_quality(x, cells):
    """(smallest angle in degrees, smallest area / median area)."""
    p = x[cells]
    e = np.stack([p[:, 1] - p[:, 0], p[:, 2] - p[:, 1], p[:, 0] - p[:, 2]], 1)
    ln = np.linalg.norm(e, axis=2)
    cosang = np.stack([-(e[:, 0] * e[:, 2]).sum(1) / (ln[:, 0] * ln[:, 2]),
                       -(e[:, 1] * e[:, 0]).sum(1) / (ln[:, 1] * ln[:, 0]),
                       -(e[:, 2] * e[:, 1]).sum(1) / (ln[:, 2] * ln[:, 1])], 1)
    ang = np.degrees(np.arccos(np.clip(cosang, -1, 1)))
    area = 0.5 * np.abs(e[:, 0, 0] * e[:, 1, 1] - e[:, 0, 1] * e[:, 1, 0])
    return float(ang.min()), float(area.min() / np.median(area))
