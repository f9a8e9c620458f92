"""Level-set descriptions of the planar shapes the tree-shaped BASELINE domain is the union of
(/root/reference/src/scenarios/stenosis_with_tree.py:256-420): the stenosed channel with cubic-Bezier walls, the
coupling trapezoid, and one offset polygon per tree branch (Bezier centreline, constant half-width).
Every shape gives phi(points) < 0 inside, a bounding box, and exact zero sets; magnitudes are true distances for
polygons and within ~12 % of the distance for the channel (enough for the mesher's crossing search)."""
from __future__ import annotations

import numpy as np


def bezier_point(P, t):
    """Cubic Bezier with control points P [4,2] at parameters t [n] -> [n,2]."""
    t = np.asarray(t, dtype=np.float64)[:, None]
    mt = 1.0 - t
    return mt**3 * P[0] + 3 * mt**2 * t * P[1] + 3 * mt * t**2 * P[2] + t**3 * P[3]


def bezier_y_of_x(P, x):
    """y on a cubic Bezier whose x(t) is monotone increasing, at abscissae x (Newton + bisection safeguard)."""
    x = np.asarray(x, dtype=np.float64)
    px, py = P[:, 0], P[:, 1]
    lo, hi = np.zeros_like(x), np.ones_like(x)
    t = np.clip((x - px[0]) / (px[3] - px[0]), 0.0, 1.0)
    for _ in range(40):
        mt = 1.0 - t
        fx = mt**3 * px[0] + 3 * mt**2 * t * px[1] + 3 * mt * t**2 * px[2] + t**3 * px[3] - x
        dfx = 3 * mt**2 * (px[1] - px[0]) + 6 * mt * t * (px[2] - px[1]) + 3 * t**2 * (px[3] - px[2])
        hi = np.where(fx > 0, t, hi)
        lo = np.where(fx <= 0, t, lo)
        tn = t - fx / np.where(dfx != 0, dfx, 1.0)
        t = np.where((tn > lo) & (tn < hi) & (dfx != 0), tn, 0.5 * (lo + hi))
    mt = 1.0 - t
    return mt**3 * py[0] + 3 * mt**2 * t * py[1] + 3 * mt * t**2 * py[2] + t**3 * py[3]


class Polygon:
    """Simple polygon (vertices counter-clockwise or clockwise): exact signed distance, negative inside."""

    def __init__(self, pts):
        self.p = np.asarray(pts, dtype=np.float64)
        self.bbox = (self.p[:, 0].min(), self.p[:, 1].min(), self.p[:, 0].max(), self.p[:, 1].max())

    def phi(self, q):
        q = np.asarray(q, dtype=np.float64)
        a = self.p
        b = np.roll(self.p, -1, axis=0)
        d2 = np.full(len(q), np.inf)
        inside = np.zeros(len(q), dtype=bool)
        for k in range(len(a)):
            e = b[k] - a[k]
            w = q - a[k]
            tt = np.clip((w @ e) / (e @ e), 0.0, 1.0)
            r = w - tt[:, None] * e
            d2 = np.minimum(d2, (r * r).sum(1))
            # crossing number of the horizontal ray to +x
            c1 = (a[k, 1] <= q[:, 1]) != (b[k, 1] <= q[:, 1])
            with np.errstate(divide="ignore", invalid="ignore"):
                xint = a[k, 0] + (q[:, 1] - a[k, 1]) * e[0] / e[1] if e[1] != 0 else np.full(len(q), np.inf)
            inside ^= c1 & (q[:, 0] < xint)
        d = np.sqrt(d2)
        return np.where(inside, -d, d)


class StenosedChannel:
    """0 <= x <= L, |y - yc| <= R(x): linear taper R_in -> R_out with a symmetric narrowing at x_sten whose wall is
    two cubic Beziers (stenosis.py:281-374 / stenosis_with_tree.py:256-310): junctions at x_sten -/+ dist_x on the
    taper line, throat radius (1 - severity) * R_taper(x_sten), handles of length tension * dist_x along the taper
    slope (C1 at junctions and throat)."""

    def __init__(self, L, R_in, R_out, x_sten, severity, slope, tension=0.5, yc=None, clamp_frac=0.05):
        self.L, self.R_in, self.R_out = float(L), float(R_in), float(R_out)
        self.yc = float(R_in if yc is None else yc)
        self.x_sten = float(x_sten)
        r_mid = R_in + (R_out - R_in) * x_sten / L
        self.R_min = (1.0 - severity) * r_mid
        if not self.R_min > 0:
            raise ValueError("stenosis severity must be strictly < 1")
        h_sten = r_mid - self.R_min
        d = h_sten / slope if slope > 0 else L / 4.0
        if clamp_frac:
            d = max(d, L * clamp_frac)  # stenosis_with_tree.py:265 (stenosis.py does not apply this lower bound)
        d = min(d, min(x_sten, L - x_sten) * 0.95)
        self.dist_x = d
        s = (R_out - R_in) / L
        x1, x2 = x_sten - d, x_sten + d
        r1, r2 = R_in + s * x1, R_in + s * x2
        ha = tension * d
        self.B1 = np.array([[x1, r1], [x1 + ha, r1 + ha * s], [x_sten - ha, self.R_min - ha * s], [x_sten, self.R_min]])
        self.B2 = np.array([[x_sten, self.R_min], [x_sten + ha, self.R_min + ha * s], [x2 - ha, r2 - ha * s], [x2, r2]])
        self.x1, self.x2, self.s = x1, x2, s
        self.bbox = (0.0, self.yc - max(R_in, R_out), self.L, self.yc + max(R_in, R_out))

    def radius(self, x):
        x = np.asarray(x, dtype=np.float64)
        R = self.R_in + self.s * x
        m1 = (x > self.x1) & (x <= self.x_sten)
        m2 = (x > self.x_sten) & (x < self.x2)
        if m1.any():
            R[m1] = bezier_y_of_x(self.B1, x[m1])
        if m2.any():
            R[m2] = bezier_y_of_x(self.B2, x[m2])
        return R

    def phi(self, q):
        q = np.asarray(q, dtype=np.float64)
        x = np.clip(q[:, 0], 0.0, self.L)
        wall = np.abs(q[:, 1] - self.yc) - self.radius(x)
        return np.maximum(wall, np.maximum(-q[:, 0], q[:, 0] - self.L))


def branch_polygon(A, B, tang_in, r, n_samples=12, handle=0.4):
    """Offset polygon of half-width r around the cubic Bezier centreline from A to B that leaves A along `tang_in`
    and arrives at B along the chord direction (stenosis_with_tree.py:379-403).  Returns (polygon [2(n+1),2],
    end cap (two points at B))."""
    A, B, tang_in = (np.asarray(v, dtype=np.float64) for v in (A, B, tang_in))
    seg = B - A
    ln = np.linalg.norm(seg)
    tang_out = seg / ln
    hh = ln * handle
    P = np.array([A, A + hh * tang_in, B - hh * tang_out, B])
    t = np.arange(n_samples + 1) / n_samples
    mt = 1.0 - t
    pt = bezier_point(P, t)
    tan = (3 * mt**2)[:, None] * (P[1] - P[0]) + (6 * mt * t)[:, None] * (P[2] - P[1]) + (3 * t**2)[:, None] * (P[3] - P[2])
    tan /= np.linalg.norm(tan, axis=1)[:, None]
    perp = np.stack([-tan[:, 1], tan[:, 0]], 1)
    top, bot = pt + r * perp, pt - r * perp
    return np.concatenate([top, bot[::-1]]), (top[-1], bot[-1])


class Union:
    """phi = min over the parts; each part is evaluated only inside its (padded) bounding box."""

    def __init__(self, parts, pad):
        self.parts, self.pad = list(parts), float(pad)
        bb = np.array([p.bbox for p in self.parts])
        self.bbox = (bb[:, 0].min(), bb[:, 1].min(), bb[:, 2].max(), bb[:, 3].max())

    def phi(self, q):
        q = np.asarray(q, dtype=np.float64)
        out = np.full(len(q), np.inf)
        for p in self.parts:
            x0, y0, x1, y1 = p.bbox
            m = (q[:, 0] >= x0 - self.pad) & (q[:, 0] <= x1 + self.pad) & (q[:, 1] >= y0 - self.pad) & (q[:, 1] <= y1 + self.pad)
            if m.any():
                out[m] = np.minimum(out[m], p.phi(q[m]))
        return np.where(np.isfinite(out), out, self.pad)

    def fill(self, xs, ys):
        """phi on the tensor grid xs x ys -> [len(ys), len(xs)]; untouched points get +pad (outside)."""
        F = np.full((len(ys), len(xs)), self.pad)
        for p in self.parts:
            x0, y0, x1, y1 = p.bbox
            i0, i1 = np.searchsorted(xs, x0 - self.pad), np.searchsorted(xs, x1 + self.pad, side="right")
            j0, j1 = np.searchsorted(ys, y0 - self.pad), np.searchsorted(ys, y1 + self.pad, side="right")
            if i1 <= i0 or j1 <= j0:
                continue
            X, Y = np.meshgrid(xs[i0:i1], ys[j0:j1], indexing="xy")
            v = p.phi(np.stack([X.ravel(), Y.ravel()], 1)).reshape(j1 - j0, i1 - i0)
            F[j0:j1, i0:i1] = np.minimum(F[j0:j1, i0:i1], v)
        return F
