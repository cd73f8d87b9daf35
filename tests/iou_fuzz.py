"""Independent float64 rotated-box IoU for LARGE batches (the pin of SURVEY §8a N1-N3 / VERDICT r3 item 8), vectorized numpy.

Third formulation in this repo, sharing nothing with the other two: `oracle/rotated_iou.c` (detectron2's scheme: edge-edge intersection
points + contained vertices -> Graham hull -> shoelace) and `tests/test_oracle_iou.py: sh_iou` (scalar Sutherland-Hodgman clip).  Here
the intersection area comes from Green's theorem on the boundary of A ∩ B, which consists of the parts of ∂A inside B and the parts of
∂B inside A: every edge P -> Q of one rectangle is clipped against the four half-planes of the other (Cyrus-Beck parameter interval
[t0, t1]) and contributes ½ · cross(P + t0 (Q − P), P + t1 (Q − P)); rectangles are oriented counter-clockwise.  Boundary that the two
rectangles SHARE is counted once: A's edges are kept where they lie inside or ON B, B's edges only where strictly inside A.

Vertex convention = Appendix A of SURVEY.md (detectron2's `get_rotated_vertices`): angle in degrees, w along the rotated x axis."""
import numpy as np


def corners(b):
    """[n, 5] (xc, yc, w, h, angle_deg) float64 -> [n, 4, 2], counter-clockwise in the (x, y) plane."""
    b = np.asarray(b, dtype=np.float64)
    t = np.deg2rad(b[:, 4])
    c, s = np.cos(t) * 0.5, np.sin(t) * 0.5
    x, y, w, h = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    p0 = np.stack([x + s * h + c * w, y + c * h - s * w], -1)
    p1 = np.stack([x - s * h + c * w, y - c * h - s * w], -1)
    ctr = np.stack([x, y], -1)
    pts = np.stack([p0, p1, 2 * ctr - p0, 2 * ctr - p1], 1)
    area2 = np.sum(pts[:, :, 0] * np.roll(pts[:, :, 1], -1, 1) - np.roll(pts[:, :, 0], -1, 1) * pts[:, :, 1], 1)
    flip = area2 < 0
    pts[flip] = pts[flip][:, ::-1]
    return pts


def _boundary_integral(A, B, strict):
    """sum over the edges of A of ½ cross(start, end) of the part of the edge inside convex CCW polygon B.  [n]"""
    n = A.shape[0]
    total = np.zeros(n)
    for i in range(4):
        P, Q = A[:, i], A[:, (i + 1) % 4]
        D = Q - P
        t0, t1 = np.zeros(n), np.ones(n)
        dead = np.zeros(n, bool)
        for j in range(4):
            E0, E1 = B[:, j], B[:, (j + 1) % 4]
            nx, ny = -(E1[:, 1] - E0[:, 1]), (E1[:, 0] - E0[:, 0])        # inward normal of a CCW edge
            f0 = nx * (P[:, 0] - E0[:, 0]) + ny * (P[:, 1] - E0[:, 1])    # signed distance (scaled) of P: >= 0 inside
            df = nx * D[:, 0] + ny * D[:, 1]
            scale = np.hypot(nx, ny) * np.maximum(np.hypot(D[:, 0], D[:, 1]), 1e-300)
            par = np.abs(df) <= 1e-14 * scale
            # parallel to this half-plane's boundary: wholly in or wholly out (ON the boundary: in for the non-strict side only)
            out_par = par & ((f0 <= 0) if strict else (f0 < 0))
            dead |= out_par
            with np.errstate(divide="ignore", invalid="ignore"):
                tc = np.where(par, 0.0, -f0 / np.where(par, 1.0, df))
            enter = (~par) & (df > 0)
            leave = (~par) & (df < 0)
            t0 = np.where(enter, np.maximum(t0, tc), t0)
            t1 = np.where(leave, np.minimum(t1, tc), t1)
        ok = (~dead) & (t1 > t0)
        S = P + t0[:, None] * D
        T = P + t1[:, None] * D
        total += np.where(ok, 0.5 * (S[:, 0] * T[:, 1] - T[:, 0] * S[:, 1]), 0.0)
    return total


def iou_fp64(b1, b2):
    """Element-wise IoU of rotated boxes b1[k] vs b2[k] in float64.  [n]"""
    b1, b2 = np.asarray(b1, np.float64), np.asarray(b2, np.float64)
    mid = 0.5 * (b1[:, :2] + b2[:, :2])                                  # translate to the pair's midpoint: keeps the cross products small
    a, b = b1.copy(), b2.copy()
    a[:, :2] -= mid
    b[:, :2] -= mid
    A, B = corners(a), corners(b)
    inter = _boundary_integral(A, B, strict=False) + _boundary_integral(B, A, strict=True)
    inter = np.maximum(inter, 0.0)
    a1, a2 = b1[:, 2] * b1[:, 3], b2[:, 2] * b2[:, 3]
    inter = np.minimum(inter, np.minimum(a1, a2))
    union = a1 + a2 - inter
    return np.where((a1 > 0) & (a2 > 0) & (union > 0), inter / np.where(union > 0, union, 1.0), 0.0)


# ---------------------------------------------------------------------------------------------------------------- fuzz families
def families(n, seed=0):
    """name -> (b1 [n, 5], b2 [n, 5]) float32: the families of VERDICT r3 item 8 — general overlap, shared edges, angle differences below
    1e-3 degrees, aspect ratios 1e-3 ... 1e3, coordinates up to 4096 * 16 (the class offset of lib/general.py:171-173 for 16 classes),
    near-identical boxes, concentric boxes with arbitrary angles, tiny boxes."""
    rs = np.random.RandomState(seed)

    def base(lo=8.0, hi=256.0, cmax=800.0):
        w = np.exp(rs.uniform(np.log(lo), np.log(hi), n))
        h = w * rs.uniform(1, 5, n)
        return np.stack([rs.uniform(0, cmax, n), rs.uniform(0, cmax, n), w, h, rs.uniform(-90, 90, n)], 1)

    out = {}
    a = base()
    b = base()
    b[:, :2] = a[:, :2] + rs.uniform(-1, 1, (n, 2)) * np.maximum(a[:, 2:4], b[:, 2:4])
    out["general"] = (a, b)
    a = base()
    b = a.copy()                                                          # same size and angle, shifted by exactly one width along its own axis
    t = np.deg2rad(a[:, 4])
    k = rs.randint(0, 3, n)                                               # 0: touching along w, 1: along h, 2: half-overlap along w
    sh = np.where(k == 1, a[:, 3], a[:, 2]) * np.where(k == 2, 0.5, 1.0)
    ax = np.where(k == 1, -np.sin(t), np.cos(t))
    ay = np.where(k == 1, np.cos(t), np.sin(t))
    b[:, 0] += sh * ax
    b[:, 1] += sh * ay
    out["shared_edges"] = (a, b)
    a = base()
    b = a.copy()
    b[:, 4] += rs.uniform(-1e-3, 1e-3, n)
    b[:, :2] += rs.uniform(-0.5, 0.5, (n, 2)) * a[:, 2:4]
    b[:, 2:4] *= rs.uniform(0.8, 1.25, (n, 2))
    out["tiny_angle_difference"] = (a, b)
    a = base()
    asp = np.exp(rs.uniform(np.log(1e-3), np.log(1e3), n))
    s = np.exp(rs.uniform(np.log(4), np.log(64), n))
    a[:, 2], a[:, 3] = s * np.sqrt(asp), s / np.sqrt(asp)
    b = a.copy()
    b[:, 4] += rs.uniform(-30, 30, n)
    b[:, :2] += rs.uniform(-0.3, 0.3, (n, 2)) * np.minimum(a[:, 2:3], a[:, 3:4]) * 4
    b[:, 2:4] *= rs.uniform(0.5, 2.0, (n, 2))
    out["extreme_aspect"] = (a, b)
    a = base()
    b = base()
    b[:, :2] = a[:, :2] + rs.uniform(-1, 1, (n, 2)) * np.maximum(a[:, 2:4], b[:, 2:4]) * 0.7
    off = rs.randint(0, 17, n)[:, None] * 4096.0
    a[:, :2] += off
    b[:, :2] += off
    out["class_offset_coordinates"] = (a, b)
    a = base()
    b = a * (1 + rs.uniform(-2e-6, 2e-6, a.shape))
    out["near_identical"] = (a, b)
    a = base()
    b = base()
    b[:, :2] = a[:, :2]
    out["concentric"] = (a, b)
    a = base(lo=0.05, hi=2.0)
    b = base(lo=0.05, hi=2.0)
    b[:, :2] = a[:, :2] + rs.uniform(-1, 1, (n, 2)) * np.maximum(a[:, 2:4], b[:, 2:4]) * 0.7
    out["tiny_boxes"] = (a, b)
    return {k: (v[0].astype(np.float32), v[1].astype(np.float32)) for k, v in out.items()}


def histogram(dev):
    """absolute deviations -> counts per decade + the tail statistics written to profiles/."""
    dev = np.asarray(dev, np.float64)
    edges = [0.0, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1, np.inf]
    counts, _ = np.histogram(dev, bins=edges)
    return {"n": int(dev.size), "max": float(dev.max()), "p999": float(np.quantile(dev, 0.999)), "p99": float(np.quantile(dev, 0.99)),
            "mean": float(dev.mean()), "bins": {f"[{edges[i]:g}, {edges[i + 1]:g})": int(c) for i, c in enumerate(counts)}}
