"""Deterministic synthetic clouds for the benchmark / parity workloads:
a jittered-grid heightfield tile with a steep ridge, a source epoch with displaced discs
(the "unstable" areas), random drop-out and a small rigid motion (the coarse-alignment
residual the reference presumes, README.md:49-50 of the reference)."""
import numpy as np

SEED0 = 20250906


def _height(x, y, L):
    ridge = 0.3 * np.clip((x - 0.6 * L) / (0.04 * L + 1e-12), 0.0, 1.0)
    return 0.15 * np.sin(1.3 * x) * np.cos(0.9 * y) + 0.05 * np.sin(5.1 * x + 1.0) * np.sin(4.3 * y) + ridge


def _normal(x, y, L, h=1e-4):
    dzdx = (_height(x + h, y, L) - _height(x - h, y, L)) / (2 * h)
    dzdy = (_height(x, y + h, L) - _height(x, y - h, L)) / (2 * h)
    n = np.stack([-dzdx, -dzdy, np.ones_like(dzdx)], axis=1)
    return n / np.linalg.norm(n, axis=1, keepdims=True)


def euler_matrix(ax, ay, az, t):
    """R = Rz(az) Ry(ay) Rx(ax) (the convention of the reference's matrix2angle)."""
    ca, sa, cb, sb, cg, sg = np.cos(ax), np.sin(ax), np.cos(ay), np.sin(ay), np.cos(az), np.sin(az)
    R = np.array([[cg * cb, -sg * ca + cg * sb * sa, sg * sa + cg * sb * ca],
                  [sg * cb, cg * ca + sg * sb * sa, -cg * sa + sg * sb * ca],
                  [-sb, cb * sa, cb * ca]])
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = t
    return T


def make_tile(n_points, r=0.005, epoch=0, offset=(0.0, 0.0, 0.0)):
    """Reference-epoch tile with ~n_points points at mean spacing r. Returns float32 (n,3)."""
    rng = np.random.default_rng(SEED0 + epoch)
    side = int(round(np.sqrt(n_points)))
    L = side * r
    gx, gy = np.meshgrid(np.arange(side), np.arange(side), indexing="ij")
    x = (gx.ravel() + 0.5) * r + rng.uniform(-0.3 * r, 0.3 * r, side * side)
    y = (gy.ravel() + 0.5) * r + rng.uniform(-0.3 * r, 0.3 * r, side * side)
    z = _height(x, y, L) + rng.normal(0.0, 0.2 * r, side * side)
    pts = np.stack([x, y, z], axis=1) + np.asarray(offset)[None, :]
    return pts.astype(np.float32), L


def make_source(n_points, r=0.005, epoch=1, offset=(0.0, 0.0, 0.0), max_angle_deg=0.2, max_trans_r=2.0):
    """Source epoch: fresh sampling of the same surface, 15 % of the area displaced 4r..40r along
    the surface normal, 5 % of the points dropped, rigid motion about the centroid.
    Returns (points float32 (n,3), T_gt 4x4 double mapping SOURCE -> reference frame)."""
    rng = np.random.default_rng(SEED0 + epoch)
    side = int(round(np.sqrt(n_points)))
    L = side * r
    gx, gy = np.meshgrid(np.arange(side), np.arange(side), indexing="ij")
    x = (gx.ravel() + 0.5) * r + rng.uniform(-0.3 * r, 0.3 * r, side * side)
    y = (gy.ravel() + 0.5) * r + rng.uniform(-0.3 * r, 0.3 * r, side * side)
    z = _height(x, y, L) + rng.normal(0.0, 0.2 * r, side * side)
    pts = np.stack([x, y, z], axis=1)
    # displaced discs: 3 discs totalling 15 % of the area
    nd = 3
    rad = np.sqrt(0.15 * L * L / (nd * np.pi))
    nrm = _normal(x, y, L)
    for _ in range(nd):
        cx, cy = rng.uniform(rad, L - rad, 2)
        amp = rng.uniform(4 * r, 40 * r) * rng.choice([-1.0, 1.0])
        inside = (x - cx) ** 2 + (y - cy) ** 2 < rad * rad
        pts[inside] += amp * nrm[inside]
    keep = rng.uniform(size=len(pts)) >= 0.05
    pts = pts[keep]
    ang = np.deg2rad(rng.uniform(-max_angle_deg, max_angle_deg, 3))
    t = rng.uniform(-max_trans_r * r, max_trans_r * r, 3)
    c = pts.mean(axis=0)
    M = euler_matrix(ang[0], ang[1], ang[2], t)          # motion applied to the source about c
    moved = (pts - c) @ M[:3, :3].T + c + M[:3, 3]
    # T_gt maps the moved source back: p = R^T (q - c - t) + c
    Tgt = np.eye(4)
    Tgt[:3, :3] = M[:3, :3].T
    Tgt[:3, 3] = c - M[:3, :3].T @ (c + M[:3, 3])
    off = np.asarray(offset, dtype=np.float64)
    # account for the global offset: q' = q + off, p' = p + off
    Tg = Tgt.copy()
    Tg[:3, 3] = Tgt[:3, 3] + off - Tgt[:3, :3] @ off
    return (moved + off[None, :]).astype(np.float32), Tg


def grid_labels(cloud, cell):
    """Cheap stand-in segmentation (square xy cells of edge `cell`), for kernel tests that do not
    need the real supervoxel front end. Returns (labels int32, n_labels)."""
    xy = np.floor((cloud[:, :2] - cloud[:, :2].min(axis=0)) / cell).astype(np.int64)
    key = xy[:, 0] * (xy[:, 1].max() + 1) + xy[:, 1]
    _, lab = np.unique(key, return_inverse=True)
    return lab.astype(np.int32), int(lab.max()) + 1
