"""Seeded synthetic inputs shaped like the reference's datasets.

Bench/test data only — there is no network for SemanticKITTI/Toronto3D/KITTI.  Pure numpy.
"""
import numpy as np


def _grid_barycentre(points, cell):
    """Voxel-grid barycentre subsample (numpy stand-in for the 0.06 m preprocess step,
    ml3d/torch/models/randlanet.py:133-139).  Order = ascending voxel key."""
    org = np.floor(points.min(0) / cell) * cell
    c = np.floor((points - org) / cell).astype(np.int64)
    dims = c.max(0) + 1
    key = c[:, 0] + dims[0] * (c[:, 1] + dims[1] * c[:, 2])
    uniq, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)
    out = np.zeros((uniq.size, 3), np.float64)
    np.add.at(out, inv, points.astype(np.float64))
    return (out / cnt[:, None]).astype(np.float32)


def lidar_sweep(seed, n_beams=64, n_azimuth=2048, with_intensity=False):
    """One spinning-lidar sweep: rays hit the nearest of a ground plane (z = -1.73 m) and 40
    axis-aligned boxes; range <= 80 m; N(0, 0.02) m range noise.  ~100-130 k points."""
    rng = np.random.default_rng(seed)
    elev = np.deg2rad(np.linspace(-24.8, 2.0, n_beams))
    azim = np.linspace(-np.pi, np.pi, n_azimuth, endpoint=False)
    el, az = np.meshgrid(elev, azim, indexing="ij")
    d = np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], -1).reshape(-1, 3)
    t_best = np.full(d.shape[0], np.inf)
    # ground plane
    with np.errstate(divide="ignore", invalid="ignore"):
        tg = np.where(d[:, 2] < -1e-6, -1.73 / d[:, 2], np.inf)
    t_best = np.minimum(t_best, tg)
    # boxes (slab test)
    n_box = 40
    ctr = np.concatenate([rng.uniform(-40, 40, (n_box, 2)), np.full((n_box, 1), -1.73)], 1)
    size = np.concatenate([rng.uniform(1, 10, (n_box, 2)), rng.uniform(1, 6, (n_box, 1))], 1)
    lo = ctr - size * np.array([0.5, 0.5, 0.0])
    hi = ctr + size * np.array([0.5, 0.5, 1.0])
    keep = np.linalg.norm(ctr[:, :2], axis=1) > 4.0  # nothing on top of the sensor
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / d
        for b in np.nonzero(keep)[0]:
            t0 = lo[b] * inv
            t1 = hi[b] * inv
            tmin = np.minimum(t0, t1).max(1)
            tmax = np.maximum(t0, t1).min(1)
            hit = (tmax >= tmin) & (tmax > 0) & (tmin > 0)
            t_best = np.where(hit & (tmin < t_best), tmin, t_best)
    ok = np.isfinite(t_best) & (t_best < 80.0)
    t = t_best[ok] + rng.normal(0, 0.02, ok.sum())
    pts = (d[ok] * t[:, None]).astype(np.float32)
    if with_intensity:
        return np.concatenate([pts, rng.uniform(0, 1, (pts.shape[0], 1)).astype(np.float32)], 1)
    return pts


def semantickitti_patch(frame_id, num_points=45056):
    """What ``RandLANet.transform`` sees after the sampler: [num_points, 3] f32, the num_points
    nearest (to a random centre) points of a 0.06 m grid-subsampled sweep, recentred in x/y
    (randlanet_semantickitti.yml augment.recenter dim [0, 1]) and shuffled.  seed = 1000 + frame_id."""
    seed = 1000 + int(frame_id)
    rng = np.random.default_rng(seed)
    sweep = lidar_sweep(seed)
    sub = _grid_barycentre(sweep, 0.06)
    if sub.shape[0] < num_points:  # extremely sparse seed: densify by jittered duplication
        extra = sub[rng.integers(0, sub.shape[0], num_points - sub.shape[0])]
        sub = np.concatenate([sub, extra + rng.normal(0, 0.03, extra.shape).astype(np.float32)], 0)
    centre = sub[rng.integers(0, sub.shape[0])]
    d2 = ((sub - centre) ** 2).sum(1)
    sel = np.argpartition(d2, num_points - 1)[:num_points]
    sel = sel[rng.permutation(num_points)]
    pc = sub[sel].copy()
    pc[:, :2] -= pc[:, :2].mean(0)
    return np.ascontiguousarray(pc, np.float32)


def semantickitti_batch(first_frame, batch, num_points=45056):
    return np.stack([semantickitti_patch(first_frame + i, num_points) for i in range(batch)])


def uniform_cloud(seed, n, extent=(10.0, 10.0, 2.0)):
    rng = np.random.default_rng(seed)
    return (rng.random((n, 3), dtype=np.float32) * np.asarray(extent, np.float32)).astype(np.float32)


def kitti_sweep(frame_id):
    """PointPillars / KITTI-shaped raw sweep: [~120k, 4] f32 = xyz + intensity U(0,1) (SURVEY.md §8d
    frame 3).  seed = 3000 + frame_id."""
    return lidar_sweep(3000 + int(frame_id), with_intensity=True)


def _urban_scene(rng, R):
    """Raw mobile-laser-scan scene on the square [-R, R]^2: ground plane, two facades, poles, car-sized boxes
    (~1000 pts/m^2), float32 [n, 3] with N(0, 5 mm) noise.  Draw order is part of the golden files' seeds."""

    def plane(n, origin, u, v):
        a = rng.random((n, 2))
        return origin + a[:, :1] * u + a[:, 1:] * v

    parts = [plane(int(1000 * (2 * R) ** 2 * 0.6), np.array([-R, -R, 0.0]), np.array([2 * R, 0, 0]), np.array([0, 2 * R, 0]))]
    for y in (-2.8, 3.1):                                            # facades
        parts.append(plane(int(600 * 2 * R * 5), np.array([-R, y, 0.0]), np.array([2 * R, 0, 0]), np.array([0, 0, 5.0])))
    for _ in range(6):                                               # poles
        c = rng.uniform(-R, R, 2)
        th, z = rng.uniform(0, 2 * np.pi, 1500), rng.uniform(0, 4.5, 1500)
        parts.append(np.stack([c[0] + 0.12 * np.cos(th), c[1] + 0.12 * np.sin(th), z], 1))
    for _ in range(3):                                               # cars
        c = np.append(rng.uniform(-R + 1, R - 1, 2), 0.0)
        s = np.array([rng.uniform(3.5, 4.5), rng.uniform(1.6, 1.9), rng.uniform(1.3, 1.6)])
        parts.append(plane(6000, c + [0, 0, s[2]], [s[0], 0, 0], [0, s[1], 0]))
        parts.append(plane(5000, c, [s[0], 0, 0], [0, 0, s[2]]))
        parts.append(plane(5000, c + [0, s[1], 0], [s[0], 0, 0], [0, 0, s[2]]))
    pts = np.concatenate(parts).astype(np.float32)
    pts += rng.normal(0, 0.005, pts.shape).astype(np.float32)
    return pts


def toronto3d_tile(seed, half=6.0, density=0.25):
    """A raw Toronto3D-shaped cloud for the segmentation PIPELINES (``run_inference`` of a whole cloud, not one sphere):
    the urban scene on a ``2 half`` x ``2 half`` m tile, thinned to ``density`` of its ~1000 pts/m^2, with RGB features
    in [0, 255] and labels 1..8 by height band.  Returns dict(point [n,3] f32, feat [n,3] f32, label [n] int32)."""
    rng = np.random.default_rng(2500 + int(seed))
    pts = _urban_scene(rng, half)
    keep = rng.random(pts.shape[0]) < density
    pts = np.ascontiguousarray(pts[keep], np.float32)
    feat = rng.uniform(0, 255, (pts.shape[0], 3)).astype(np.float32)
    label = (1 + np.clip((pts[:, 2] * 1.6).astype(np.int32), 0, 7)).astype(np.int32)
    return dict(point=pts, feat=feat, label=label)


def toronto3d_sphere(frame_id, max_points=10000, radius=4.0, grid=0.08):
    """KPConv / Toronto3D-shaped input sphere (SURVEY.md §8d frame 2): an urban mobile-laser-scan scene
    (ground plane, two facades, poles, car-sized boxes; ~1000 pts/m^2 before subsampling) grid-subsampled
    at ``grid`` and cropped to the ``max_points`` points nearest to the sphere centre within ``radius``.
    Returns [n, 3] f32 centred on the sphere centre.  seed = 2000 + frame_id."""
    rng = np.random.default_rng(2000 + int(frame_id))
    pts = _urban_scene(rng, radius * 1.05)
    sub = _grid_barycentre(pts, grid)
    centre = np.array([rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5), 1.0], np.float32)
    d2 = ((sub - centre) ** 2).sum(1)
    sel = np.nonzero(d2 < radius * radius)[0]
    if sel.size > max_points:
        sel = sel[np.argsort(d2[sel], kind="stable")[:max_points]]
    sel = sel[rng.permutation(sel.size)]
    return np.ascontiguousarray(sub[sel] - centre, np.float32)
