"""Seeded synthetic scenes for the scan-to-map ICP path.

World  : axis-aligned room-and-corridor box model (floor z=-1.5, ceiling z=+6, wall lines every
         `spacing` metres with door gaps, a few 45-degree slabs so that all 6 DoF are observable),
         spanning 3x3x1 map cubes (150 m x 150 m) including negative coordinates; no plane passes
         through the world origin (the reference's A x = -1 plane form is singular there).
Map    : surfaces sampled with N(0, sigma) noise along the normal (mandatory: LidarSlam.cpp:772 rejects
         noise-free planes), reduced to one centroid per `plane_res` voxel, trimmed to exactly M points.
Scan   : ray-cast from a ground-truth pose, rings x azimuth beams, range noise N(0, sigma), max range
         100 m, misses re-sampled so that exactly rings*azimuth points come back, sensor frame, fp32.
Guess  : ground truth (+) U(-0.1,0.1) m per axis, U(-1,1) deg per axis.
Seeds  : world=1, map noise=2, scan noise=3+i, guess=1000+i for scan i.

Everything is numpy; nothing here touches the GPU library or the oracle."""
import os

import numpy as np

CUBE = 50.0

CONFIGS = {
    # BASELINE.json configs[1]: 16-ring x 1800 scans vs 200k-pt map
    "vlp16_200k": dict(rings=16, azimuth=1800, fov_deg=15.0, map_points=200_000, extent=40.0, spacing=10.0, plane_res=0.2),
    # BASELINE.json configs[2]: OS1-128 (131 072 pts/scan) vs 2M-pt local map -- the headline workload
    "os1_128_2m": dict(rings=128, azimuth=1024, fov_deg=22.5, map_points=2_000_000, extent=75.0, spacing=9.0, plane_res=0.2),
    # small cases the CPU oracle finishes in seconds
    "tiny": dict(rings=16, azimuth=256, fov_deg=15.0, map_points=30_000, extent=14.0, spacing=7.0, plane_res=0.2),
    "small": dict(rings=32, azimuth=512, fov_deg=22.5, map_points=120_000, extent=30.0, spacing=10.0, plane_res=0.2),
    # stand-in for config/livox_mid360.yaml (planeRes 0.1, max_surface_features 4000): a 20 000-point sweep on a ring x azimuth grid
    # (the Mid-360's non-repetitive pattern is not modelled) in a 60 m x 60 m part of the same world, map at planeRes 0.1
    "mid360_like": dict(rings=40, azimuth=500, fov_deg=26.0, map_points=400_000, extent=30.0, spacing=10.0, plane_res=0.1),
    # second perf scene: an OPEN hall -- floor, ceiling, outer walls and 1 m high interior walls that occlude almost nothing, so
    # the sweep reaches out to its 100 m range and touches nearly every occupied 50 m cube (M_t ~ the whole 2M-point map)
    "open_2m": dict(rings=128, azimuth=1024, fov_deg=22.5, map_points=2_000_000, extent=95.0, spacing=9.0, plane_res=0.2, wall_height=1.0),
}


# --------------------------------------------------------------------------------------------
# world
# --------------------------------------------------------------------------------------------
class World:
    """A set of finite rectangles o + a*u + b*v, a,b in [0,1]."""

    def __init__(self, extent=75.0, spacing=12.5, seed=1, z0=-1.5, z1=6.1, wall_height=None):
        """wall_height: height of the INTERIOR walls above the floor (None = floor to ceiling, with lintels above the doors)."""
        rng = np.random.default_rng(seed)
        E = float(extent)
        R = []

        def add(o, u, v):
            R.append((np.array(o, float), np.array(u, float), np.array(v, float)))

        add([-E, -E, z0], [2 * E, 0, 0], [0, 2 * E, 0])  # floor
        add([-E, -E, z1], [2 * E, 0, 0], [0, 2 * E, 0])  # ceiling
        H = z1 - z0
        for s in (-E, E):  # outer walls
            add([s, -E, z0], [0, 2 * E, 0], [0, 0, H])
            add([-E, s, z0], [2 * E, 0, 0], [0, 0, H])
        if wall_height is not None:
            H = float(wall_height)
        off = 3.7  # keeps every wall plane away from the world origin
        lines = [off + k * spacing for k in range(-int(2 * E / spacing) - 1, int(2 * E / spacing) + 2)]
        lines = [c for c in lines if -E + 1.0 < c < E - 1.0]
        door = 2.5
        for c in lines:  # interior walls along y (x = c) and along x (y = c), one door gap per segment
            for s0 in np.arange(-E, E, spacing):
                s1 = min(s0 + spacing, E)
                if s1 - s0 < door + 1.0:
                    add([c, s0, z0], [0, s1 - s0, 0], [0, 0, H]); add([s0, c, z0], [s1 - s0, 0, 0], [0, 0, H])
                    continue
                for axis in (0, 1):
                    g = s0 + 0.5 + rng.random() * (s1 - s0 - door - 1.0)
                    for a, b in ((s0, g), (g + door, s1)):
                        if b - a < 0.05:
                            continue
                        if axis == 0:
                            add([c, a, z0], [0, b - a, 0], [0, 0, H])
                        else:
                            add([a, c, z0], [b - a, 0, 0], [0, 0, H])
                    if H <= 2.2:
                        continue
                    # lintel above the door
                    if axis == 0:
                        add([c, g, z0 + 2.2], [0, door, 0], [0, 0, H - 2.2])
                    else:
                        add([g, c, z0 + 2.2], [door, 0, 0], [0, 0, H - 2.2])
        # 45-degree slabs (ramps / leaning panels) near the trajectory and scattered around
        n_slabs = max(6, int((2 * E / spacing) ** 2 / 3))
        for _ in range(n_slabs):
            cx, cy = (rng.random(2) * 2 - 1) * (E - 6.0)
            L, W = 2.0 + 2.5 * rng.random(), 2.0 + 2.0 * rng.random()
            if rng.random() < 0.5:
                add([cx, cy, z0], [L, 0, L], [0, W, 0])
            else:
                add([cx, cy, z0], [0, L, L], [W, 0, 0])
        for o, u, v in (([-6.0, -7.5, z0], [3.0, 0, 3.0], [0, 3.0, 0]), ([1.0, -9.0, z0], [0, 2.5, 2.5], [2.5, 0, 0]),
                        ([-2.5, 1.5, z0 + 1.0], [2.0, 2.0, 0], [0, 0, 2.5])):
            add(o, u, v)
        self.o = np.stack([r[0] for r in R]); self.u = np.stack([r[1] for r in R]); self.v = np.stack([r[2] for r in R])
        n = np.cross(self.u, self.v)
        self.n = n / np.linalg.norm(n, axis=1, keepdims=True)
        self.extent = E

    def area(self):
        return float(np.sum(np.linalg.norm(np.cross(self.u, self.v), axis=1)))

    def plane_groups(self):
        """[(unit normal, offset n.o, member rectangle indices)] -- rectangles grouped by supporting plane."""
        if getattr(self, "_groups", None) is None:
            sign = np.where((self.n @ np.array([1.0, 1e-3, 1e-6])) < 0, -1.0, 1.0)  # canonical orientation
            nn = self.n * sign[:, None]
            off = np.sum(nn * self.o, 1)
            key = np.round(np.c_[nn, off], 6)
            _, inv = np.unique(key, axis=0, return_inverse=True)
            inv = inv.ravel()
            self._groups = [(nn[np.nonzero(inv == g)[0][0]], off[np.nonzero(inv == g)[0][0]], np.nonzero(inv == g)[0])
                            for g in range(inv.max() + 1)]
        return self._groups


def _voxel_keys(p32, inv_leaf32):
    """(cube, voxel) key with the float arithmetic of pcl::VoxelGrid (floor(x * inv_leaf) in float)."""
    ijk = np.floor(p32 * inv_leaf32).astype(np.int64) + (1 << 19)
    return (ijk[:, 0] << 42) | (ijk[:, 1] << 21) | ijk[:, 2]


def sample_map(world, plane_res=0.2, target_points=None, seed=2, sigma=0.01, ds=None):
    """One noisy centroid per plane_res voxel; exactly target_points points when given."""
    rng = np.random.default_rng(seed)
    ds = ds or plane_res * 0.5
    inv = np.float32(1.0) / np.float32(plane_res)
    chunks = []
    for o, u, v, n in zip(world.o, world.u, world.v, world.n):
        lu, lv = np.linalg.norm(u), np.linalg.norm(v)
        nu, nv = max(1, int(np.ceil(lu / ds))), max(1, int(np.ceil(lv / ds)))
        a = (np.arange(nu)[:, None] + rng.random((nu, nv))) / nu
        b = (np.arange(nv)[None, :] + rng.random((nu, nv))) / nv
        p = o + a[..., None] * u + b[..., None] * v + (rng.standard_normal((nu, nv)) * sigma)[..., None] * n
        chunks.append(p.reshape(-1, 3))
    P = np.concatenate(chunks).astype(np.float32)
    for _ in range(4):  # centroid per voxel until every voxel holds exactly one point
        k = _voxel_keys(P, inv)
        uk, invk, cnt = np.unique(k, return_inverse=True, return_counts=True)
        if len(uk) == len(P):
            break
        C = np.zeros((len(uk), 3), np.float64)
        for a in range(3):
            C[:, a] = np.bincount(invk, weights=P[:, a].astype(np.float64), minlength=len(uk))
        P = (C / cnt[:, None]).astype(np.float32)
    k = _voxel_keys(P, inv)
    _, first = np.unique(k, return_index=True)
    P = P[np.sort(first)]
    if target_points is not None:
        if len(P) < target_points:
            raise ValueError(f"world too small: {len(P)} voxels < {target_points}; raise extent or lower spacing")
        keep = np.sort(rng.permutation(len(P))[:target_points])
        P = P[keep]
    return np.ascontiguousarray(P)


# --------------------------------------------------------------------------------------------
# poses
# --------------------------------------------------------------------------------------------
def quat_mul(a, b):
    ax, ay, az, aw = a; bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz])


def quat_from_rotvec(r):
    th = np.linalg.norm(r)
    if th < 1e-12:
        return np.array([0.5 * r[0], 0.5 * r[1], 0.5 * r[2], 1.0]) / np.linalg.norm([0.5 * r[0], 0.5 * r[1], 0.5 * r[2], 1.0])
    ax = np.asarray(r) / th
    return np.array([*(ax * np.sin(th / 2)), np.cos(th / 2)])


def quat_to_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def trajectory_pose(i, n=32):
    """Ground-truth pose of scan i: a gentle arc with roll/pitch/yaw motion, away from every wall."""
    s = i / max(n - 1, 1)
    t = np.array([-4.5 + 5.0 * s, -5.5 + 3.0 * s + 0.6 * np.sin(2 * np.pi * s), 0.25 * np.sin(3.0 * s)])
    q = quat_from_rotvec(np.array([0.03 * np.sin(5 * s), 0.04 * np.cos(3 * s), 0.3 + 0.9 * s]))
    return np.concatenate([t, q])


def perturb_pose(pose, seed, dt=0.10, dth_deg=1.0):
    rng = np.random.default_rng(seed)
    d = (rng.random(3) * 2 - 1) * dt
    r = np.deg2rad((rng.random(3) * 2 - 1) * dth_deg)
    q = quat_mul(pose[3:], quat_from_rotvec(r))
    return np.concatenate([pose[:3] + d, q / np.linalg.norm(q)])


def pose_compose(a, d):
    """a o d, operation by operation as soicp::pose_compose (csrc/so_math.h) evaluates it in IEEE double -- the guess of a chained
    registration (so_icp_register_sequence): t = a.t + R(a.q) d.t (Eigen's _transformVector form), q = normalize(a.q (x) d.q)."""
    a = [float(v) for v in a]; d = [float(v) for v in d]
    q0, q1, q2, q3 = a[3], a[4], a[5], a[6]
    vx, vy, vz = d[0], d[1], d[2]
    ux = q1 * vz - q2 * vy; uy = q2 * vx - q0 * vz; uz = q0 * vy - q1 * vx
    ux += ux; uy += uy; uz += uz
    ox = vx + q3 * ux + (q1 * uz - q2 * uy)
    oy = vy + q3 * uy + (q2 * ux - q0 * uz)
    oz = vz + q3 * uz + (q0 * uy - q1 * ux)
    b0, b1, b2, b3 = d[3], d[4], d[5], d[6]
    w = q3 * b3 - q0 * b0 - q1 * b1 - q2 * b2
    x = q3 * b0 + q0 * b3 + q1 * b2 - q2 * b1
    y = q3 * b1 + q1 * b3 + q2 * b0 - q0 * b2
    z = q3 * b2 + q2 * b3 + q0 * b1 - q1 * b0
    nrm = float(np.sqrt(np.float64(x * x + y * y + z * z + w * w)))
    return np.array([a[0] + ox, a[1] + oy, a[2] + oz, x / nrm, y / nrm, z / nrm, w / nrm])


def pose_between(a, b):
    """the motion d with a o d ~ b (a^-1 o b): what an odometry source predicts between two scans, in the frame of the first"""
    qa = np.asarray(a[3:], float) / np.linalg.norm(a[3:])
    qi = np.array([-qa[0], -qa[1], -qa[2], qa[3]])
    dt = quat_to_R(qi) @ (np.asarray(b[:3], float) - np.asarray(a[:3], float))
    dq = quat_mul(qi, np.asarray(b[3:], float))
    return np.concatenate([dt, dq / np.linalg.norm(dq)])


def pose_error(a, b):
    """(translation distance, rotation angle) between two poses."""
    dt = float(np.linalg.norm(np.asarray(a[:3]) - np.asarray(b[:3])))
    qa = np.asarray(a[3:]) / np.linalg.norm(a[3:]); qb = np.asarray(b[3:]) / np.linalg.norm(b[3:])
    qi = np.array([-qa[0], -qa[1], -qa[2], qa[3]])
    dq = quat_mul(qi, qb)
    return dt, float(2 * np.arctan2(np.linalg.norm(dq[:3]), abs(dq[3])))


# --------------------------------------------------------------------------------------------
# LiDAR
# --------------------------------------------------------------------------------------------
def lidar_dirs(rings, azimuth, fov_deg):
    el = np.deg2rad(np.linspace(-fov_deg, fov_deg, rings))
    az = 2 * np.pi * np.arange(azimuth) / azimuth
    ce, se = np.cos(el)[:, None], np.sin(el)[:, None]
    d = np.stack([ce * np.cos(az)[None, :], ce * np.sin(az)[None, :], np.broadcast_to(se, (rings, azimuth))], -1)
    return d.reshape(-1, 3)


def raycast(world, pose, dirs, seed, sigma=0.01, max_range=100.0, min_range=0.5):
    """Sensor-frame fp32 points, exactly len(dirs) of them.  Rectangles sharing a plane are tested
    together (one ray/plane intersection per plane), which keeps the 128x1024 scans to seconds."""
    rng = np.random.default_rng(seed)
    Rm = quat_to_R(pose[3:]); o = np.asarray(pose[:3], float)
    dw = dirs @ Rm.T
    best = np.full(len(dirs), np.inf)
    uu = np.sum(world.u * world.u, 1); vv = np.sum(world.v * world.v, 1)
    for n, off, members in world.plane_groups():
        denom = dw @ n
        with np.errstate(divide="ignore", invalid="ignore"):
            s = (off - o @ n) / denom
        idx = np.nonzero((s > min_range) & (s < best) & np.isfinite(s))[0]
        if len(idx) == 0:
            continue
        h = o + s[idx, None] * dw[idx]
        hit = np.zeros(len(idx), bool)
        for c0 in range(0, len(members), 64):
            mem = members[c0:c0 + 64]
            rel = h[:, None, :] - world.o[mem][None, :, :]
            a = np.einsum("ijk,jk->ij", rel, world.u[mem]) / uu[mem]
            b = np.einsum("ijk,jk->ij", rel, world.v[mem]) / vv[mem]
            hit |= ((a >= 0) & (a <= 1) & (b >= 0) & (b <= 1)).any(1)
        best[idx[hit]] = s[idx[hit]]
    s = best + rng.standard_normal(len(dirs)) * sigma
    valid = np.isfinite(best) & (s < max_range)
    vi = np.nonzero(valid)[0]
    if len(vi) == 0:
        raise ValueError("no LiDAR returns: sensor outside the world?")
    pts = dirs[vi] * s[vi, None]
    miss = len(dirs) - len(vi)
    if miss > 0:  # re-sample hits (fresh noise) so that exactly Q points come back
        pick = vi[rng.integers(0, len(vi), miss)]
        s2 = best[pick] + rng.standard_normal(miss) * sigma
        pts = np.concatenate([pts, dirs[pick] * s2[:, None]])
    return np.ascontiguousarray(pts.astype(np.float32))


# --------------------------------------------------------------------------------------------
# packaged scenes
# --------------------------------------------------------------------------------------------
class Scene:
    """World + map points + scan factory for one named configuration (cached on disk)."""

    def __init__(self, name="tiny", cache_dir=None, **override):
        cfg = dict(CONFIGS[name]); cfg.update(override)
        self.name, self.cfg = name, cfg
        self.plane_res = cfg["plane_res"]
        self.world = World(extent=cfg["extent"], spacing=cfg["spacing"], seed=1, wall_height=cfg.get("wall_height"))
        self.dirs = lidar_dirs(cfg["rings"], cfg["azimuth"], cfg["fov_deg"])
        cache_dir = cache_dir or os.environ.get("SOICP_CACHE", "/tmp/soicp_cache")
        key = "_".join(f"{k}{cfg[k]}" for k in ("map_points", "extent", "spacing", "plane_res") + (("wall_height",) if "wall_height" in cfg else ()))
        path = os.path.join(cache_dir, f"map_{key}.npy")
        if os.path.exists(path):
            self.map_points = np.load(path)
        else:
            self.map_points = sample_map(self.world, cfg["plane_res"], cfg["map_points"], seed=2)
            try:  # several ranks may build the same scene at once: write aside, then rename (atomic), so that nobody loads half a file
                os.makedirs(cache_dir, exist_ok=True)
                tmp = f"{path}.{os.getpid()}.tmp.npy"
                np.save(tmp, self.map_points); os.replace(tmp, path)
            except OSError:
                pass

    @property
    def n_queries(self):
        return len(self.dirs)

    def gt_pose(self, i):
        return trajectory_pose(i)

    def scan(self, i):
        return raycast(self.world, self.gt_pose(i), self.dirs, seed=3 + i)

    def guess(self, i, dt=0.10, dth_deg=1.0, seed_base=1000):
        return perturb_pose(self.gt_pose(i), seed_base + i, dt, dth_deg)
