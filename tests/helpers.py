"""Shared helpers for the parity tests (test infrastructure)."""
import numpy as np

from superodom_amd import synth


def build_oracle_map_from(oracle, points, plane_res, origin_t, raw=True):
    m = oracle.OracleMap(plane_res=plane_res)
    m.set_origin(np.asarray(origin_t, float))
    m.add_surf(points, raw=raw)
    return m


def pose_close(a, b, tol_t=1e-4, tol_r=1e-4):
    dt, dr = synth.pose_error(a, b)
    return dt <= tol_t and dr <= tol_r, dt, dr


def noisy_planes_cloud(n, rng, offset=(0.0, 0.0, 0.0), sigma=0.01):
    """Four noisy planes (floor/ceiling/2 walls) inside one 50 m cube -- the SURVEY App. D probe scene."""
    k = n // 4
    parts = []
    xy = rng.random((k, 2)) * 40 - 20
    parts.append(np.c_[xy, -1.5 + sigma * rng.standard_normal(k)])
    xy = rng.random((k, 2)) * 40 - 20
    parts.append(np.c_[xy, 6.0 + sigma * rng.standard_normal(k)])
    xz = np.c_[rng.random(k) * 40 - 20, rng.random(k) * 7.5 - 1.5]
    parts.append(np.c_[xz[:, 0], 10.0 + sigma * rng.standard_normal(k), xz[:, 1]])
    yz = np.c_[rng.random(n - 3 * k) * 40 - 20, rng.random(n - 3 * k) * 7.5 - 1.5]
    parts.append(np.c_[-12.0 + sigma * rng.standard_normal(n - 3 * k), yz[:, 0], yz[:, 1]])
    return (np.concatenate(parts) + np.asarray(offset)).astype(np.float32)


# ------------------------------------------------------------------------------------------------------------------------
# A scene on which the reference's OWN k-NN engine (flann/octree.h, compiled into oracle/_ref) is exact, so that a whole
# registration can be run with it in the loop and compared with the exact-search oracle and the product.
#
# The stock octree has two defects: initialize() derives the y / z bounds of the root cube from comparisons
# against x (oct.h:384-385), and inside() tests the query's X coordinate against all three octant centres (oct.h:988-990).
# Both are inert when, in every 50 m block, (a) the points' x range is at least as large as their y and z ranges -- the root
# cube, sized by the x range, then contains every point wherever the broken y / z centres land -- and (b) x is far from every
# y and z coordinate of the block -- inside() then never answers "yes" early and the search degenerates to an exhaustive,
# correctly pruned one.  A corridor along x at x ~ 1 100 m (y within +-6.5 m, z within -1.5 .. 4.1 m) has both properties.
# ------------------------------------------------------------------------------------------------------------------------
class CorridorScene:
    X0, LEN = 1003.7, 170.0

    def __init__(self, plane_res=0.2, rings=32, azimuth=512, fov_deg=22.5, seed=21):
        rng = np.random.default_rng(seed)
        R = []

        def add(o, u, v):
            R.append((np.array(o, float), np.array(u, float), np.array(v, float)))
        x0, L, z0, z1, ya, yb = self.X0, self.LEN, -1.5, 4.1, -6.3, 5.9
        H = z1 - z0
        add([x0, ya, z0], [L, 0, 0], [0, yb - ya, 0])          # floor
        add([x0, ya, z1], [L, 0, 0], [0, yb - ya, 0])          # ceiling
        add([x0, ya, z0], [L, 0, 0], [0, 0, H])                # side walls
        add([x0, yb, z0], [L, 0, 0], [0, 0, H])
        add([x0, ya, z0], [0, yb - ya, 0], [0, 0, H])          # end walls
        add([x0 + L, ya, z0], [0, yb - ya, 0], [0, 0, H])
        k = 0
        for xc in np.arange(x0 + 11.6, x0 + L - 5.0, 12.3):    # partial cross walls, alternating sides: they fix the pose along x
            if k % 2 == 0:
                add([xc, ya, z0], [0, 4.6 + rng.random(), 0], [0, 0, H])
            else:
                add([xc, yb, z0], [0, -(4.3 + rng.random()), 0], [0, 0, H])
            k += 1
        for _ in range(14):                                     # leaning panels: roll / pitch observability
            cx = x0 + 4.0 + rng.random() * (L - 12.0); cy = ya + 0.8 + rng.random() * (yb - ya - 4.5)
            Lp, Wp = 1.5 + 1.5 * rng.random(), 1.5 + 1.5 * rng.random()
            if rng.random() < 0.5:
                add([cx, cy, z0], [Lp, 0, Lp], [0, Wp, 0])
            else:
                add([cx, cy, z0], [0, Lp, Lp], [Wp, 0, 0])
        w = synth.World.__new__(synth.World)
        w.o = np.stack([r[0] for r in R]); w.u = np.stack([r[1] for r in R]); w.v = np.stack([r[2] for r in R])
        n = np.cross(w.u, w.v)
        w.n = n / np.linalg.norm(n, axis=1, keepdims=True)
        w.extent = L
        w._groups = None
        self.world = w
        self.plane_res = plane_res
        self.map_points = synth.sample_map(w, plane_res, None, seed=seed + 1)
        self.dirs = synth.lidar_dirs(rings, azimuth, fov_deg)

    def gt_pose(self, i):
        s = i / 7.0
        t = np.array([self.X0 + 62.0 + 9.0 * s, -0.9 + 1.1 * np.sin(2.0 * s), 0.15 * np.cos(3.0 * s)])
        q = synth.quat_from_rotvec(np.array([0.02 * np.sin(5 * s), 0.03 * np.cos(3 * s), 0.2 + 0.7 * s]))
        return np.concatenate([t, q])

    def scan(self, i):
        return synth.raycast(self.world, self.gt_pose(i), self.dirs, seed=40 + i)

    def guess(self, i, dt=0.10, dth_deg=1.0):
        return synth.perturb_pose(self.gt_pose(i), 2000 + i, dt, dth_deg)


# ------------------------------------------------------------------------------------------------------------------------
# Rank-deficient geometry (VERDICT r04 "weak" 1 / "next" 6b): scenes whose normal equations J^T J are ill conditioned because part
# of the pose is not observable.  The product solves the LM step by Cholesky on the scaled normal equations (lm_solver.h), the
# oracle by QR on the stacked Jacobian like Ceres DENSE_QR: normal equations square the condition number, so this is where the
# two could part.  sigma = noise of map and scan along the normals (LidarSlam.cpp:772 rejects noise-free planes).
# ------------------------------------------------------------------------------------------------------------------------
DEGENERATE_RECTS = {
    # one plane: x, y and yaw are unobservable (they move only through the 1 cm noise of the normals)
    "floor_only": [([-20, -20, -1.5], [40, 0, 0], [0, 40, 0])],
    # corridor along x without end or cross walls: the translation along x is unobservable
    "open_corridor": [([-24, -6.3, -1.5], [48, 0, 0], [0, 12.2, 0]), ([-24, -6.3, 4.1], [48, 0, 0], [0, 12.2, 0]),
                      ([-24, -6.3, -1.5], [48, 0, 0], [0, 0, 5.6]), ([-24, 5.9, -1.5], [48, 0, 0], [0, 0, 5.6])],
    # two parallel walls: only the translation along their normal and the two rotations that tilt them are observable
    "two_walls": [([-20, -6.3, -1.5], [40, 0, 0], [0, 0, 7.5]), ([-20, 5.9, -1.5], [40, 0, 0], [0, 0, 7.5])],
    # a room corner AT THE WORLD ORIGIN: floor z = 0 and the walls x = 0, y = 0 all contain the origin, where the reference's plane
    # parameterisation A x = -1 (LidarSlam.cpp:798-816) is ill conditioned (|x| = 1 / offset; only the 1 cm noise keeps it finite).
    # Well conditioned as a registration (three orthogonal planes): every pose component is observable.
    "origin_corner": [([0, 0, 0], [24, 0, 0], [0, 24, 0]), ([0, 0, 0], [0, 24, 0], [0, 0, 8]), ([0, 0, 0], [24, 0, 0], [0, 0, 8])],
}
# rows of the 6-vector [dt_x, dt_y, dt_z, rotvec_x, rotvec_y, rotvec_z] (world frame) the scene DOES constrain
DEGENERATE_OBSERVABLE = {"floor_only": [2, 3, 4], "open_corridor": [1, 2, 3, 4, 5], "two_walls": [1, 3, 5], "origin_corner": [0, 1, 2, 3, 4, 5]}
DEGENERATE_SENSOR_OFFSET = {"origin_corner": [5.3, 4.9, 1.8]}  # (the sensor stands inside the corner, 2 m above the floor)


class DegenerateScene:
    def __init__(self, name, plane_res=0.2, rings=32, azimuth=512, fov_deg=22.5, sigma=0.01):
        rects = DEGENERATE_RECTS[name]
        w = synth.World.__new__(synth.World)
        w.o = np.stack([np.array(r[0], float) for r in rects]); w.u = np.stack([np.array(r[1], float) for r in rects])
        w.v = np.stack([np.array(r[2], float) for r in rects])
        n = np.cross(w.u, w.v)
        w.n = n / np.linalg.norm(n, axis=1, keepdims=True)
        w.extent = 24.0
        w._groups = None
        self.name, self.world, self.plane_res, self.sigma = name, w, plane_res, sigma
        self.map_points = synth.sample_map(w, plane_res, None, seed=5, sigma=sigma)
        self.dirs = synth.lidar_dirs(rings, azimuth, fov_deg)
        self.observable = DEGENERATE_OBSERVABLE[name]

    def gt_pose(self, i):
        off = np.array(DEGENERATE_SENSOR_OFFSET.get(self.name, [0, 0, 0]), float)
        return np.concatenate([off + [0.7 + 0.5 * i, -0.9 + 0.3 * i, 0.2], synth.quat_from_rotvec(np.array([0.02, -0.03, 0.4 + 0.3 * i]))])

    def scan(self, i):
        return synth.raycast(self.world, self.gt_pose(i), self.dirs, seed=60 + i, sigma=self.sigma)

    def guess(self, i, dt=0.10, dth_deg=1.0):
        return synth.perturb_pose(self.gt_pose(i), 3000 + i, dt, dth_deg)


def pose_delta6(a, b):
    """[dt (world), rotation vector of qa^-1 qb rotated to the world frame] -- small-angle 6-vector between two poses"""
    a, b = np.asarray(a, float), np.asarray(b, float)
    qa = a[3:] / np.linalg.norm(a[3:]); qb = b[3:] / np.linalg.norm(b[3:])
    dq = synth.quat_mul(np.array([-qa[0], -qa[1], -qa[2], qa[3]]), qb)
    if dq[3] < 0:
        dq = -dq
    rv = 2.0 * dq[:3]
    return np.concatenate([b[:3] - a[:3], synth.quat_to_R(qa) @ rv])
