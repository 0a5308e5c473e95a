"""Checker for adapter/laser_mapping_soicp.cpp: the per-frame logic of laserMapping (laserMapping.cpp:265-381 setInitialGuess,
:729-765 updatePoseAndPublish) restated on scipy.spatial.transform.Rotation -- a different code path from the hand-written
Eigen / tf2 arithmetic of adapter/node_math.h.  Quaternions are (x, y, z, w); poses are [tx ty tz qx qy qz qw].
Test infrastructure only."""
import numpy as np
from scipy.spatial.transform import Rotation as R


def _rot(q):
    return R.from_quat(np.asarray(q, float))


def extract_roll_pitch(q):
    """utils::extractRollPitch: tf2 getRPY -> setRPY(roll, pitch, 0); tf2's fixed-axis RPY = scipy's extrinsic 'xyz'"""
    roll, pitch, _ = _rot(q).as_euler("xyz")
    return R.from_euler("xyz", [roll, pitch, 0.0]).as_quat()


def compose(a, b):
    """Transformd a * b"""
    ra = _rot(a[3:])
    return np.concatenate([ra.apply(b[:3]) + a[:3], (ra * _rot(b[3:])).as_quat()])


def inverse(a):
    ri = _rot(a[3:]).inv()
    return np.concatenate([-ri.apply(a[:3]), ri.as_quat()])


def same_rotation(qa, qb, tol):
    return (_rot(qa).inv() * _rot(qb)).magnitude() <= tol


class NodeMirror:
    """State machine of the node between Localization() calls."""

    def __init__(self, localization_mode=False):
        self.initialization = False
        self.startup_count = 10
        self.T = np.array([0, 0, 0, 0, 0, 0, 1.0])
        self.last_T = self.T.copy()
        self.q_wodom_pre = np.array([0, 0, 0, 1.0])
        self.q_wodom_curr = np.array([0, 0, 0, 1.0])
        self.q_w_curr = np.array([0, 0, 0, 1.0])
        self.source = "IMU Only Orientation Prediction"
        self.t_prev = 0.0

    def initial_guess(self, imu_q):
        imu_q = np.asarray(imu_q, float)
        n = np.linalg.norm(imu_q)
        imu = imu_q / n if n > 0 else imu_q
        has_imu = imu[3] != 0
        if not self.initialization:  # initializeFirstFrame (identity extrinsic)
            self.q_w_curr = extract_roll_pitch(imu) if has_imu else np.array([0, 0, 0, 1.0])
            self.q_wodom_pre = self.q_w_curr.copy()
            self.T = np.concatenate([np.zeros(3), self.q_w_curr])
        elif self.startup_count > 0:  # initializeWithIMU
            self.T = np.concatenate([self.last_T[:3], imu]) if has_imu else self.last_T.copy()
            self.q_w_curr = self.T[3:].copy()
            self.startup_count -= 1
        elif has_imu:  # IMU_ORIENTATION
            self.q_wodom_curr = imu
            q = (_rot(self.q_w_curr) * _rot(self.q_wodom_pre).inv() * _rot(self.q_wodom_curr)).as_quat()
            self.T = np.concatenate([self.T[:3], q])
            self.q_wodom_pre = self.q_wodom_curr.copy()
            self.q_w_curr = q
            self.source = "IMU Only Orientation Prediction"
        else:  # CONSTANT_VELOCITY
            self.T = compose(self.T, compose(inverse(self.last_T), self.T))
            self.q_w_curr = self.T[3:].copy()
            self.source = "Using Constant Velocity Prediction"
        return self.T.copy()

    def update(self, pose_out, startup_count, t):
        """updatePoseAndPublish: returns (vel_b, ang_vel_b)"""
        pose_out = np.asarray(pose_out, float)
        self.T = pose_out.copy()
        self.q_w_curr = pose_out[3:].copy()
        self.startup_count = startup_count
        self.initialization = True
        dt = t - self.t_prev
        if dt > 1e-6:
            rq = _rot(pose_out[3:])
            vel_b = rq.inv().apply((pose_out[:3] - self.last_T[:3]) / dt)
            ang_b = rq.inv().apply((rq * _rot(self.last_T[3:]).inv()).as_rotvec() / dt)
        else:
            vel_b, ang_b = np.zeros(3), np.zeros(3)
        self.last_T = pose_out.copy()
        self.t_prev = t
        return vel_b, ang_b
