"""Synthetic world for filter-level parity tests and bench.py: an analytic trajectory, world points on walls, a
pinhole camera, IMU samples and id'd feature tracks — the "identical IMU + feature-track inputs" both the
oracle and the HIP path are fed with. It follows the shape of the reference's simulator
(src/VIOSimulator.cpp:63-310, src/dataserver/SimulationDataServer.cpp:23-237: wave / sine trajectories, EuRoC
pinhole intrinsics, camera looking along body x, lowest-id feature selection, IMU 200 Hz / camera 20 Hz) but
uses analytic derivatives and numpy's PRNG (the reference uses rand(), not reproducible across libcs)."""
import numpy as np

from eqvio_amd.capi import Camera


def euroc_camera():
    """generatePinholeCameraSquare (src/dataserver/SimulationDataServer.cpp:162-176)."""
    return Camera.pinhole(458.654, 457.296, 367.215, 248.375, 752, 480)


def quat_mul(a, b):
    return np.array([
        a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
        a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
        a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3],
        a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1],
    ])

GRAVITY = 9.80665

# cameraRotation of SimulationDataServer.cpp:234-236 as quaternion (w,x,y,z)
R_IC = np.array([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]])


def mat_to_quat(m):
    t = np.trace(m)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        return np.array([0.25 * s, (m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s])
    i = int(np.argmax(np.diag(m)))
    j, k = (i + 1) % 3, (i + 2) % 3
    s = np.sqrt(m[i, i] - m[j, j] - m[k, k] + 1.0) * 2
    q = np.zeros(4)
    q[0] = (m[k, j] - m[j, k]) / s
    q[1 + i] = 0.25 * s
    q[1 + j] = (m[j, i] + m[i, j]) / s
    q[1 + k] = (m[k, i] + m[i, k]) / s
    return q


def rotz(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])


class SimWorld:
    def __init__(self, seed=0, num_points=2000, max_features=40, trajectory="wave", imu_freq=200.0, image_freq=20.0, noise_px=0.0, camera=None):
        self.rng = np.random.default_rng(seed)
        self.cam = camera or euroc_camera()
        self.max_features = max_features
        self.trajectory = trajectory
        self.imu_freq, self.image_freq = imu_freq, image_freq
        self.noise_px = noise_px
        self.q_ic = mat_to_quat(R_IC)
        self.x_ic = np.zeros(3)
        n = num_points
        if trajectory == "wave":  # cylinder wall around the circular path (generateWaveTrajectory :46-65)
            ang = self.rng.uniform(0, 2 * np.pi, n)
            rad = self.rng.uniform(3.0, 6.0, n)
            self.points = np.stack([rad * np.cos(ang), rad * np.sin(ang), self.rng.uniform(-2.0, 2.0, n)], axis=1)
        else:  # "hover": one wall in front of a gently oscillating camera, every point stays in view
            d = self.rng.uniform(4.0, 8.0, n)
            self.points = np.stack([d, self.rng.uniform(-0.55, 0.55, n) * d, self.rng.uniform(-0.32, 0.32, n) * d], axis=1)

    # ---- analytic trajectory: R(t), x(t), v(t), a(t), omega_body(t)
    def kinematics(self, t):
        if self.trajectory == "wave":
            w = 2 * np.pi / 20.0
            a = w * t
            R = rotz(a)
            x = np.array([np.cos(a), np.sin(a), 0.2 * np.sin(10 * a)])
            v = w * np.array([-np.sin(a), np.cos(a), 2.0 * np.cos(10 * a)])
            acc = w * w * np.array([-np.cos(a), -np.sin(a), -20.0 * np.sin(10 * a)])
            om = np.array([0.0, 0.0, w])
        else:
            w = 2 * np.pi / 8.0
            amp = 0.15
            R = rotz(0.05 * np.sin(w * t))
            x = amp * np.array([np.sin(w * t), np.sin(1.3 * w * t), 0.5 * np.sin(0.7 * w * t)])
            v = amp * w * np.array([np.cos(w * t), 1.3 * np.cos(1.3 * w * t), 0.35 * np.cos(0.7 * w * t)])
            acc = -amp * w * w * np.array([np.sin(w * t), 1.69 * np.sin(1.3 * w * t), 0.245 * np.sin(0.7 * w * t)])
            om = np.array([0.0, 0.0, 0.05 * w * np.cos(w * t)])
        return R, x, v, acc, om

    def imu(self, t):
        R, x, v, acc, om = self.kinematics(t)
        out = np.zeros(13)
        out[0] = t
        out[1:4] = om
        out[4:7] = R.T @ (acc + np.array([0.0, 0.0, GRAVITY]))
        return out

    def sensor_state(self, t):
        R, x, v, acc, om = self.kinematics(t)
        s = np.zeros(23)
        s[6:10] = mat_to_quat(R)
        s[10:13] = x
        s[13:16] = R.T @ v
        s[16:20] = self.q_ic
        s[20:23] = self.x_ic
        return s

    def camera_points(self, t):
        R, x, _, _, _ = self.kinematics(t)
        Rc = R @ R_IC
        xc = x + R @ self.x_ic
        return (self.points - xc) @ Rc  # rows: Rc^T (p - xc)

    def vision(self, t):
        """ids (ascending) and pixel coordinates of the max_features lowest-id visible points (VIOSimulator.cpp:216-265)."""
        pc = self.camera_points(t)
        cam = self.cam
        z = pc[:, 2]
        ok = z > 0.1
        u = np.where(ok, cam.fx * pc[:, 0] / np.where(ok, z, 1.0) + cam.cx, -1.0)
        v = np.where(ok, cam.fy * pc[:, 1] / np.where(ok, z, 1.0) + cam.cy, -1.0)
        vis = ok & (u >= 0) & (u < cam.width) & (v >= 0) & (v < cam.height)
        ids = np.nonzero(vis)[0][: self.max_features].astype(np.int32)
        y = np.stack([u[ids], v[ids]], axis=1)
        if self.noise_px > 0:
            y = y + self.rng.normal(size=y.shape) * self.noise_px
        return ids, y.reshape(-1)

    def true_state(self, t, ids=None):
        pc = self.camera_points(t)
        if ids is None:
            ids = np.arange(len(pc), dtype=np.int32)
        return self.sensor_state(t), np.asarray(ids, dtype=np.int32), pc[np.asarray(ids)]

    def frames(self, n_frames, t0=0.0):
        """Yield (imu_samples[k,13], stamp, ids, y) per camera frame; IMU stamps lie in (prev_stamp, stamp]."""
        k = int(round(self.imu_freq / self.image_freq))
        for f in range(n_frames):
            stamp = t0 + (f + 1) / self.image_freq
            ts = t0 + f / self.image_freq + (np.arange(k) + 0.0) / self.imu_freq
            imus = np.stack([self.imu(tt) for tt in ts])
            ids, y = self.vision(stamp)
            yield imus, stamp, ids, y
