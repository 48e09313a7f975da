"""Seeded synthetic inputs shared by the parity tests (shapes follow the reference's test fixtures,
test/testing_utilities.cpp:24-124 and SURVEY.md §8(d) config 3) and parity metrics."""
import numpy as np

from eqvio_amd.capi import COORD_EUCLIDEAN, COORD_INVDEPTH, Camera, Settings


def unit_quat(rng):
    q = rng.normal(size=4)
    return q / np.linalg.norm(q)


def so3_exp(w):
    th = np.linalg.norm(w)
    if th < 1e-12:
        return np.array([1.0, 0.5 * w[0], 0.5 * w[1], 0.5 * w[2]])
    return np.concatenate([[np.cos(th / 2)], np.sin(th / 2) * w / th])


def quat_mul(a, b):
    return np.array([
        a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
        a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
        a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3],
        a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1],
    ])


def quat_rot(q, v):
    u = q[1:]
    uv = 2 * np.cross(u, v)
    return v + q[0] * uv + np.cross(u, uv)


def default_camera():
    """createDefaultCamera (test/testing_utilities.cpp:175-184)."""
    return Camera.pinhole(450.0, 450.0, 400.0, 240.0, 800, 480)


def euroc_radtan_camera():
    """EuRoC cam0 with its radial-tangential distortion (reference intrinsics.yaml:7-8)."""
    return Camera.radtan(458.654, 457.296, 367.215, 248.375, 752, 480, -0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0)


def uzhfpv_equidistant_camera():
    """UZH-FPV indoor forward snapdragon cam0-like equidistant camera (640x480, f ~ 278; Kannala-Brandt k1..k4)."""
    return Camera.equidistant(278.66723066149086, 278.48991409740296, 319.75221200593535, 241.96858910358173, 640, 480, -0.013721808247486035, 0.020727425669427896,
                              -0.012786476702685545, 0.0025242267320687625)


CAMERAS = {"pinhole": default_camera, "radtan": euroc_radtan_camera, "equidistant": uzhfpv_equidistant_camera}


def euroc_camera():
    """generatePinholeCameraSquare (src/dataserver/SimulationDataServer.cpp:162-176)."""
    return Camera.pinhole(458.654, 457.296, 367.215, 248.375, 752, 480)


def reasonable_state(rng, N, id_offset=0, shuffle_ids=False):
    """reasonableStateElement + reasonableGroupElement. Returns xi0_sensor[23], X_sensor[23], ids, q0[N,3], Q[N,5]."""
    xi0 = np.zeros(23)
    xi0[0:6] = rng.uniform(-1, 1, 6) * 0.1
    xi0[6:10] = unit_quat(rng)
    xi0[10:13] = rng.uniform(-1, 1, 3)
    xi0[13:16] = rng.uniform(-1, 1, 3)
    xi0[16:20] = unit_quat(rng)
    xi0[20:23] = rng.uniform(-1, 1, 3) * 0.2
    Xs = np.zeros(23)
    Xs[0:6] = rng.uniform(-1, 1, 6) * 0.1
    Xs[6:10] = so3_exp(rng.uniform(-1, 1, 3) * 0.1)
    Xs[10:13] = rng.uniform(-1, 1, 3) * 0.1
    Xs[13:16] = rng.uniform(-1, 1, 3) * 0.1
    Xs[16:20] = so3_exp(rng.uniform(-1, 1, 3) * 0.1)
    Xs[20:23] = rng.uniform(-1, 1, 3) * 0.1
    ids = np.arange(N, dtype=np.int32) * 3 + id_offset
    if shuffle_ids:
        ids = rng.permutation(ids).astype(np.int32)
    q0 = rng.uniform(-1, 1, (N, 3)) * 10.0
    q0[:, 2] += 20.0
    Q = np.zeros((N, 5))
    for i in range(N):
        Q[i, :4] = so3_exp(rng.uniform(-1, 1, 3) * 0.02)
        Q[i, 4] = 2.0 * rng.uniform() + 1.0
    return xi0, Xs, ids, q0, Q


def random_spd(rng, n, scale=0.05, diag=None):
    """A covariance with realistic correlation: diag + low-rank coupling (SPD by construction)."""
    G = rng.normal(size=(n, max(8, n // 8))) * scale
    S = G @ G.T
    d = np.full(n, 0.5) if diag is None else np.asarray(diag)
    S[np.diag_indices(n)] += d
    return 0.5 * (S + S.T)


def estimate_landmarks(q0, Q):
    """q_hat_i = Q_i^-1 q0_i = (1/a) R^T q0 (src/mathematical/VIOGroup.cpp:44-52)."""
    out = np.zeros_like(q0)
    for i in range(len(q0)):
        qi = Q[i, :4] * np.array([1, -1, -1, -1])
        out[i] = quat_rot(qi, q0[i]) / Q[i, 4]
    return out


def project(cam, p):
    """Pixels of camera-frame points for any model of include/eqvio_types.h (only used to synthesise measurements)."""
    x, y = p[:, 0] / p[:, 2], p[:, 1] / p[:, 2]
    d = list(cam.dist)
    if cam.model == 1:
        r2 = x * x + y * y
        rad = 1 + d[0] * r2 + d[1] * r2**2 + d[4] * r2**3
        x, y = x * rad + 2 * d[2] * x * y + d[3] * (r2 + 2 * x * x), y * rad + d[2] * (r2 + 2 * y * y) + 2 * d[3] * x * y
    elif cam.model == 2:
        r = np.sqrt(x * x + y * y)
        th = np.arctan(r)
        s = np.where(r > 1e-8, th * (1 + d[0] * th**2 + d[1] * th**4 + d[2] * th**6 + d[3] * th**8) / np.maximum(r, 1e-300), 1.0)
        x, y = s * x, s * y
    return np.stack([cam.fx * x + cam.cx, cam.fy * y + cam.cy], axis=1)


def synth_measurement(rng, cam, ids, q0, Q, noise_px=1.0, subset=None):
    """y = project(q_hat) + N(0, noise^2), returned sorted by ascending id (the reference's std::map order)."""
    qh = estimate_landmarks(q0, Q)
    y = project(cam, qh) + rng.normal(size=(len(ids), 2)) * noise_px
    sel = np.arange(len(ids)) if subset is None else np.asarray(subset)
    order = np.argsort(ids[sel])
    sel = sel[order]
    return ids[sel].astype(np.int32), y[sel].reshape(-1)


def random_imu(rng, stamp=0.0, bias_vel=False):
    imu = np.zeros(13)
    imu[0] = stamp
    imu[1:4] = rng.uniform(-1, 1, 3)
    imu[4:7] = rng.uniform(-1, 1, 3) + np.array([0, 0, 9.8])
    if bias_vel:
        imu[7:13] = rng.uniform(-1, 1, 6) * 0.01
    return imu


def settings_for(chart, **kw):
    s = Settings.defaults()
    s.coordinateChoice = chart
    for k, v in kw.items():
        setattr(s, k, v)
    return s


def rel_fro(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


CHARTS = {"euclid": COORD_EUCLIDEAN, "invdepth": COORD_INVDEPTH}


def teacher_force(flt, orc):
    """Reset the device filter to the oracle's (xi0, X, Sigma) (SURVEY.md section 8(d) "Parity definition": teacher-forced parity compares ONE frame's
    arithmetic at a time; a free-running comparison also measures how the configuration amplifies last-bit differences over the frames before)."""
    xi0, Xs, ids, q0, Q = orc.get_eqf()
    _, _, ids_f, _, _ = flt.get_eqf()
    assert np.array_equal(ids, ids_f), "teacher forcing needs identical landmark bookkeeping"
    flt.force_eqf(xi0, Xs, ids, q0, Q, orc.get_sigma())
