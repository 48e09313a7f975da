"""The synthetic-world data server (eqvio_amd/host/VIOSimulator.*, include/eqvio_sim.h), SURVEY.md §8 row f-1.

CPU part: the measurement model of the reference's VIOSimulator / SimulationDataServer (src/VIOSimulator.cpp,
src/dataserver/SimulationDataServer.cpp) checked through properties the reference's design implies, and the oracle
filter driven main_sim-style (src/main_sim.cpp:128-184) by the C++ simulator.
GPU part: the same run on the device filter, parity against the oracle frame by frame, and the eqvio_sim executable."""
import math
import os
import re
import subprocess

import numpy as np
import pytest

from eqvio_amd.capi import COORD_EUCLIDEAN, COORD_INVDEPTH, Settings, SimSettings, SimulationDataServer
from oracle_binding import OracleFilter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PI_REF = 3.14  # the reference's trajectory generators use 3.14, not pi (SimulationDataServer.cpp:58)


def filter_settings(chart=COORD_EUCLIDEAN, **kw):
    s = Settings.defaults()
    s.coordinateChoice = chart
    s.fastRiccati = 1
    s.initialPointVariance = 1e-2
    s.measurementNoise = 0.5
    s.useMedianDepth = 1
    for k, v in kw.items():
        setattr(s, k, v)
    return s


def make_server(fs=None, **kw):
    fs = fs or filter_settings()
    sim = SimSettings.defaults(**{"duration": 3.0, "randomSeed": 11, **kw})
    srv = SimulationDataServer(sim, fs)
    fs.cameraOffset[:] = srv.camera_offset()  # main_sim.cpp:97-101
    return srv, fs


def qrot(q, v):
    w, u = q[0], np.asarray(q[1:4])
    return v + 2 * w * np.cross(u, v) + 2 * np.cross(u, np.cross(u, v))


def qmul(a, b):
    return np.array([a[0] * b[0] - np.dot(a[1:], b[1:]), *(a[0] * b[1:] + b[0] * a[1:] + np.cross(a[1:], b[1:]))])


def test_measurement_schedule():
    """nextMeasurementType (SimulationDataServer.cpp:182-190): image first at equal stamps, 200 Hz IMU / 20 Hz image."""
    srv, _ = make_server(duration=1.0)
    kinds, stamps = [], []
    while srv.next_measurement_type() != srv.NONE:
        k = srv.next_measurement_type()
        t = srv.next_time()
        kinds.append(k)
        stamps.append(t)
        if k == srv.IMAGE:
            st, _, _ = srv.get_vision()
        else:
            st = srv.get_imu()[0]
        assert st == t
    assert kinds[0] == srv.IMAGE and kinds[1] == srv.IMU
    assert kinds.count(srv.IMAGE) == 20 and kinds.count(srv.IMU) == 200
    assert stamps == sorted(stamps)
    assert math.isnan(srv.next_time())


def test_wave_trajectory_imu_and_state():
    """wave trajectory (SimulationDataServer.cpp:45-65): yaw rate 2*3.14/20 rad/s about z; position on the unit circle
    with a 0.2 m vertical wave; getIMU / getFullState differentiate it numerically (VIOSimulator.cpp:129-208, 272-309)."""
    srv, _ = make_server(duration=2.0)
    t0 = 0.5 / 200.0  # initialTime = 0.5 / imuFreq (SimulationDataServer.cpp:145)
    for t in [0.3, 0.777, 1.5]:
        s, ids, p = srv.true_state(t)
        ang = PI_REF * 2 * (t + t0) / 20.0
        np.testing.assert_allclose(s[10:13], [math.cos(ang), math.sin(ang), 0.2 * math.sin(10 * ang)], atol=1e-8)
        np.testing.assert_allclose(s[6:10], [math.cos(ang / 2), 0, 0, math.sin(ang / 2)], atol=1e-9)
        w = PI_REF * 2 / 20.0
        v_in = np.array([-math.sin(ang) * w, math.cos(ang) * w, 2.0 * math.cos(10 * ang) * w])
        np.testing.assert_allclose(qrot([s[6], -s[7], -s[8], -s[9]], v_in), s[13:16], atol=1e-6)
        assert np.all(s[0:6] == 0)
        assert len(ids) == 1000 and sorted(ids.tolist()) == list(range(1000))
    while srv.next_measurement_type() != srv.NONE:
        if srv.next_measurement_type() == srv.IMAGE:
            srv.get_vision()
            continue
        imu = srv.get_imu()
        t = imu[0]
        if t < 0.05:
            continue
        ang = PI_REF * 2 * (t + t0) / 20.0
        w = PI_REF * 2 / 20.0
        np.testing.assert_allclose(imu[1:4], [0, 0, w], atol=1e-9)
        a_in = np.array([-math.cos(ang) * w * w, -math.sin(ang) * w * w, -20.0 * math.sin(10 * ang) * w * w])
        q = [math.cos(ang / 2), 0, 0, -math.sin(ang / 2)]  # R^-1
        np.testing.assert_allclose(imu[4:7], qrot(q, a_in + np.array([0, 0, 9.80665])), atol=2e-4)
        assert np.all(imu[7:13] == 0)


@pytest.mark.parametrize("traj", ["wave", "square", "sine", "line"])
def test_vision_matches_true_state(traj):
    """getVision (VIOSimulator.cpp:210-268): at most maxFeatures visible points, ascending ids, pixels = projection of the
    true camera-frame landmark (the pose interpolation differs from getFullState's cubic fit by O(dt^2) only)."""
    srv, _ = make_server(duration=1.0, trajectory=traj, maxFeatures=25, numWalls=4)
    cam = srv.cam
    seen = 0
    while srv.next_measurement_type() != srv.NONE:
        if srv.next_measurement_type() == srv.IMU:
            srv.get_imu()
            continue
        stamp, ids, y = srv.get_vision()
        if stamp < 0.02:
            continue  # the boundary handling at the first poses extrapolates (VIOSimulator.cpp:142-147)
        assert len(ids) <= 25 and np.all(np.diff(ids) > 0)
        s, tids, tp = srv.true_state(stamp)
        lut = {int(i): k for k, i in enumerate(tids)}
        for j, i in enumerate(ids):
            q = tp[lut[int(i)]]
            assert q[2] > 0
            u, v = cam.fx * q[0] / q[2] + cam.cx, cam.fy * q[1] / q[2] + cam.cy
            assert abs(u - y[2 * j]) < 5e-2 and abs(v - y[2 * j + 1]) < 5e-2
            assert 0 <= y[2 * j] < cam.width and 0 <= y[2 * j + 1] < cam.height
        seen += len(ids)
    assert seen > 0


def test_world_points_sit_on_the_walls():
    """generateWorldPoints (VIOSimulator.cpp:65-127): numWalls = 1 puts every point on the +x wall of the trajectory box
    (wallDistance outside the trajectory); numWalls = 4 uses the +x, +y, -y, -x walls in equal shares."""
    for walls in (1, 4):
        srv, _ = make_server(duration=1.0, numWalls=walls, wallDistance=2.0, numPoints=400)
        s, ids, p = srv.true_state(0.5)
        PC_R, PC_x = qmul(s[6:10], s[16:20]), s[10:13] + qrot(s[6:10], s[20:23])
        world = np.array([qrot(PC_R, q) + PC_x for q in p])
        world = world[np.argsort(ids)]  # undo the shuffle: wall index = (numWalls * id) // num
        wall = (walls * np.arange(400)) // 400
        assert np.ptp(world[wall == 0][:, 0]) < 1e-9 and world[wall == 0][0, 0] > 2.9  # x = max(traj x) + 2
        if walls == 4:
            assert np.ptp(world[wall == 1][:, 1]) < 1e-9 and np.ptp(world[wall == 2][:, 1]) < 1e-9 and np.ptp(world[wall == 3][:, 0]) < 1e-9
            assert world[wall == 1][0, 1] > 0 > world[wall == 2][0, 1] and world[wall == 3][0, 0] < 0


def test_seed_reproducibility_and_noise_flags():
    a, _ = make_server(randomSeed=5)
    b, _ = make_server(randomSeed=5)
    c, _ = make_server(randomSeed=6)
    assert np.array_equal(a.true_state(0.5)[2], b.true_state(0.5)[2])
    assert not np.array_equal(a.true_state(0.5)[2], c.true_state(0.5)[2])
    n, _ = make_server(randomSeed=5, inputNoise=1, outputNoise=1, initialNoise=1)
    clean_imu, noisy_imu = a.get_imu() if a.next_measurement_type() == a.IMU else None, None
    sa, _, pa = a.true_state(0.5, True)  # initialNoise off: withNoise has no effect
    sn, _, pn = n.true_state(0.5, True)
    assert np.array_equal(sa, a.true_state(0.5, False)[0])
    assert not np.array_equal(sa, sn) and not np.array_equal(pa, pn)
    assert np.array_equal(n.true_state(0.5, False)[0], sa)  # noise only when asked for


def drive(srv, filters, frames, on_frame=None):
    """main_sim.cpp:128-184: Image -> augmentLandmarkStates(ids, true state) -> processVisionData; IMU -> processIMUData."""
    k = 0
    while srv.next_measurement_type() != srv.NONE and k < frames:
        if srv.next_measurement_type() == srv.IMU:
            imu = srv.get_imu()
            for f in filters:
                f.process_imu(imu)
            continue
        stamp, ids, y = srv.get_vision()
        s, tids, tp = srv.true_state(stamp, True)
        for f in filters:
            f.augment_landmark_states(ids, s, tids, tp)
            f.process_vision(stamp, srv.cam, ids, y)
        k += 1
        if on_frame:
            on_frame(stamp)


@pytest.mark.parametrize("chart", [COORD_EUCLIDEAN, COORD_INVDEPTH])
def test_oracle_filter_tracks_the_simulated_world(chart):
    """The C++ simulator's data is self-consistent: the (CPU) oracle filter, started at the true state and driven
    main_sim-style for 3 s with exact measurements, stays at the truth and is consistent (NEES well below 1)."""
    fs = filter_settings(chart)
    srv, fs = make_server(fs, duration=3.0, maxFeatures=20)
    s0, ids0, p0 = srv.true_state(0.0, True)
    orc = OracleFilter(fs, s0, ids0, p0, 0.0)
    nees = []

    def on_frame(stamp):
        ts, tids, tp = srv.true_state(orc.get_time())
        est, eids, ep = orc.state_estimate()
        assert len(eids) <= 20
        nees.append(orc.compute_nees(ts, tids, tp))
        assert np.linalg.norm(est[10:13] - ts[10:13]) < 2e-2

    drive(srv, [orc], 60, on_frame)
    assert len(nees) == 60 and np.all(np.isfinite(nees)) and max(nees[5:]) < 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("chart,fast", [(COORD_EUCLIDEAN, 1), (COORD_INVDEPTH, 1), (COORD_EUCLIDEAN, 0)])
def test_device_filter_matches_oracle_on_the_simulated_world(chart, fast):
    """SURVEY.md §8(d) config 1 (main_sim, 20 features): device filter vs oracle, frame by frame, 1e-9."""
    from eqvio_amd.capi import VIOFilter
    from test_gpu_filter import compare

    fs = filter_settings(chart, fastRiccati=fast)
    srv, fs = make_server(fs, duration=2.0, maxFeatures=20, numWalls=4)
    s0, ids0, p0 = srv.true_state(0.0, True)
    orc = OracleFilter(fs, s0, ids0, p0, 0.0)
    flt = VIOFilter(fs, max_landmarks=1024, sensor=s0, ids=ids0, p=p0, time=0.0)

    def on_frame(stamp):
        compare(flt, orc)
        ts, tids, tp = srv.true_state(stamp)
        a, b = flt.compute_nees(ts, tids, tp), orc.compute_nees(ts, tids, tp)
        assert abs(a - b) <= 1e-7 * max(1.0, abs(b))

    drive(srv, [orc, flt], 30 if fast else 12, on_frame)


@pytest.mark.gpu
def test_eqvio_sim_executable(tmp_path):
    """The simulation main (eqvio_amd/host/main_sim.cpp) end to end, noisy measurements, InvDepth, fast Riccati."""
    exe = os.path.join(ROOT, "eqvio_amd", "lib", "eqvio_sim")
    out = subprocess.run([exe, "--duration", "5", "--maxFeatures", "30", "--numWalls", "4", "--seed", "3", "--coordinateChoice", "InvDepth", "--fastRiccati", "1",
                          "--outputNoise", "--inputNoise", "--measurementNoise", "0.5", "--initialPointVariance", "0.01", "--quiet", "--output", str(tmp_path / "run")],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "Processed 1000 IMU and 100 vision measurements." in out.stdout
    m = re.search(r"mean NEES ([0-9.eE+-]+)\s+max NEES ([0-9.eE+-]+)\s+final position error ([0-9.eE+-]+)", out.stdout)
    assert m, out.stdout
    mean_nees, max_nees, pos_err = map(float, m.groups())
    assert 0.0 < mean_nees < 3.0 and pos_err < 0.05
    run = tmp_path / "run"
    rows = (run / "IMUState.csv").read_text().strip().splitlines()
    assert rows[0] == "time, px, py, pz, qw, qx, qy, qz, vx, vy, vz" and len(rows) == 101
    # the reference's writer formats (src/VIOWriter.cpp:33-228): one row per vision frame in every file
    headers = {"camera.csv": "time, px, py, pz, qw, qx, qy, qz", "bias.csv": "time, bias_gyr_x, bias_gyr_y, bias_gyr_z, bias_acc_x, bias_acc_y, bias_acc_z",
               "points.csv": "time, p1id, p1x, p1y, p1z, ...", "features.csv": "time, z1id, z1x, z1y, ...", "landmarkError.csv": "time, lm_err_1, lm_err_2, ...",
               "nees.csv": "time, NEES, DoF, PoseNEES, AttitudeNEES",
               "timing.csv": "time, correction, features, preprocessing, propagation, total, total vision update, write output"}
    for name, header in headers.items():
        lines = (run / name).read_text().strip().splitlines()
        assert lines[0] == header and len(lines) == 101, name
    for name in ["trueState.csv", "poseConsistency.csv", "cameraConsistency.csv", "biasConsistency.csv"]:
        assert len((run / name).read_text().strip().splitlines()) == 101, name
    nees = np.array([[float(v) for v in l.split(", ")] for l in (run / "nees.csv").read_text().strip().splitlines()[1:]])
    assert abs(nees[:, 1].mean() - mean_nees) <= 1e-4 * mean_nees and np.all(nees[:, 2] == 21 + 3 * 30) and np.all(nees[:, 3] >= 0) and np.all(nees[:, 4] >= 0)
    last_state = [float(v) for v in rows[-1].split(", ")]
    assert abs(last_state[0] - 4.95) < 1e-12 and abs(np.linalg.norm(last_state[4:8]) - 1.0) < 1e-5
    feat = (run / "features.csv").read_text().strip().splitlines()[-1].split(", ")
    assert (len(feat) - 1) % 3 == 0 and (len(feat) - 1) // 3 == 30
    pose_c = np.array([[float(v) for v in l.split(", ")] for l in (run / "poseConsistency.csv").read_text().strip().splitlines()[1:]])
    assert pose_c.shape == (100, 13) and np.all(pose_c[:, 7:] > 0)  # variances
    # errors within 5 sigma of the filter's own uncertainty almost always (consistency of the whole chain)
    assert np.mean(np.abs(pose_c[10:, 1:7]) <= 5 * np.sqrt(pose_c[10:, 7:])) > 0.95


def _csv(path):
    return [[float(v) for v in l.split(", ")] for l in path.read_text().strip().splitlines()[1:]]


def _qinv(q):
    return np.array([q[0], -q[1], -q[2], -q[3]])


def _angle(q):
    return 2.0 * math.atan2(np.linalg.norm(q[1:4]), abs(q[0]))


@pytest.mark.gpu
def test_eqvio_sim_csv_values(tmp_path):
    """Row f-2, values (not only headers and row counts) of what VIOWriter writes (src/VIOWriter.cpp:33-228):
    (1) the files against each other: points.csv holds WORLD-frame points, so mapped back through IMUState.csv x camera.csv they must
        reproduce landmarkError.csv against trueState.csv; the error columns of bias / pose / cameraConsistency.csv against trueState.csv and the
        state files; nees.csv's PoseNEES / AttitudeNEES against the errors and variances of poseConsistency.csv;
    (2) against the ORACLE's filter on the same simulator stream (same settings and seed; CPU restatement of the reference, nothing of the device on that
        side): state files, points, variances and NEES of every frame to the 6 significant digits the writer prints, and the printed forms themselves
        (src/VIOWriter.cpp:40, include/eqvio/csv/CSVLine.h:108-113, 204-248) as literal lines built from the oracle's state."""
    exe = os.path.join(ROOT, "eqvio_amd", "lib", "eqvio_sim")
    run = tmp_path / "run"
    out = subprocess.run([exe, "--duration", "3", "--maxFeatures", "25", "--numWalls", "4", "--seed", "3", "--coordinateChoice", "InvDepth", "--fastRiccati", "1",
                          "--outputNoise", "--inputNoise", "--measurementNoise", "0.5", "--initialPointVariance", "0.01", "--quiet", "--output", str(run)],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    imu_state, camera, bias, true_state = _csv(run / "IMUState.csv"), _csv(run / "camera.csv"), _csv(run / "bias.csv"), _csv(run / "trueState.csv")
    points, lm_err, nees = _csv(run / "points.csv"), _csv(run / "landmarkError.csv"), _csv(run / "nees.csv")
    pose_c, cam_c, bias_c = _csv(run / "poseConsistency.csv"), _csv(run / "cameraConsistency.csv"), _csv(run / "biasConsistency.csv")
    F = len(imu_state)
    assert F == 60
    PRINT = 2e-5  # the writer prints 6 significant digits

    for f in range(F):
        t = imu_state[f][0]
        assert all(abs(rows[f][0] - t) < 1e-12 for rows in (camera, bias, true_state, points, lm_err, nees, pose_c, cam_c, bias_c))
        # (1a) points.csv -> camera frame -> distance to the true landmark = landmarkError.csv
        x, q = np.array(imu_state[f][1:4]), np.array(imu_state[f][4:8])
        cx, cq = np.array(camera[f][1:4]), np.array(camera[f][4:8])
        PC_q, PC_x = qmul(q, cq), x + qrot(q, cx)
        ts = true_state[f]
        n_true = int(ts[24])
        true_lm = {int(ts[25 + 4 * i]): np.array(ts[26 + 4 * i:29 + 4 * i]) for i in range(n_true)}
        pts = points[f][1:]
        est = {int(pts[4 * i]): qrot(_qinv(PC_q), np.array(pts[4 * i + 1:4 * i + 4]) - PC_x) for i in range(len(pts) // 4)}
        assert len(est) > 0 and len(lm_err[f]) - 1 == n_true
        for col, lid in enumerate(true_lm):
            e = lm_err[f][1 + col]
            if lid in est:
                assert abs(np.linalg.norm(est[lid] - true_lm[lid]) - e) <= PRINT * max(1.0, np.linalg.norm(true_lm[lid])), (f, lid)
            else:
                assert math.isnan(e)
        # (1b) bias error = true bias - estimated bias; the rotation parts of the pose / camera errors are conjugates of the attitude differences
        tb = np.array(ts[18:24])
        np.testing.assert_allclose(bias_c[f][1:7], tb - np.array(bias[f][1:7]), rtol=0, atol=PRINT)
        tq, tcq = np.array(ts[4:8]), np.array(ts[14:18])
        assert abs(np.linalg.norm(pose_c[f][1:4]) - _angle(qmul(tq, _qinv(q)))) <= PRINT
        assert abs(np.linalg.norm(cam_c[f][1:4]) - _angle(qmul(tcq, _qinv(cq)))) <= PRINT
        # (1c) NEES of a sub-vector is bounded below by every e_i^2 / Sigma_ii and the pose NEES by the attitude NEES
        e6, v6 = np.array(pose_c[f][1:7]), np.array(pose_c[f][7:13])
        assert np.all(v6 > 0) and np.all(np.array(cam_c[f][7:13]) > 0) and np.all(np.array(bias_c[f][7:13]) > 0)
        pose_nees, att_nees = nees[f][3], nees[f][4]
        assert pose_nees >= np.max(e6**2 / v6) * (1 - 1e-4) and att_nees >= np.max(e6[:3]**2 / v6[:3]) * (1 - 1e-4) and pose_nees >= att_nees * (1 - 1e-4)

    # (2) the same scenario through the ORACLE's filter (CPU restatement of the reference; no device on this side): the same C++ simulator settings and seed give the
    #     same IMU / feature stream, the oracle filter is driven main_sim-style, and every number the executable printed is compared with the oracle's state, Sigma
    #     diagonal and NEES at the writer's print precision
    fs = Settings.defaults()
    fs.coordinateChoice, fs.fastRiccati, fs.measurementNoise, fs.initialPointVariance = COORD_INVDEPTH, 1, 0.5, 0.01
    srv, fs = make_server(fs, duration=3.0, maxFeatures=25, numWalls=4, randomSeed=3, outputNoise=1, inputNoise=1)
    s0, ids0, p0 = srv.true_state(0.0, True)
    orc = OracleFilter(fs, s0, ids0, p0, 0.0)
    text = {name: (run / name).read_text().strip().splitlines()[1:] for name in ("IMUState.csv", "camera.csv", "bias.csv", "points.csv")}
    frame = [0]

    def same_print(tok, value, is_int=False):
        """tok is what a default stream (6 significant digits, %g) prints for `value` - or for a double within 1e-9 relative of it (one unit of the last printed digit)"""
        if is_int:
            return tok == str(int(value))
        if tok == "%g" % value:
            return True
        return abs(float(tok) - value) <= 1.01e-6 * max(abs(value), 1e-300) + 1e-300 and tok == "%g" % float(tok)

    def on_frame(stamp):
        f = frame[0]
        frame[0] += 1
        s, ids, p = orc.state_estimate()
        assert abs(imu_state[f][0] - orc.get_time()) < 1e-12
        np.testing.assert_allclose(imu_state[f][1:], np.concatenate([s[10:13], s[6:10], s[13:16]]), rtol=PRINT, atol=PRINT)
        np.testing.assert_allclose(camera[f][1:], np.concatenate([s[20:23], s[16:20]]), rtol=PRINT, atol=PRINT)
        np.testing.assert_allclose(bias[f][1:], s[0:6], rtol=PRINT, atol=PRINT)
        PC_q, PC_x = qmul(s[6:10], s[16:20]), s[10:13] + qrot(s[6:10], s[20:23])
        pts = points[f][1:]
        assert [int(pts[4 * i]) for i in range(len(pts) // 4)] == ids.tolist()
        world = np.array([qrot(PC_q, q) + PC_x for q in p])
        np.testing.assert_allclose(np.array(pts).reshape(-1, 4)[:, 1:], world, rtol=PRINT, atol=PRINT)
        # the printed forms (src/VIOWriter.cpp:40: stamp at setprecision(20), then setprecision(6); include/eqvio/csv/CSVLine.h:108-113: every value through a
        # fresh default stringstream = %g; :217-219 quaternions as w, x, y, z; :247 SE(3) as position then attitude; points: id, then PC * q.p): the literal line
        # built from the ORACLE's state by those rules, token by token
        st = "%.20g" % orc.get_time()
        expect = {"IMUState.csv": [s[10], s[11], s[12], s[6], s[7], s[8], s[9], s[13], s[14], s[15]],
                  "camera.csv": [s[20], s[21], s[22], s[16], s[17], s[18], s[19]],
                  "bias.csv": list(s[0:6])}
        for name, vals in expect.items():
            toks = text[name][f].split(", ")
            assert toks[0] == st and float(toks[0]) == orc.get_time(), (name, f, toks[0], st)  # 20 significant digits: the double round-trips
            assert len(toks) == 1 + len(vals) and all(same_print(t, v) for t, v in zip(toks[1:], vals)), (name, f, toks, vals)
        toks = text["points.csv"][f].split(", ")
        assert toks[0] == st and len(toks) == 1 + 4 * len(ids)
        for i, lid in enumerate(ids.tolist()):
            assert same_print(toks[1 + 4 * i], lid, True) and all(same_print(toks[2 + 4 * i + k], world[i][k]) for k in range(3)), (f, lid)
        S = orc.get_sigma()
        d = np.diag(S)
        np.testing.assert_allclose(pose_c[f][7:13], d[6:12], rtol=PRINT)
        np.testing.assert_allclose(cam_c[f][7:13], d[15:21], rtol=PRINT)
        np.testing.assert_allclose(bias_c[f][7:13], d[0:6], rtol=PRINT)
        e6 = np.array(pose_c[f][1:7])
        assert abs(nees[f][3] - e6 @ np.linalg.solve(S[6:12, 6:12], e6)) <= 1e-3 * max(nees[f][3], 1e-12)  # e6 itself is rounded to 6 digits
        ts, tids, tp = srv.true_state(orc.get_time())
        assert abs(nees[f][1] - orc.compute_nees(ts, tids, tp)) <= PRINT * nees[f][1] + 1e-20 and nees[f][2] == S.shape[0]  # (frame 0 starts at the truth: NEES ~ 1e-30, rounding only)

    drive(srv, [orc], F, on_frame)
    assert frame[0] == F
    # every value token of every file the writer produced is in the default-stream form (6 significant digits), every stamp at 20
    for name in ("IMUState.csv", "camera.csv", "bias.csv", "points.csv", "features.csv", "landmarkError.csv", "nees.csv", "poseConsistency.csv", "cameraConsistency.csv",
                 "biasConsistency.csv", "trueState.csv"):
        for line in (run / name).read_text().strip().splitlines()[1:]:
            toks = line.split(", ")
            assert toks[0] == "%.20g" % float(toks[0]), (name, toks[0])
            for t in toks[1:]:
                assert t in ("nan", "-nan") or t == "%g" % float(t) or t == str(int(float(t))), (name, t)
