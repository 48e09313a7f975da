"""Dataset ingestion + feature-track replay (eqvio_amd/host/DatasetReplay.*, main_opt.cpp), SURVEY.md §8 row f-3.

CPU: the ASL / UZH-FPV text layouts the reference reads (ASLDatasetReader.cpp:43-53, 104-130; UZHFPVDatasetReader.cpp:48-58,
117-139) and the stamp-ordered merge of SimpleDataServer.cpp:20-30, through `eqvio_opt --dumpMeasurements` (no device).
CPU also: the datasets' camera files (sensor.yaml / camchain yaml: intrinsics, distortion, camera offset; ASLDatasetReader.cpp:76-101, UZHFPVDatasetReader.cpp:78-115).
GPU: simulated runs written out by eqvio_sim, converted to both dataset layouts with a radial-tangential / an equidistant camera, and replayed through eqvio_opt (the
main_opt loop) on the device filter, the state after EVERY frame against the oracle's filter fed with the same parsed measurements (1e-9)."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OPT = os.path.join(ROOT, "eqvio_amd", "lib", "eqvio_opt")
SIM = os.path.join(ROOT, "eqvio_amd", "lib", "eqvio_sim")


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g

    if not (os.path.exists(OPT) and os.path.exists(SIM)):
        g.build()
    return True


def dump(args):
    out = subprocess.run([OPT, *args, "--dumpMeasurements"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    return [l.split() for l in out.stdout.strip().splitlines()]


def test_asl_layout_and_merge_order(built, tmp_path):
    (tmp_path / "imu.csv").write_text(
        "#timestamp [ns],w_RS_S_x [rad s^-1],w_RS_S_y,w_RS_S_z,a_RS_S_x [m s^-2],a_RS_S_y,a_RS_S_z\r\n"
        "1403715273262142976,-0.099134701513277898,0.14730578886832138,0.02722713633111154,8.1476917083333333,-0.37592158333333331,-2.4026292499999999\r\n"
        "1403715273267142912,-0.099134701513277898,0.14032447186034408,0.029321531433504733,8.033280791666666,-0.40861041666666664,-2.4026292499999999\r\n"
        "1403715273272143104,-0.098436569812480182,0.12775810124598493,0.037699111843077518,7.8861810416666662,-0.42495483333333328,-2.4353180833333332\r\n")
    (tmp_path / "features.csv").write_text("time, z1id, z1x, z1y, ...\n1403715273.262142976, 4, 100.5, 200.25, 9, 300, 50\n1403715273.2671, 9, 301, 51\n1403715273.3, \n")
    (tmp_path / "gt.csv").write_text("#timestamp, p_RS_R_x [m], p_RS_R_y [m], p_RS_R_z [m], q_RS_w [], q_RS_x [], q_RS_y [], q_RS_z [], v...\n"
                                     "1403715273262142976,4.688,-1.786,0.783,0.534,-0.153,-0.827,-0.082,0,0,0\n"
                                     "1403715273262142976,9,9,9,1,0,0,0,0,0,0\n"  # duplicate stamp: dropped (ASLDatasetReader.cpp:120-125)
                                     "1403715273267142912,4.688,-1.786,0.783,0.534,-0.153,-0.827,-0.082,0,0,0\n")
    rows = dump(["--imu", str(tmp_path / "imu.csv"), "--features", str(tmp_path / "features.csv"), "--groundtruth", str(tmp_path / "gt.csv")])
    kinds = [r[0] for r in rows]
    # image first when stamps tie (SimpleDataServer.cpp:22), then strictly by stamp; the empty measurement is still served
    assert kinds == ["IMG", "IMU", "IMG", "IMU", "IMU", "IMG", "GT"]
    assert float(rows[1][1]) == 1403715273262142976 * 1e-9 and float(rows[1][2]) == -0.099134701513277898 and float(rows[1][7]) == -2.4026292499999999
    assert rows[0][2:] == ["2", "4", "100.5", "200.25", "9", "300", "50"] and rows[5][2] == "0"
    assert rows[6][1] == "2" and abs(float(rows[6][2]) - 1403715273.262143) < 1e-6
    q = np.array([float(v) for v in rows[6][6:10]])
    assert abs(np.linalg.norm(q) - 1) < 1e-12 and np.allclose(q, np.array([0.534, -0.153, -0.827, -0.082]) / np.linalg.norm([0.534, -0.153, -0.827, -0.082]))


def test_uzhfpv_layout(built, tmp_path):
    """Space separated, header line, leading index column, stamps in seconds (UZHFPVDatasetReader.cpp:48-58)."""
    (tmp_path / "imu.txt").write_text("# id timestamp ang_vel_x ang_vel_y ang_vel_z lin_acc_x lin_acc_y lin_acc_z\n"
                                      "0 1540821838.89 0.01 -0.02 0.03 9.7 0.1 -0.2\n1  1540821838.892 0.011 -0.021 0.031 9.71 0.11 -0.21\n")
    (tmp_path / "features.csv").write_text("time, z1id, z1x, z1y, ...\n1540821838.891, 7, 320, 240\n")
    (tmp_path / "groundtruth.txt").write_text("# timestamp tx ty tz qw qx qy qz\n1540821838.8 1 2 3 1 0 0 0\n1540821838.9 1.1 2 3 1 0 0 0\n")
    rows = dump(["--imu", str(tmp_path / "imu.txt"), "--features", str(tmp_path / "features.csv"), "--format", "uzhfpv", "--groundtruth", str(tmp_path / "groundtruth.txt")])
    assert [r[0] for r in rows] == ["IMU", "IMG", "IMU", "GT"]
    assert [float(v) for v in rows[0][1:]] == [1540821838.89, 0.01, -0.02, 0.03, 9.7, 0.1, -0.2]
    assert [float(v) for v in rows[2][1:]] == [1540821838.892, 0.011, -0.021, 0.031, 9.71, 0.11, -0.21]
    assert rows[3][1] == "2" and [float(v) for v in rows[3][2:6]] == [1540821838.8, 1, 2, 3]


def test_bad_input_is_reported(built, tmp_path):
    (tmp_path / "imu.csv").write_text("#h\n1,2,3\n")
    (tmp_path / "features.csv").write_text("time\n")
    out = subprocess.run([OPT, "--imu", str(tmp_path / "imu.csv"), "--features", str(tmp_path / "features.csv"), "--dumpMeasurements"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 1 and "short IMU line" in out.stderr
    out = subprocess.run([OPT, "--imu", str(tmp_path / "nope.csv"), "--features", str(tmp_path / "features.csv")], capture_output=True, text=True, timeout=60)
    assert out.returncode == 1 and "cannot open" in out.stderr


def _pose_matrix(q7):
    """(qw qx qy qz x y z) -> homogeneous 4 x 4"""
    w, x, y, z = q7[:4] / np.linalg.norm(q7[:4])
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, q7[4:7]
    return T


def _write_camera_file(path, layout, cam, offset7):
    """The dataset's own camera file in the layout the reference reads (ASLDatasetReader.cpp:76-101 sensor.yaml / UZHFPVDatasetReader.cpp:78-115 camchain yaml)."""
    T = _pose_matrix(np.asarray(offset7, float))
    if layout == "asl":
        rows = ",\n         ".join(", ".join(repr(float(v)) for v in T[r]) for r in range(4))
        open(path, "w").write(
            "# General sensor definitions.\nsensor_type: camera\ncomment: VI-Sensor cam0 (MT9M034)\n\n# Sensor extrinsics wrt. the body-frame.\nT_BS:\n  cols: 4\n  rows: 4\n"
            f"  data: [{rows}]\n\n# Camera specific definitions.\nrate_hz: 20\nresolution: [{cam.width}, {cam.height}]\ncamera_model: pinhole\n"
            f"intrinsics: [{cam.fx!r}, {cam.fy!r}, {cam.cx!r}, {cam.cy!r}] #fu, fv, cu, cv\ndistortion_model: radial-tangential\n"
            f"distortion_coefficients: [{cam.dist[0]!r}, {cam.dist[1]!r}, {cam.dist[2]!r}, {cam.dist[3]!r}]\n")
    else:
        Ti = np.linalg.inv(T)  # the file holds the pose of the IMU w.r.t. the camera
        rows = "\n".join("  - [" + ", ".join(repr(float(v)) for v in Ti[r]) + "]" for r in range(4))
        open(path, "w").write(
            f"cam0:\n  T_cam_imu:\n{rows}\n  cam_overlaps: []\n  camera_model: pinhole\n  distortion_coeffs: [{cam.dist[0]!r}, {cam.dist[1]!r}, {cam.dist[2]!r}, {cam.dist[3]!r}]\n"
            f"  distortion_model: equidistant\n  intrinsics: [{cam.fx!r}, {cam.fy!r}, {cam.cx!r}, {cam.cy!r}]\n  resolution: [{cam.width}, {cam.height}]\n  rostopic: /snappy_cam/stereo_l\n"
            "  timeshift_cam_imu: -0.01\n")


def test_camera_files_are_read_like_the_reference_does(built, tmp_path):
    """sensor.yaml (EuRoC) and the camchain yaml (UZH-FPV): intrinsics, distortion and the camera offset as main_opt.cpp:114-147 uses them - T_BS as it is, T_cam_imu
    inverted - checked through the offset that eqvio_opt puts into the state it dumps (no device needed: no frame is processed)."""
    from util import euroc_radtan_camera, uzhfpv_equidistant_camera

    (tmp_path / "imu.csv").write_text("#h\n")
    (tmp_path / "features.csv").write_text("time\n")
    offset = np.array([0.7071, 0.0, 0.7071, 0.0, 0.05, -0.02, 0.1])
    for layout, cam in (("asl", euroc_radtan_camera()), ("uzhfpv", uzhfpv_equidistant_camera())):
        camfile = str(tmp_path / f"cam_{layout}.yaml")
        _write_camera_file(camfile, layout, cam, offset)
        out = subprocess.run([OPT, "--imu", str(tmp_path / "imu.csv"), "--features", str(tmp_path / "features.csv"), "--format", layout, "--cameraFile", camfile, "--printCamera"],
                             capture_output=True, text=True, timeout=60)
        assert out.returncode == 0, out.stderr
        v = [float(t) for t in out.stdout.split()[1:]]
        assert int(v[0]) == cam.model and np.allclose(v[1:7], [cam.fx, cam.fy, cam.cx, cam.cy, cam.width, cam.height], rtol=0, atol=0)
        assert np.allclose(v[7:11], list(cam.dist)[:4], rtol=0, atol=0)
        q = offset[:4] / np.linalg.norm(offset[:4])
        assert np.allclose(v[12:16], q, atol=1e-12) and np.allclose(v[16:19], offset[4:], atol=1e-12)
    bad = tmp_path / "bad.yaml"
    bad.write_text("resolution: [752, 480]\nintrinsics: [1, 2, 3]\n")
    out = subprocess.run([OPT, "--imu", str(tmp_path / "imu.csv"), "--features", str(tmp_path / "features.csv"), "--cameraFile", str(bad), "--printCamera"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 1 and "intrinsics" in out.stderr


def _replay_case(tmp_path, layout, cam):
    """A simulated run written out by eqvio_sim (IMU text file + feature tracks), its tracks re-projected through a DISTORTED camera and its files converted to the given
    dataset layout; returns what eqvio_opt needs to replay it and what the oracle needs to follow it."""
    from util import euroc_camera

    run, ds = str(tmp_path / "run"), str(tmp_path / "ds")
    sim_flags = ["--coordinateChoice", "InvDepth", "--fastRiccati", "1", "--measurementNoise", "0.5", "--initialPointVariance", "1.0", "--useMedianDepth", "0", "--initialSceneDepth", "3.0"]
    sim = subprocess.run([SIM, "--duration", "4", "--maxFeatures", "40", "--numWalls", "4", "--seed", "2", "--quiet", "--output", run, "--writeDataset", ds, *sim_flags],
                         capture_output=True, text=True, timeout=120)
    assert sim.returncode == 0, sim.stderr
    pin = euroc_camera()  # the simulator's camera (SimulationDataServer.cpp:162-176): its pixels become bearings, the bearings pixels of the distorted camera
    feats = str(tmp_path / f"features_{layout}.csv")
    with open(feats, "w") as f:
        lines = open(run + "/features.csv").read().splitlines()
        f.write(lines[0] + "\n")
        for ln in lines[1:]:
            c = [t.strip() for t in ln.split(",") if t.strip() != ""]
            out = [c[0]]
            for k in range(1, len(c), 3):
                b = np.array([(float(c[k + 1]) - pin.cx) / pin.fx, (float(c[k + 2]) - pin.cy) / pin.fy, 1.0])
                y = cam.project(b)
                out += [c[k], repr(float(y[0])), repr(float(y[1]))]
            f.write(", ".join(out) + "\n")
    imu = ds + "/imu.csv"
    if layout == "uzhfpv":  # index column, blanks, stamps in seconds (UZHFPVDatasetReader.cpp:48-58)
        imu = str(tmp_path / "imu.txt")
        with open(imu, "w") as f:
            f.write("# id timestamp ang_vel_x ang_vel_y ang_vel_z lin_acc_x lin_acc_y lin_acc_z\n")
            for k, ln in enumerate(open(ds + "/imu.csv").read().splitlines()[1:]):
                c = ln.split(",")
                f.write(" ".join([str(k), repr(int(c[0]) * 1e-9)] + c[1:7]) + "\n")
    return imu, feats, sim_flags


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["asl", "uzhfpv"])
def test_replayed_dataset_follows_the_oracle_frame_by_frame(built, tmp_path, layout):
    """SURVEY.md section 8 row f-3 with the oracle in the loop: the SAME replayed files - IMU text file in the dataset's layout, feature tracks, the dataset's own
    camera file (radial-tangential for ASL / EuRoC, equidistant for UZH-FPV) - go through eqvio_opt on the device and, parsed measurement by measurement (eqvio_opt
    --dumpMeasurements), through the oracle's VIOFilter, which starts uninitialised and sets its attitude from the first IMU sample like main_opt's does (VIOFilter.cpp:65-78).
    The state estimate after EVERY vision measurement (--dumpStates, full precision) must agree to 1e-9: pose, velocity, biases, camera offset, every landmark; the landmark
    sets must coincide (lost landmarks dropped, new ones added in step)."""
    from eqvio_amd.capi import COORD_INVDEPTH, Settings
    from oracle_binding import OracleFilter, se3_log_dist
    from util import euroc_radtan_camera, uzhfpv_equidistant_camera

    cam = euroc_radtan_camera() if layout == "asl" else uzhfpv_equidistant_camera()
    offset = np.array([0.5, -0.5, 0.5, -0.5, 0.02, -0.01, 0.03])
    imu, feats, flags = _replay_case(tmp_path, layout, cam)
    camfile = str(tmp_path / "camera.yaml")
    _write_camera_file(camfile, layout, cam, offset)
    states = str(tmp_path / "states.txt")
    common = ["--imu", imu, "--features", feats, "--format", layout, "--cameraFile", camfile]
    out = subprocess.run([OPT, *common, "--dumpStates", states, *flags], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    meas = subprocess.run([OPT, *common, "--dumpMeasurements"], capture_output=True, text=True, timeout=60)
    assert meas.returncode == 0, meas.stderr
    s = Settings.defaults()
    s.coordinateChoice, s.fastRiccati, s.measurementNoise, s.initialPointVariance, s.useMedianDepth, s.initialSceneDepth = COORD_INVDEPTH, 1, 0.5, 1.0, 0, 3.0
    s.cameraOffset[:] = list(offset)
    orc = OracleFilter(s)
    dev = [ln.split() for ln in open(states).read().splitlines()]
    frame, worst, sizes = 0, 0.0, set()
    for tok in (ln.split() for ln in meas.stdout.splitlines()):
        if tok[0] == "IMU":
            orc.process_imu(np.array([float(v) for v in tok[1:8]] + [0.0] * 6))
            continue
        assert tok[0] == "IMG"
        M = int(tok[2])
        ids = np.array([int(tok[3 + 3 * k]) for k in range(M)], np.int32)
        y = np.array([[float(tok[4 + 3 * k]), float(tok[5 + 3 * k])] for k in range(M)]).reshape(-1)
        orc.process_vision(float(tok[1]), cam, ids, y)
        d = dev[frame]
        frame += 1
        if float(d[0]) < 0:  # the frame before the first IMU sample: main_opt writes the uninitialised filter (getTime() = -1)
            continue
        s_o, ids_o, p_o = orc.state_estimate()
        s_g = np.array([float(v) for v in d[1:24]])
        N = int(d[24])
        ids_g = np.array([int(d[25 + 4 * k]) for k in range(N)])
        p_g = np.array([[float(d[26 + 4 * k + j]) for j in range(3)] for k in range(N)]).reshape(-1, 3)
        assert float(d[0]) == orc.get_time() and np.array_equal(ids_g, ids_o), frame
        e = max(se3_log_dist(s_g[6:13], s_o[6:13]) / max(1.0, np.linalg.norm(s_o[10:13])), se3_log_dist(s_g[16:23], s_o[16:23]), np.max(np.abs(s_g[13:16] - s_o[13:16])),
                np.max(np.abs(s_g[0:6] - s_o[0:6])))
        if N:
            e = max(e, np.max(np.linalg.norm(p_g - p_o, axis=1) / np.maximum(1.0, np.linalg.norm(p_o, axis=1))))
        assert e <= 1e-9, (frame, e)
        worst = max(worst, e)
        sizes |= set(ids_g.tolist())
    assert frame == len(dev) >= 75 and len(sizes) >= 50, (frame, len(sizes))  # 40 tracked at a time: landmarks entered and left
    print(f"{layout} replay, {frame} frames: worst deviation of the device state from the oracle {worst:.1e}")


@pytest.mark.gpu
def test_simulated_run_replayed_through_eqvio_opt(built, tmp_path):
    """The main_opt loop end to end: measurement counters, the ground-truth file, one writer row per frame."""
    run, ds = str(tmp_path / "run"), str(tmp_path / "ds")
    common = ["--coordinateChoice", "InvDepth", "--fastRiccati", "1", "--measurementNoise", "0.5", "--initialPointVariance", "1.0", "--useMedianDepth", "0",
              "--initialSceneDepth", "3.0"]
    sim = subprocess.run([SIM, "--duration", "6", "--maxFeatures", "40", "--numWalls", "4", "--seed", "2", "--quiet", "--output", run, "--writeDataset", ds, *common],
                         capture_output=True, text=True, timeout=120)
    assert sim.returncode == 0, sim.stderr
    out = subprocess.run([OPT, "--imu", ds + "/imu.csv", "--features", run + "/features.csv", "--groundtruth", ds + "/groundtruth.csv", "--cameraOffset", "0.5", "-0.5", "0.5",
                          "-0.5", "0", "0", "0", "--output", str(tmp_path / "replay"), *common], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "Processed 1200 IMU and 120 vision measurements." in out.stdout
    m = re.search(r"final time ([0-9.eE+-]+)\s+position ([0-9.eE+-]+) ([0-9.eE+-]+) ([0-9.eE+-]+)\s+landmarks (\d+)", out.stdout)
    g = re.search(r"groundtruth poses (\d+)\s+nearest stamp ([0-9.eE+-]+)\s+position ([0-9.eE+-]+) ([0-9.eE+-]+) ([0-9.eE+-]+)", out.stdout)
    assert m and g, out.stdout
    assert abs(float(m.group(1)) - 5.95) < 1e-9 and int(m.group(5)) > 10 and int(g.group(1)) == 1200
    # (what the replayed filter computes is compared with the oracle frame by frame in test_replayed_dataset_follows_the_oracle_frame_by_frame above; here: the
    # main_opt loop itself - counters, the ground-truth reader, the writer)
    rows = open(str(tmp_path / "replay") + "/IMUState.csv").read().strip().splitlines()
    # header + one row per frame; the frame at t = 0 precedes the first IMU sample, so the filter is still uninitialised
    # there and the row carries getTime() = -1 with the identity state, exactly as main_opt.cpp:225-229 would write it
    assert len(rows) == 121 and rows[1].startswith("-1, 0, 0, 0, 1, 0, 0, 0")
