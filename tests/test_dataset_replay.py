"""Dataset ingestion + feature-track replay (eqvio_amd/host/DatasetReplay.*, main_opt.cpp), SURVEY.md §8 row f-3.

CPU: the ASL / UZH-FPV text layouts the reference reads (ASLDatasetReader.cpp:43-53, 104-130; UZHFPVDatasetReader.cpp:48-58,
117-139) and the stamp-ordered merge of SimpleDataServer.cpp:20-30, through `eqvio_opt --dumpMeasurements` (no device).
GPU: a simulated run written out by eqvio_sim and replayed through eqvio_opt (the main_opt loop) on the device filter."""
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


@pytest.mark.gpu
def test_simulated_run_replayed_through_eqvio_opt(built, tmp_path):
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
    # the replayed filter starts at the origin with gravity-aligned attitude (VIOFilter.cpp:65-78): compare the distance
    # travelled with the ground truth's (wave trajectory: from (1,0,0) a quarter-plus turn of the unit circle)
    est = np.array([float(m.group(k)) for k in (2, 3, 4)])
    gt_end = np.array([float(g.group(k)) for k in (3, 4, 5)])
    gt0 = np.array([float(v) for v in open(ds + "/groundtruth.csv").read().splitlines()[1].split(",")[1:4]])
    travelled = np.linalg.norm(gt_end - gt0)
    assert travelled > 1.0 and abs(np.linalg.norm(est) - travelled) < 0.35 * travelled
    rows = open(str(tmp_path / "replay") + "/IMUState.csv").read().strip().splitlines()
    # header + one row per frame; the frame at t = 0 precedes the first IMU sample, so the filter is still uninitialised
    # there and the row carries getTime() = -1 with the identity state, exactly as main_opt.cpp:225-229 would write it
    assert len(rows) == 121 and rows[1].startswith("-1, 0, 0, 0, 1, 0, 0, 0")
