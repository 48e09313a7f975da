"""Scenario files for tests/integration/run_filter_frames (the replay driver over the reference-side VIO_eqf binding): settings, camera, initial condition, the
frames of a synthetic world and the PLAN of every frame (clipped IMU intervals, lost / rejected / new landmarks) as one binary file; readers for what the driver
writes back. Shared by tests/test_integration_filter.py and bench.py (which uses plan_without_decisions only: the oracle has no part in a bench leg)."""
import os
import struct
import subprocess

import numpy as np

from eqvio_amd.capi import _SETTINGS_DOUBLES, _SETTINGS_INTS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD_DIR = os.path.join(ROOT, "tests", "integration")
EXE = os.path.join(BUILD_DIR, "_build", "run_filter_frames")


def build_driver():
    subprocess.run(["make", "-s", "-C", BUILD_DIR], check=True)
    return EXE


def clip_intervals(frames, t0):
    """The IMU bookkeeping of VIOFilter::integrateUpToTime (src/VIOFilter.cpp:134-192) for a whole run, on the host of the TEST: per frame the buffered samples
    and the part of each sample's interval that falls into (t_previous frame, t_frame] (a sample holds until the next one's stamp; the last one until the frame's),
    and what stays in the buffer for the next frame (everything from the last sample at or before the frame's stamp)."""
    buf, t_cur, out = [], float(t0), []
    for imus, stamp, _, _ in frames:
        buf += [np.asarray(u, np.float64) for u in imus]
        dts = []
        for i, u in enumerate(buf):
            lo = max(u[0], t_cur)
            hi = min(buf[i + 1][0], stamp) if i + 1 < len(buf) else stamp
            dts.append(max(hi - lo, 0.0))
        out.append((list(buf), dts))
        t_cur = float(stamp)
        first = next((i for i, u in enumerate(buf) if u[0] >= t_cur), len(buf))
        if first > 0:
            buf = buf[first - 1:]
    return out


def plan_without_decisions(frames, t0):
    """The plan of a run in which the landmark set never changes (bench.build_workload's hover world with the outlier thresholds off): nothing is lost, nothing
    rejected, nothing new - no filter has to be consulted."""
    return [dict(samples=sm, dts=dt, lost=[], outliers=[], new_ids=[], new_p=np.zeros((0, 3))) for sm, dt in clip_intervals(frames, t0)]


def plan_from_oracle(settings, cam, sensor, ids, p, t0, frames):
    """The plan of a run with landmark turnover and outliers: which landmarks the reference's VIOFilter drops as lost (src/VIOFilter.cpp:280-302), which its
    removeOutliers rejects (:304-364) and which it adds (:258-278) in every frame, read off the ORACLE's filter (the landmark ids before and after each frame).
    Returns (plan, oracle state estimate per frame, oracle Sigma per frame)."""
    from oracle_binding import OracleFilter, oracle_cam_undistort

    assert not settings.useMedianDepth, "the plan computes the initial points of new landmarks with the fixed initialSceneDepth"
    orc = OracleFilter(settings, sensor, ids, p, t0)
    plan, states, sigmas = [], [], []
    for (imus, stamp, mid, y), (sm, dt) in zip(frames, clip_intervals(frames, t0)):
        before = [int(i) for i in orc.state_estimate()[1]]
        for u in imus:
            orc.process_imu(u)
        orc.process_vision(stamp, cam, mid, y)
        st = orc.state_estimate()
        after = {int(i) for i in st[1]}
        measured = {int(i): np.asarray(y, np.float64).reshape(-1, 2)[k] for k, i in enumerate(mid)}
        lost = [i for i in reversed(before) if i not in measured] if settings.removeLostLandmarks else []
        outliers = [i for i in before if i in measured and i not in after]
        new_ids = [i for i in measured if i not in before and i in after]
        new_p = np.array([oracle_cam_undistort(cam, measured[i]) * settings.initialSceneDepth for i in new_ids]).reshape(-1, 3)
        plan.append(dict(samples=sm, dts=dt, lost=lost, outliers=outliers, new_ids=new_ids, new_p=new_p))
        states.append(st)
        sigmas.append(orc.get_sigma())
    return plan, states, sigmas


def write_scenario(path, settings, cam, sensor, ids, p, t0, frames, plan):
    with open(path, "wb") as f:
        f.write(struct.pack("<26d", *[getattr(settings, n) for n in _SETTINGS_DOUBLES]))
        f.write(struct.pack("<9i", *[getattr(settings, n) for n in _SETTINGS_INTS]))
        f.write(struct.pack("<7d", *list(settings.cameraOffset)))
        f.write(struct.pack("<6d", cam.fx, cam.fy, cam.cx, cam.cy, cam.width, cam.height))
        f.write(np.asarray(sensor, np.float64).tobytes())
        f.write(struct.pack("<i", len(ids)))
        f.write(np.asarray(ids, np.int32).tobytes())
        f.write(np.asarray(p, np.float64).tobytes())
        f.write(struct.pack("<d", t0))
        f.write(struct.pack("<i", len(frames)))
        for (imus, stamp, mid, y), pl in zip(frames, plan):
            f.write(struct.pack("<i", len(pl["samples"])))
            for u, dt in zip(pl["samples"], pl["dts"]):
                f.write(np.asarray(u, np.float64).tobytes())
                f.write(struct.pack("<d", dt))
            f.write(struct.pack("<d", stamp))
            f.write(struct.pack("<i", len(mid)))
            f.write(np.asarray(mid, np.int32).tobytes())
            f.write(np.asarray(y, np.float64).tobytes())
            for key in ("lost", "outliers", "new_ids"):
                f.write(struct.pack("<i", len(pl[key])))
                f.write(np.asarray(pl[key], np.int32).tobytes())
            f.write(np.asarray(pl["new_p"], np.float64).tobytes())


def run_driver(scenario, out, fused, state_every=0, sigma_every=0, warm=0, timeout=1200):
    res = subprocess.run([EXE, scenario, out, str(int(fused)), str(state_every), str(sigma_every), str(warm)], check=True, capture_output=True, text=True, timeout=timeout)
    tok = res.stdout.split()
    out = {"frames": int(tok[1]), "seconds": float(tok[3]), "updates_per_s": float(tok[5])}
    if len(tok) >= 8:  # time inside the reference's own dense gain-matrix constructors (caller side of the member-for-member sequence)
        out["gain_matrix_seconds"] = float(tok[7])
    return out


def read_records(path):
    """-> {frame: (sensor23, ids, p)}, {frame: Sigma}"""
    states, sigmas = {}, {}
    buf = open(path, "rb").read()
    pos = 0
    while pos < len(buf):
        kind, frame, n = struct.unpack_from("<3i", buf, pos)
        pos += 12
        if kind == 1:
            sensor = np.frombuffer(buf, np.float64, 23, pos)
            pos += 23 * 8
            ids = np.frombuffer(buf, np.int32, n, pos)
            pos += 4 * n
            p = np.frombuffer(buf, np.float64, 3 * n, pos).reshape(n, 3)
            pos += 24 * n
            states[frame] = (sensor.copy(), ids.copy(), p.copy())
        else:
            sigmas[frame] = np.frombuffer(buf, np.float64, n * n, pos).reshape(n, n, order="F").copy()
            pos += 8 * n * n
    return states, sigmas
