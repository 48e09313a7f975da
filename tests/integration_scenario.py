"""Scenario files for tests/integration/run_filter_frames (the reference-side VIOFilter binding's driver): settings, camera, initial condition and the
frames of a synthetic world as one binary file; readers for what the driver writes back. Shared by tests/test_integration_filter.py and bench.py."""
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


def write_scenario(path, settings, cam, sensor, ids, p, t0, frames):
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
        for imus, stamp, mid, y in frames:
            f.write(struct.pack("<i", len(imus)))
            f.write(np.asarray(imus, np.float64).tobytes())
            f.write(struct.pack("<d", stamp))
            f.write(struct.pack("<i", len(mid)))
            f.write(np.asarray(mid, np.int32).tobytes())
            f.write(np.asarray(y, np.float64).tobytes())


def run_driver(scenario, out, fused, state_every=0, sigma_every=0, warm=0, timeout=1200):
    res = subprocess.run([EXE, scenario, out, str(int(fused)), str(state_every), str(sigma_every), str(warm)], check=True, capture_output=True, text=True, timeout=timeout)
    tok = res.stdout.split()
    return {"frames": int(tok[1]), "seconds": float(tok[3]), "updates_per_s": float(tok[5])}


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
