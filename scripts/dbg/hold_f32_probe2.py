import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from eqvio_amd.capi import OPT_SIGMA_FP32, OPT_HOLD_NEW_LANDMARKS, SimSettings, SimulationDataServer, VIOFilter
from test_gpu_fp32_sigma import uzh_like_settings
fs = uzh_like_settings(); mf = 200
def run(mode, extra):
    sim = SimSettings.defaults(duration=1.0, trajectory="sine", numPoints=12000, wallDistance=3.0, numWalls=6, randomSeed=5, maxFeatures=mf, imuFreq=500.0, imageFreq=30.0, outputNoise=1)
    srv = SimulationDataServer(sim, fs)
    fs.cameraOffset[:] = srv.camera_offset()
    s0, ids0, p0 = srv.true_state(0.0, True)
    fl = []
    for hold in (1, 0):
        f = VIOFilter(fs, max_landmarks=2 * mf + 64, sensor=s0, ids=ids0[:0], p=p0[:0], time=0.0)
        f.set_core_option(OPT_SIGMA_FP32, mode); f.set_core_option(OPT_HOLD_NEW_LANDMARKS, hold)
        for o, v in extra: f.set_core_option(o, v)
        fl.append(f)
    frames = 0; res = []
    while srv.next_measurement_type() != srv.NONE and frames < 3:
        if srv.next_measurement_type() == srv.IMU:
            imu = srv.get_imu()
            for f in fl: f.process_imu(imu)
            continue
        stamp, ids, y = srv.get_vision()
        for f in fl: f.process_vision(stamp, srv.cam, ids, y)
        frames += 1
        d = np.abs(fl[0].get_sigma() - fl[1].get_sigma()); res.append((float(d.max()), int((d > 0).sum())))
    return res
for mode in (1, 0):
    for name, extra in (("defaults", []), ("speculative=0", [(7, 0)]), ("measure_in_propagate=0", [(19, 0)]), ("z_in_lookahead=0", [(17, 0)]), ("lookahead=0", [(12, 0)]), ("early_lift=0", [(8, 0)]), ("doorbell=0", [(6, 0)]), ("tiles_per_wg/gather off", [(23, 0)])):
        print("mode", mode, name, run(mode, extra), flush=True)
