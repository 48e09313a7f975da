import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from eqvio_amd.capi import VIOFilter, PreparedFrames, Settings, COORD_NORMAL, COORD_INVDEPTH, COORD_EUCLIDEAN
from oracle_binding import OracleFilter
from run_configs import parity
from util import teacher_force
from eqvio_amd.simworld import SimWorld
from test_gpu_filter import sim_settings
import test_gpu_filter_headline as H

def run(name, world, s, nfr, force, maxlm=64, every=1):
    ids0, _ = world.vision(0.0)
    sensor, ids, p = world.true_state(0.0, ids0)
    orc = OracleFilter(s, sensor, ids, p, 0.0); flt = VIOFilter(s, max_landmarks=maxlm, sensor=sensor, ids=ids, p=p, time=0.0)
    ws = wS = 0.0
    for imus, stamp, mid, y in world.frames(nfr):
        for k in range(len(imus)):
            orc.process_imu(imus[k]); flt.process_imu(imus[k])
        orc.process_vision(stamp, world.cam, mid, y); flt.process_vision(stamp, world.cam, mid, y)
        es, eS = parity(flt, orc); ws, wS = max(ws, es), max(wS, eS)
        if force: teacher_force(flt, orc)
    print(f"{name} force={force}: state {ws:.2e} Sigma {wS:.2e}", flush=True)

for force in (0, 1):
    s = Settings.defaults(); s.coordinateChoice = COORD_NORMAL
    s.fastRiccati, s.useDiscreteInnovationLift, s.useMedianDepth = 1, 0, 1
    s.initialSceneDepth, s.initialPointVariance, s.measurementNoise = 4.0, 4.0, 1.5
    s.cameraOffset[:] = [0.5, -0.5, 0.5, -0.5, 0, 0, 0]
    run("normal chart 25 frames", SimWorld(seed=4, num_points=1200, max_features=25, trajectory="wave", noise_px=0.3), s, 25, force)
    for chart in (COORD_EUCLIDEAN, COORD_INVDEPTH, 2):
        run(f"discrete A chart {chart}", SimWorld(seed=13, num_points=500, max_features=14, trajectory="wave", noise_px=0.3), sim_settings(chart, fastRiccati=0, useDiscreteStateMatrix=1), 6, force, 48)
    run("long 600", SimWorld(seed=31, num_points=3000, max_features=20, trajectory="wave", noise_px=0.3), sim_settings(COORD_INVDEPTH), 600, force)
    # frame mix
    N, s, world, frames, sensor, ids, p = H._frame_mix_world()
    flt = VIOFilter(s, max_landmarks=N + 120, sensor=sensor, ids=ids, p=p, time=0.0); orc = OracleFilter(s, sensor, ids, p, 0.0)
    prepared = PreparedFrames(world.cam, *bench.flatten_frames(frames))
    for f, (imus, stamp, mid, y) in enumerate(frames[:8]):
        assert flt.run_prepared(prepared, f, 1) == 1
        for k_ in range(len(imus)): orc.process_imu(imus[k_])
        orc.process_vision(stamp, world.cam, mid, y)
        es, eS = parity(flt, orc)
        print(f"frame mix force={force} frame {f}: state {es:.2e} Sigma {eS:.2e}", flush=True)
        if force: teacher_force(flt, orc)
