"""Long lockstep run of the realistic frame mix against the oracle: wave world, landmark turnover, gross outliers in every frame, the shipped
retention; device-side outlier decision + masked update + deferred landmark bookkeeping on the device side, the reference's order in the oracle.
usage: python scripts/soak_mix_vs_oracle.py [frames] [features]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from eqvio_amd.capi import COORD_INVDEPTH, VIOFilter, load_eqf_lib
from oracle_binding import OracleFilter
from eqvio_amd.simworld import SimWorld
from test_gpu_filter import compare, sim_settings
nfr = int(sys.argv[1]) if len(sys.argv) > 1 else 400
nfe = int(sys.argv[2]) if len(sys.argv) > 2 else 60
world = SimWorld(seed=23, num_points=2500, max_features=nfe, trajectory="wave", noise_px=0.4)
settings = sim_settings(COORD_INVDEPTH, useMedianDepth=0, outlierThresholdAbs=6.0, outlierThresholdProb=4.0, featureRetention=0.186, initialPointVariance=0.05)
ids0, _ = world.vision(0.0)
sensor, ids, p = world.true_state(0.0, ids0)
orc = OracleFilter(settings, sensor, ids, p, 0.0)
flt = VIOFilter(settings, max_landmarks=32, sensor=sensor, ids=ids, p=p, time=0.0)
rng = np.random.default_rng(5)
for f, (imus, stamp, mid, y) in enumerate(world.frames(nfr)):
    y = y.copy()
    nb = int(rng.integers(0, 5))
    if nb:
        bad = rng.choice(len(mid), nb, replace=False)
        y.reshape(-1, 2)[bad] += rng.normal(size=(nb, 2)) * 25.0
    for s in range(len(imus)):
        orc.process_imu(imus[s]); flt.process_imu(imus[s])
    orc.process_vision(stamp, world.cam, mid, y); flt.process_vision(stamp, world.cam, mid, y)
    compare(flt, orc, 1e-8)
fr, di = C.c_long(), C.c_long()
load_eqf_lib().eqf_selection_stats(flt.core_handle(), C.byref(fr), C.byref(di), 0)
print(f"{nfr} frames in step with the oracle (1e-8); device-side decisions in {fr.value} frames, {di.value} landmarks discarded; landmarks now {len(flt.state_estimate()[1])}")
