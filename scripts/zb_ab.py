"""A/B of EQF_OPT_Z_IN_LOOKAHEAD on the stand-alone entry points (eqf_integrate_riccati_fast + eqf_vision_update: k_measure, [k_build_Z,] look-ahead kernel, lift, SYRK),
same context, alternating. usage: python scripts/zb_ab.py [N] [iterations]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from eqvio_amd.capi import EqfCore, OPT_Z_IN_LOOKAHEAD
from util import CHARTS, default_camera, random_imu, random_spd, reasonable_state, settings_for, synth_measurement

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
its = int(sys.argv[2]) if len(sys.argv) > 2 else 300
rng = np.random.default_rng(1)
xi0, Xs, ids, q0, Q = reasonable_state(rng, N)
S = random_spd(rng, 21 + 3 * N)
settings = settings_for(CHARTS["invdepth"])
cam = default_camera()
mid, y = synth_measurement(rng, cam, ids, q0, Q, noise_px=1.0, subset=np.arange(N))
imu = random_imu(rng)
for rep in range(3):
    for val in (1, 0):
        c = EqfCore(N, CHARTS["invdepth"])
        c.set_state(xi0, Xs, ids, q0, Q)
        c.set_sigma(S)
        c.set_option(OPT_Z_IN_LOOKAHEAD, val)
        for k in range(20):
            c.integrate_riccati_fast(imu, 0.005, settings.input_gain_diag12(), settings.state_gain_diag8())
            c.vision_update(cam, mid, y + 0.01 * k, settings.measurementNoise**2, True, False)
        t0 = time.perf_counter()
        for k in range(its):
            c.integrate_riccati_fast(imu, 0.005, settings.input_gain_diag12(), settings.state_gain_diag8())
            c.vision_update(cam, mid, y + 0.01 * (k % 7), settings.measurementNoise**2, True, False)
        el = time.perf_counter() - t0
        print(f"N={N} Z in look-ahead kernel = {val}: {1e6 * el / its:7.1f} us per propagate + update", flush=True)
