import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from eqvio_amd.capi import EqfCore, COORD_INVDEPTH
from oracle_binding import OracleFilter
from util import *
for N in [int(a) for a in sys.argv[1:]]:
    rng = np.random.default_rng(1)
    settings = settings_for(COORD_INVDEPTH, fastRiccati=1, useDiscreteInnovationLift=0, initialPointVariance=9.0, measurementNoise=1.93)
    xi0, Xs, ids, q0, Q = reasonable_state(rng, N)
    S = np.diag(settings.initial_cov_diag(N))
    core = EqfCore(N, COORD_INVDEPTH); core.set_state(xi0, Xs, ids, q0, Q); core.set_sigma(S)
    orc = OracleFilter(settings); orc.set_eqf(xi0, Xs, ids, q0, Q, S); orc.set_arithmetic(2)
    cam = euroc_camera()
    imu = random_imu(rng)
    core.integrate_riccati_fast(imu, 0.05, settings.input_gain_diag12(), settings.state_gain_diag8()); orc.integrate_riccati_fast(imu, 0.05)
    print(N, "prop err", rel_fro(core.get_sigma(), orc.get_sigma()))
    mid, y = synth_measurement(rng, cam, ids, q0, Q, noise_px=1.0)
    try:
        core.vision_update(cam, mid, y, settings.measurementNoise**2, True, False)
        orc.vision_update(cam, mid, y)
        print(N, "upd err", rel_fro(core.get_sigma(), orc.get_sigma()), "gamma", np.linalg.norm(core.last_gamma()-orc.last_gamma())/np.linalg.norm(orc.last_gamma()))
    except Exception as e:
        print(N, "FAILED", e)
