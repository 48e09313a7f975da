import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from eqvio_amd.capi import EqfCore, COORD_INVDEPTH
from oracle_binding import OracleFilter
from util import *
N = int(sys.argv[1]); use_orc = len(sys.argv) > 2
rng = np.random.default_rng(1)
settings = settings_for(COORD_INVDEPTH, fastRiccati=1, useDiscreteInnovationLift=0, initialPointVariance=9.0, measurementNoise=1.93)
xi0, Xs, ids, q0, Q = reasonable_state(rng, N)
S = np.diag(settings.initial_cov_diag(N))
core = EqfCore(N, COORD_INVDEPTH); core.set_state(xi0, Xs, ids, q0, Q); core.set_sigma(S)
if use_orc:
    orc = OracleFilter(settings); orc.set_eqf(xi0, Xs, ids, q0, Q, S); orc.set_arithmetic(2)
cam = euroc_camera()
Qd, Pd = settings.input_gain_diag12(), settings.state_gain_diag8()
k = 10
for f in range(10):
    imus = np.stack([random_imu(rng) * np.array([1] + [0.02] * 3 + [0.1] * 3 + [0] * 6) for _ in range(k)])
    dts = np.full(k, 0.005)
    mean = (imus * dts[:, None]).sum(0) / dts.sum()
    core.integrate_riccati_fast(mean, dts.sum(), Qd, Pd)
    core.integrate_observer(imus, dts, True)
    _, Xs_, ids_, q0_, Q_ = core.get_state()
    mid, y = synth_measurement(rng, cam, ids_, q0_, Q_, noise_px=1.0)
    Sg = core.get_sigma()
    msg = f"frame {f}: prior min eig {np.linalg.eigvalsh(0.5*(Sg+Sg.T)).min():.3e} Qa range {Q_[:,4].min():.3f} {Q_[:,4].max():.3f}"
    if use_orc:
        orc.integrate_riccati_fast(mean, dts.sum())
        for s in range(k): orc.integrate_observer(imus[s], dts[s], True)
        msg += f" prior err {rel_fro(Sg, orc.get_sigma()):.2e}"
    try:
        core.vision_update(cam, mid, y, settings.measurementNoise**2, True, False)
    except Exception as e:
        print(msg, "GPU FAILED", e); break
    Sg = core.get_sigma()
    msg += f" post min eig {np.linalg.eigvalsh(0.5*(Sg+Sg.T)).min():.3e} gamma max {np.abs(core.last_gamma()).max():.3e}"
    if use_orc:
        orc.vision_update(cam, mid, y); msg += f" post err {rel_fro(Sg, orc.get_sigma()):.2e}"
    print(msg)
