"""Live cross-check of the two CPU restatements of the reference arithmetic: oracle/*.hpp (C++, quaternions, hand-written
loops) against oracle/indep/eqvio_ref.py (numpy, rotation matrices, dense expressions; written from /root/reference, not
from oracle/).  Analytic paths must agree to 1e-12; paths the reference itself evaluates by central differences
(h = cbrt(eps): Normal chart, discrete state matrix; Geometry.cpp:25-36) carry the differencing's own rounding noise
eps/h ~ 4e-11 per unit entry, so two evaluations of the same formulas agree to a few 1e-9 at best - which is also the bar any
device implementation of those paths can be held to."""
import os
import sys

import numpy as np
import pytest

from oracle_binding import OracleFilter
from util import CAMERAS, random_imu, random_spd, reasonable_state, rel_fro, settings_for, synth_measurement

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "indep"))
from eqvio_ref import Camera as RCam  # noqa: E402
from eqvio_ref import F64, EqVIORef  # noqa: E402

CHART_IDS = {"euclid": 0, "invdepth": 1, "normal": 2}


def _qerr(Qf, Q1):
    sg = np.sign(np.sum(Qf[:, :4] * Q1[:, :4], axis=1))[:, None]
    return max(np.abs(Qf[:, :4] * sg - Q1[:, :4]).max(), np.abs(Qf[:, 4] / Q1[:, 4] - 1).max())


def _xerr(a, b):
    a = a.copy()
    for sl in (slice(6, 10), slice(16, 20)):
        if np.dot(a[sl], b[sl]) < 0:
            a[sl] = -a[sl]
    return np.abs(a - b).max()


@pytest.mark.parametrize("cam_name", ["pinhole", "radtan", "equidistant"])
@pytest.mark.parametrize("chart_name", ["euclid", "invdepth", "normal"])
def test_oracle_matches_independent_restatement(chart_name, cam_name):
    numeric = chart_name == "normal"  # the Normal suite differentiates its chart numerically (normal.cpp:37-50, VIOState.cpp:391-401)
    tA, tS = (2e-9, 5e-9) if numeric else (1e-12, 1e-12)
    N = 6
    rng = np.random.default_rng(5)
    s = settings_for(CHART_IDS[chart_name], fastRiccati=0, useDiscreteInnovationLift=1, useDiscreteVelocityLift=0, measurementNoise=2.0)
    xi0, Xs, ids, q0, Q = reasonable_state(rng, N, shuffle_ids=True)
    S0 = random_spd(rng, 21 + 3 * N)
    cam = CAMERAS[cam_name]()
    imu = random_imu(rng, bias_vel=True)
    orc = OracleFilter(s)
    orc.set_eqf(xi0, Xs, ids, q0, Q, S0)
    r = EqVIORef(F64())
    X, st, im = r.group_from_flat(Xs, ids, Q), r.state_from_flat(xi0, ids, q0), r.imu_from_flat(imu)
    assert np.abs(r.state_matrix_A(chart_name, X, st, im) - orc.state_matrix_A(imu)).max() <= tA * 10
    assert np.abs(r.input_matrix_B(chart_name, X, st) - orc.input_matrix_B()).max() <= tA * 10
    Qin, P = r.diag(s.input_gain_diag12()), r.state_gain(s.state_gain_diag8(), N)
    # fast, accurate and (the reference's numerically differentiated) discrete Riccati, chained
    S1 = r.riccati_fast(chart_name, X, st, S0, im, 0.05, Qin, P)
    orc.integrate_riccati_fast(imu, 0.05)
    assert rel_fro(S1, orc.get_sigma()) <= tS
    S1 = r.riccati_accurate(chart_name, X, st, orc.get_sigma(), im, 0.02, Qin, P)
    orc.integrate_riccati_accurate(imu, 0.02)
    assert rel_fro(S1, orc.get_sigma()) <= tS
    S1 = r.riccati_discrete(chart_name, X, st, orc.get_sigma(), im, 0.02, Qin, P)
    orc.integrate_riccati_discrete(imu, 0.02)
    assert rel_fro(S1, orc.get_sigma()) <= 5e-9  # central differences on both sides
    S1 = orc.get_sigma()
    # observer: continuous and discrete lift
    for disc in (False, True):
        imu2 = random_imu(rng)
        X = r.integrate_observer(X, st, r.imu_from_flat(imu2), 0.004, disc)
        orc.integrate_observer(imu2, 0.004, disc)
    _, Xs1, _, _, Q1 = orc.get_eqf()
    Xf, Qf = r.group_to_flat(X)
    assert _xerr(Xf, Xs1) <= 1e-13 and _qerr(Qf, Q1) <= 1e-13
    # measurement side: C*, C, outlier statistics, update with the discrete innovation lift, NEES
    rc = RCam(cam.model, cam.fx, cam.fy, cam.cx, cam.cy, list(cam.dist))
    sub = np.sort(rng.permutation(N)[: N - 2])
    mid, y = synth_measurement(rng, cam, ids, q0, Q1, noise_px=1.5, subset=sub)
    meas = r.meas_from_flat(mid, y)
    for eqv in (True, False):
        Cr, Co = r.output_matrix_C(chart_name, st, X, rc, meas, eqv), orc.output_matrix_C(cam, mid, y, eqv)
        assert np.abs(Cr - Co).max() <= 1e-12 * max(1.0, np.abs(Co).max())
    a, p = r.outlier_stats(chart_name, X, st, S1, rc, meas)
    ao, po = orc.outlier_stats(cam, mid, y)
    idx = [list(ids).index(i) for i in sorted(mid)]
    assert np.abs(a - ao[idx]).max() <= 1e-12 and np.abs(p / po[idx] - 1).max() <= 1e-10
    X2, S2, g = r.vision_update(chart_name, X, st, S1, rc, meas, s.measurementNoise**2, True, True)
    orc.vision_update(cam, mid, y)
    _, Xs2, _, _, Q2 = orc.get_eqf()
    Xf, Qf = r.group_to_flat(X2)
    assert rel_fro(S2, orc.get_sigma()) <= 1e-11 and rel_fro(g, orc.last_gamma()) <= 1e-10
    assert _xerr(Xf, Xs2) <= 1e-11 and _qerr(Qf, Q2) <= 1e-11
    es, eids, ep = orc.state_estimate()
    ts = es.copy()
    ts[0:6] += rng.normal(size=6) * 1e-3
    ts[13:16] += rng.normal(size=3) * 1e-2
    tp = ep + rng.normal(size=ep.shape) * 1e-2
    ne = r.compute_nees(chart_name, X2, st, orc.get_sigma(), r.sensor_from_flat(ts), eids, tp)
    assert abs(ne / orc.compute_nees(ts, eids, tp) - 1) <= 1e-9


def test_continuous_innovation_lift_and_fast_riccati_default_path():
    """The shipped dataset configuration (InvDepth, fast Riccati, continuous innovation lift, discrete velocity lift), N = 12."""
    N = 12
    rng = np.random.default_rng(11)
    s = settings_for(1, fastRiccati=1, useDiscreteInnovationLift=0, useDiscreteVelocityLift=1, measurementNoise=1.5)
    xi0, Xs, ids, q0, Q = reasonable_state(rng, N, shuffle_ids=True)
    S0 = random_spd(rng, 21 + 3 * N)
    cam = CAMERAS["pinhole"]()
    orc = OracleFilter(s)
    orc.set_eqf(xi0, Xs, ids, q0, Q, S0)
    r = EqVIORef(F64())
    X, st = r.group_from_flat(Xs, ids, Q), r.state_from_flat(xi0, ids, q0)
    Qin, P = r.diag(s.input_gain_diag12()), r.state_gain(s.state_gain_diag8(), N)
    S = S0
    rc = RCam(0, cam.fx, cam.fy, cam.cx, cam.cy)
    for f in range(3):
        imu = random_imu(rng, bias_vel=True)
        S = r.riccati_fast("invdepth", X, st, S, r.imu_from_flat(imu), 0.05, Qin, P)
        orc.integrate_riccati_fast(imu, 0.05)
        for k in range(3):
            imu2 = random_imu(rng)
            X = r.integrate_observer(X, st, r.imu_from_flat(imu2), 0.005, True)
            orc.integrate_observer(imu2, 0.005, True)
        _, _, _, _, Q1 = orc.get_eqf()
        mid, y = synth_measurement(rng, cam, ids, q0, Q1, noise_px=1.0)
        X, S, g = r.vision_update("invdepth", X, st, S, rc, r.meas_from_flat(mid, y), s.measurementNoise**2, True, False)
        orc.vision_update(cam, mid, y)
        _, Xs2, _, _, Q2 = orc.get_eqf()
        Xf, Qf = r.group_to_flat(X)
        assert rel_fro(S, orc.get_sigma()) <= 1e-10 and _xerr(Xf, Xs2) <= 1e-9 and _qerr(Qf, Q2) <= 1e-9  # free running: the state follows Gamma = K y~ (conditioning of S)
