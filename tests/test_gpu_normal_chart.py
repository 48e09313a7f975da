"""Normal coordinate chart on the device (coordinateSuite/normal.cpp:37-65): the reference obtains the change of
coordinates M by CENTRAL DIFFERENCES of the charts (VIOState.cpp:391-401, h = cbrt(eps)) and forms M A_e M^-1, M B_e, lifts through M^-1.
The device uses the closed form of M (block diagonal: eqf_math.hpp normal_M) and applies it as two congruences around the Euclidean
propagation. The oracle differentiates numerically like the reference, so the two agree to the differencing's own rounding noise
(eps / h ~ 4e-11 per unit entry; tests/test_indep_restatement.py shows the same level between the two CPU restatements): A, B to 1e-8
absolute, Sigma / state to 1e-9 relative."""
import numpy as np
import pytest

from eqvio_amd.capi import COORD_NORMAL, EqfCore, Settings, VIOFilter
from oracle_binding import OracleFilter, se3_log_dist
from eqvio_amd.simworld import SimWorld
from util import CAMERAS, random_imu, random_spd, reasonable_state, rel_fro, settings_for, synth_measurement

from util import teacher_force  # noqa: E402

pytestmark = pytest.mark.gpu


def _pair(N, seed, **kw):
    rng = np.random.default_rng(seed)
    s = settings_for(COORD_NORMAL, **kw)
    xi0, Xs, ids, q0, Q = reasonable_state(rng, N, shuffle_ids=True)
    S = random_spd(rng, 21 + 3 * N)
    orc = OracleFilter(s)
    orc.set_eqf(xi0, Xs, ids, q0, Q, S)
    core = EqfCore(N, COORD_NORMAL)
    core.set_state(xi0, Xs, ids, q0, Q)
    core.set_sigma(S)
    return rng, s, orc, core


def _state_close(core, orc, tol):
    _, Xg, idg, _, Qg = core.get_state()
    _, Xo, ido, _, Qo = orc.get_eqf()
    assert np.array_equal(idg, ido)
    assert np.max(np.abs(Xg[0:6] - Xo[0:6])) <= tol and np.max(np.abs(Xg[13:16] - Xo[13:16])) <= tol
    assert se3_log_dist(Xg[6:13], Xo[6:13]) <= tol and se3_log_dist(Xg[16:23], Xo[16:23]) <= tol
    sg = np.sign(np.sum(Qg[:, :4] * Qo[:, :4], axis=1))[:, None]
    assert np.max(np.abs(Qg[:, :4] * sg - Qo[:, :4])) <= tol and np.max(np.abs(Qg[:, 4] / Qo[:, 4] - 1)) <= tol


@pytest.mark.parametrize("N", [1, 7, 40])
def test_matrices_A_B_C(N):
    rng, s, orc, core = _pair(N, 100 + N)
    imu = random_imu(rng, bias_vel=True)
    A, B = core.debug_matrices_AB(imu)
    assert np.max(np.abs(A - orc.state_matrix_A(imu))) <= 1e-8 * max(1.0, np.max(np.abs(A)))
    assert np.max(np.abs(B - orc.input_matrix_B())) <= 1e-8 * max(1.0, np.max(np.abs(B)))
    for cam_name in ("pinhole", "radtan", "equidistant"):
        cam = CAMERAS[cam_name]()
        _, _, ids, q0, Q = core.get_state()
        mid, y = synth_measurement(rng, cam, ids, q0, Q, noise_px=1.0, subset=np.sort(rng.permutation(N)[: max(1, N - 2)]))
        for eqv in (True, False):  # the Normal suite's C*_i ignores the pixel: both give the same block (normal.cpp:57-65)
            C, _ = core.debug_matrix_C(cam, mid, y, eqv)
            Co = orc.output_matrix_C(cam, mid, y, eqv)
            assert np.max(np.abs(C - Co)) <= 1e-11 * max(1.0, np.max(np.abs(Co)))


@pytest.mark.parametrize("mode", ["fast", "accurate", "discrete"])
def test_riccati(mode):
    rng, s, orc, core = _pair(23, 7)
    imu = random_imu(rng, bias_vel=True)
    Qd, Pd = s.input_gain_diag12(), s.state_gain_diag8()
    if mode == "fast":
        core.integrate_riccati_fast(imu, 0.05, Qd, Pd)
        orc.integrate_riccati_fast(imu, 0.05)
    elif mode == "accurate":
        core.integrate_riccati_accurate(imu, 0.02, Qd, Pd)
        orc.integrate_riccati_accurate(imu, 0.02)
    else:  # the reference differentiates a0Discrete in normal coordinates; the device conjugates the Euclidean A_d with the closed-form M
        core.integrate_riccati_discrete(imu, 0.01, Qd, Pd)
        orc.integrate_riccati_discrete(imu, 0.01)
    Sg, So = core.get_sigma(), orc.get_sigma()
    assert np.array_equal(Sg, Sg.T) or mode == "discrete"
    assert rel_fro(Sg, So) <= (5e-9 if mode == "discrete" else 1e-9), rel_fro(Sg, So)


@pytest.mark.parametrize("discrete", [False, True])
def test_vision_update_and_nees(discrete):
    rng, s, orc, core = _pair(31, 11, useDiscreteInnovationLift=int(discrete), measurementNoise=1.5)
    cam = CAMERAS["pinhole"]()
    _, _, ids, q0, Q = core.get_state()
    mid, y = synth_measurement(rng, cam, ids, q0, Q, noise_px=1.0, subset=np.sort(rng.permutation(31)[:27]))
    a_g, p_g, _ = core.outlier_stats(cam, mid, y)
    a_o, p_o = orc.outlier_stats(cam, mid, y)
    seen = p_o >= 0
    assert np.allclose(a_g[seen], a_o[seen], rtol=1e-10, atol=1e-10) and np.allclose(p_g[seen], p_o[seen], rtol=1e-8, atol=1e-10)
    core.vision_update(cam, mid, y, s.measurementNoise**2, True, discrete)
    orc.vision_update(cam, mid, y)
    assert rel_fro(core.get_sigma(), orc.get_sigma()) <= 1e-9
    assert rel_fro(core.last_gamma(), orc.last_gamma()) <= 1e-9
    _state_close(core, orc, 1e-9)
    es, eids, ep = orc.state_estimate()
    ts = es.copy()
    ts[0:6] += rng.normal(size=6) * 1e-3
    ts[6:10] = ts[6:10] + rng.normal(size=4) * 1e-3
    ts[6:10] /= np.linalg.norm(ts[6:10])
    ts[13:16] += rng.normal(size=3) * 1e-2
    tp = ep + rng.normal(size=ep.shape) * 1e-2
    assert abs(core.compute_nees(ts, eids, tp) / orc.compute_nees(ts, eids, tp) - 1) <= 1e-8


@pytest.mark.parametrize("teacher_forced", [True, False])
def test_filter_run_with_landmark_turnover(teacher_forced):
    """The host VIOFilter mirror with coordinateChoice = Normal against the oracle's filter on the wave world (landmarks enter and leave),
    both lifts as in the template configuration (discrete velocity lift, continuous innovation lift). Teacher forced (every frame from the oracle's state;
    measured 2e-11 / 9e-11) and free running (5e-10 / 1e-10 after 25 frames): flat 1e-9 either way."""
    world = SimWorld(seed=4, num_points=1200, max_features=25, trajectory="wave", noise_px=0.3)
    s = Settings.defaults()
    s.coordinateChoice = COORD_NORMAL
    s.fastRiccati, s.useDiscreteInnovationLift, s.useMedianDepth = 1, 0, 1
    s.initialSceneDepth, s.initialPointVariance, s.measurementNoise = 4.0, 4.0, 1.5
    s.cameraOffset[:] = [0.5, -0.5, 0.5, -0.5, 0, 0, 0]
    ids0, _ = world.vision(0.0)
    sensor, ids, p = world.true_state(0.0, ids0)
    orc = OracleFilter(s, sensor, ids, p, 0.0)
    flt = VIOFilter(s, max_landmarks=64, sensor=sensor, ids=ids, p=p, time=0.0)
    for imus, stamp, mid, y in world.frames(25):
        for k in range(len(imus)):
            orc.process_imu(imus[k])
            flt.process_imu(imus[k])
        orc.process_vision(stamp, world.cam, mid, y)
        flt.process_vision(stamp, world.cam, mid, y)
        s_g, ids_g, p_g = flt.state_estimate()
        s_o, ids_o, p_o = orc.state_estimate()
        assert np.array_equal(ids_g, ids_o)
        assert np.max(np.abs(s_g - s_o)) <= 1e-9 and np.max(np.abs(p_g - p_o) / np.maximum(1.0, np.abs(p_o))) <= 1e-9
        assert rel_fro(flt.get_sigma(), orc.get_sigma()) <= 1e-9
        if teacher_forced:
            teacher_force(flt, orc)
