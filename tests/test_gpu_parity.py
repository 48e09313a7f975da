"""Parity of the HIP EqF core (libeqf_hip.so, through the C-ABI of include/eqf_hip.h) against the CPU oracle
(the restated reference) on identical seeded inputs. fp64; tolerance per BASELINE.json north_star: 1e-9
relative on pose, landmarks and Sigma (kernel-level blocks are held to much tighter bounds)."""
import numpy as np
import pytest

from eqvio_amd.capi import COORD_INVDEPTH, OPT_RICCATI_DENSE, EqfCore, EqfError
from oracle_binding import OracleFilter, se3_log_dist
from util import CHARTS, default_camera, euroc_camera, random_imu, random_spd, reasonable_state, rel_fro, settings_for, synth_measurement

pytestmark = pytest.mark.gpu

TOL = 1e-9  # north_star: "within 1e-9 relative in fp64"


def make_pair(chart, N, seed, cap=None, sigma="spd", **skw):
    rng = np.random.default_rng(seed)
    settings = settings_for(chart, **skw)
    xi0, Xs, ids, q0, Q = reasonable_state(rng, N, shuffle_ids=True)
    n = 21 + 3 * N
    S = random_spd(rng, n) if sigma == "spd" else np.diag(settings.initial_cov_diag(N))
    orc = OracleFilter(settings)
    orc.set_eqf(xi0, Xs, ids, q0, Q, S)
    core = EqfCore(cap or max(N, 1), chart)
    core.set_state(xi0, Xs, ids, q0, Q)
    core.set_sigma(S)
    return rng, settings, orc, core, (xi0, Xs, ids, q0, Q, S)


def check_state(core, orc, tol=TOL):
    xi0_g, Xs_g, ids_g, q0_g, Q_g = core.get_state()
    xi0_o, Xs_o, ids_o, q0_o, Q_o = orc.get_eqf()
    assert np.array_equal(ids_g, ids_o)
    np.testing.assert_allclose(xi0_g, xi0_o, rtol=0, atol=1e-14)
    np.testing.assert_allclose(q0_g, q0_o, rtol=0, atol=0)
    # group sensor part: beta, w absolute; A, B by SE3 log distance (SURVEY.md §8(d) parity definition)
    assert np.max(np.abs(Xs_g[0:6] - Xs_o[0:6])) <= tol * max(1.0, np.max(np.abs(Xs_o[0:6])))
    assert np.max(np.abs(Xs_g[13:16] - Xs_o[13:16])) <= tol * max(1.0, np.max(np.abs(Xs_o[13:16])))
    assert se3_log_dist(Xs_g[6:13], Xs_o[6:13]) <= tol * max(1.0, np.linalg.norm(Xs_o[10:13]))
    assert se3_log_dist(Xs_g[16:23], Xs_o[16:23]) <= tol * max(1.0, np.linalg.norm(Xs_o[20:23]))
    # landmark transforms: quaternion up to sign, scale relative
    for i in range(len(ids_g)):
        qg, qo = Q_g[i, :4], Q_o[i, :4]
        if np.dot(qg, qo) < 0:
            qg = -qg
        assert np.max(np.abs(qg - qo)) <= tol, (i, qg, qo)
        assert abs(Q_g[i, 4] - Q_o[i, 4]) <= tol * abs(Q_o[i, 4])
    # state estimate (pose, velocity, landmarks)
    s_g, _, p_g = core.state_estimate()
    s_o, _, p_o = orc.state_estimate()
    assert se3_log_dist(s_g[6:13], s_o[6:13]) <= tol * max(1.0, np.linalg.norm(s_o[10:13]))
    assert np.max(np.abs(s_g[13:16] - s_o[13:16])) <= tol * max(1.0, np.max(np.abs(s_o[13:16])))
    if len(p_o):
        assert np.max(np.linalg.norm(p_g - p_o, axis=1) / np.maximum(1.0, np.linalg.norm(p_o, axis=1))) <= tol


def check_sigma(core, orc, tol=TOL):
    Sg, So = core.get_sigma(), orc.get_sigma()
    assert Sg.shape == So.shape
    assert np.all(np.isfinite(Sg))
    assert rel_fro(Sg, So) <= tol, rel_fro(Sg, So)
    return Sg


@pytest.mark.parametrize("chart", list(CHARTS))
@pytest.mark.parametrize("N", [1, 5, 20, 50])
def test_matrices_AB(chart, N):
    """K1 vs EqFStateMatrixA / EqFInputMatrixB (euclid.cpp:99-233, invdepth.cpp:36-181), dense comparison."""
    rng, settings, orc, core, _ = make_pair(CHARTS[chart], N, seed=100 + N)
    imu = random_imu(rng, bias_vel=True)
    A_g, B_g = core.debug_matrices_AB(imu)
    A_o, B_o = orc.state_matrix_A(imu), orc.input_matrix_B()
    assert np.max(np.abs(A_g - A_o)) <= 1e-11 * max(1.0, np.max(np.abs(A_o)))
    assert np.max(np.abs(B_g - B_o)) <= 1e-11 * max(1.0, np.max(np.abs(B_o)))
    # structural zeros of the packed form: columns 3:12 of landmark rows are zero in the reference too
    assert np.all(A_o[21:, 3:12] == 0)


@pytest.mark.parametrize("chart", list(CHARTS))
@pytest.mark.parametrize("star", [True, False])
def test_matrix_C_and_residual(chart, star):
    """K3 vs outputMatrixC (EqFMatrices.cpp:43-82) and yTilde (VisionMeasurement.cpp:60-79), M < N and ragged ids."""
    N = 23
    rng, settings, orc, core, (xi0, Xs, ids, q0, Q, S) = make_pair(CHARTS[chart], N, seed=7)
    cam = default_camera()
    subset = rng.permutation(N)[:17]
    mid, y = synth_measurement(rng, cam, ids, q0, Q, noise_px=2.0, subset=subset)
    C_g, yt_g = core.debug_matrix_C(cam, mid, y, star)
    C_o = orc.output_matrix_C(cam, mid, y, star)
    assert np.max(np.abs(C_g - C_o)) <= 1e-11 * max(1.0, np.max(np.abs(C_o)))
    # residual: y - project(q_hat)
    _, ids_e, p_e = orc.state_estimate()
    lut = {int(i): k for k, i in enumerate(ids_e)}
    yhat = np.array([[cam.fx * p_e[lut[int(i)], 0] / p_e[lut[int(i)], 2] + cam.cx, cam.fy * p_e[lut[int(i)], 1] / p_e[lut[int(i)], 2] + cam.cy] for i in mid]).reshape(-1)
    np.testing.assert_allclose(yt_g, y - yhat, rtol=0, atol=1e-10)


@pytest.mark.parametrize("chart", list(CHARTS))
@pytest.mark.parametrize("N", [0, 1, 5, 20, 50])
def test_riccati_fast(chart, N):
    """K2 (arrow form) vs integrateRiccatiStateFast (VIO_eqf.cpp:62-72), including N = 0 (sensor block only)."""
    rng, settings, orc, core, _ = make_pair(CHARTS[chart], N, seed=200 + N, cap=max(N, 4))
    imu = random_imu(rng, bias_vel=True)
    dt = 0.05
    orc.integrate_riccati_fast(imu, dt)
    core.integrate_riccati_fast(imu, dt, settings.input_gain_diag12(), settings.state_gain_diag8())
    Sg = check_sigma(core, orc, 1e-12)
    assert np.max(np.abs(Sg - Sg.T)) <= 1e-13 * np.max(np.abs(Sg))  # (i,j) and (j,i) tiles are computed independently


@pytest.mark.parametrize("chart", list(CHARTS))
@pytest.mark.parametrize("N,dt", [(0, 0.005), (1, 0.005), (5, 0.05), (20, 0.005), (50, 0.005), (20, 0.7)])
def test_riccati_accurate(chart, N, dt):
    """Structured exp(dt [[A,B],[0,0]]) + dense MFMA GEMMs vs integrateRiccatiStateAccurate (VIO_eqf.cpp:74-91).
    dt = 0.7 forces the scaling-and-squaring branch of the device exponential (several squarings)."""
    rng, settings, orc, core, _ = make_pair(CHARTS[chart], N, seed=300 + N, cap=max(N, 4))
    Qd, Pd = settings.input_gain_diag12(), settings.state_gain_diag8()
    for rep in range(2):  # twice: the second call starts from a non-trivial ping-pong state
        imu = random_imu(rng, bias_vel=True)
        orc.integrate_riccati_accurate(imu, dt)
        core.integrate_riccati_accurate(imu, dt, Qd, Pd)
        check_sigma(core, orc, 1e-11)


@pytest.mark.parametrize("chart", list(CHARTS))
def test_riccati_dense_mode_matches_structured(chart):
    """EQF_OPT_RICCATI_DENSE (two fp64 MFMA GEMMs, F materialised) == arrow-form kernel == oracle."""
    N = 37
    rng, settings, orc, core, data = make_pair(CHARTS[chart], N, seed=31)
    imu = random_imu(rng)
    orc.integrate_riccati_fast(imu, 0.04)
    core.set_option(OPT_RICCATI_DENSE, 1)
    core.integrate_riccati_fast(imu, 0.04, settings.input_gain_diag12(), settings.state_gain_diag8())
    check_sigma(core, orc, 1e-12)


@pytest.mark.parametrize("chart", list(CHARTS))
@pytest.mark.parametrize("discrete", [True, False])
def test_observer_integration(chart, discrete):
    """K4 + host sensor part vs integrateObserverState (VIO_eqf.cpp:47-60) over 12 IMU samples."""
    N = 20
    rng, settings, orc, core, _ = make_pair(CHARTS[chart], N, seed=5)
    k = 12
    imus = np.stack([random_imu(rng, stamp=0.005 * s, bias_vel=True) for s in range(k)])
    dts = rng.uniform(0.001, 0.006, k)
    dts[3] = 0.0  # clipped interval (VIOFilter.cpp:161-163 yields dt = 0 for stale samples)
    for s in range(k):
        orc.integrate_observer(imus[s], dts[s], discrete)
    core.integrate_observer(imus, dts, discrete)
    check_state(core, orc, 1e-12)


@pytest.mark.parametrize("chart", list(CHARTS))
@pytest.mark.parametrize("N,M", [(1, 1), (5, 5), (20, 20), (20, 13), (50, 50)])
@pytest.mark.parametrize("discrete", [False, True])
def test_vision_update(chart, N, M, discrete):
    """K3/K8/K9/K10 vs performVisionUpdate (VIO_eqf.cpp:105-135): Gamma, Sigma+, X+ — incl. M < N."""
    rng, settings, orc, core, (xi0, Xs, ids, q0, Q, S) = make_pair(CHARTS[chart], N, seed=300 + N + M, useDiscreteInnovationLift=int(discrete))
    cam = euroc_camera()
    subset = rng.permutation(N)[:M]
    mid, y = synth_measurement(rng, cam, ids, q0, Q, noise_px=1.5, subset=subset)
    orc.vision_update(cam, mid, y)
    core.vision_update(cam, mid, y, settings.measurementNoise**2, True, discrete)
    g_o, g_g = orc.last_gamma(), core.last_gamma()
    assert np.linalg.norm(g_g - g_o) <= TOL * max(1.0, np.linalg.norm(g_o))
    check_sigma(core, orc)
    check_state(core, orc)


@pytest.mark.parametrize("chart", list(CHARTS))
def test_full_frame_sequence(chart):
    """Several frames of propagate (fast Riccati) + observer steps + update, free running, N = 30."""
    N = 30
    rng, settings, orc, core, (xi0, Xs, ids, q0, Q, S) = make_pair(
        CHARTS[chart], N, seed=11, sigma="init", fastRiccati=1, useDiscreteInnovationLift=0, initialPointVariance=4.0, measurementNoise=1.5)
    cam = euroc_camera()
    Qd, Pd = settings.input_gain_diag12(), settings.state_gain_diag8()
    for frame in range(6):
        k = 10
        imus = np.stack([random_imu(rng) * np.array([1] + [0.05] * 3 + [1] * 3 + [0] * 6) for _ in range(k)])
        dts = np.full(k, 0.005)
        mean = (imus * dts[:, None]).sum(0) / dts.sum()
        orc.integrate_riccati_fast(mean, dts.sum())
        core.integrate_riccati_fast(mean, dts.sum(), Qd, Pd)
        for s in range(k):
            orc.integrate_observer(imus[s], dts[s], True)
        core.integrate_observer(imus, dts, True)
        _, Xs_o, ids_o, q0_o, Q_o = orc.get_eqf()
        mid, y = synth_measurement(rng, cam, ids_o, q0_o, Q_o, noise_px=1.0)
        orc.vision_update(cam, mid, y)
        core.vision_update(cam, mid, y, settings.measurementNoise**2, True, False)
        check_sigma(core, orc)
        check_state(core, orc)


@pytest.mark.parametrize("chart", list(CHARTS))
def test_landmark_bookkeeping(chart):
    """K5/K7 vs removeLandmarkByIndex / addNewLandmarks (VIO_eqf.cpp:172-178, 225-245): remove a ragged set,
    append, remove everything, append again; Sigma and X must match entry for entry."""
    N = 19
    rng, settings, orc, core, _ = make_pair(CHARTS[chart], N, seed=77, cap=40)
    drop = sorted([0, 3, 4, 11, 18])
    for idx in reversed(drop):
        orc.remove_landmark_by_index(idx)
    core.remove_landmarks(drop[::-1])
    assert np.array_equal(core.get_sigma(), orc.get_sigma())
    check_state(core, orc, 1e-15)
    new_ids = np.array([1000, 7, 1002], dtype=np.int32)
    new_p = rng.uniform(-1, 1, (3, 3)) + np.array([0, 0, 5.0])
    orc.add_landmarks(new_ids, new_p, 2.5)
    core.add_landmarks(new_ids, new_p, 2.5)
    assert np.array_equal(core.get_sigma(), orc.get_sigma())
    check_state(core, orc, 1e-15)
    # remove all, then re-add (empty state is legal: VIOFilter.cpp:31-41 starts with N = 0)
    nN = core.N
    for idx in reversed(range(nN)):
        orc.remove_landmark_by_index(idx)
    core.remove_landmarks(np.arange(nN))
    assert core.N == 0 and core.get_sigma().shape == (21, 21)
    assert np.array_equal(core.get_sigma(), orc.get_sigma())
    orc.add_landmarks(new_ids, new_p, 1.0)
    core.add_landmarks(new_ids, new_p, 1.0)
    assert np.array_equal(core.get_sigma(), orc.get_sigma())
    # far more landmarks than the context was created for (cap = 40): the reference has no cap (VIO_eqf.cpp:225-245); the context grows
    many_ids, many_p = np.arange(100, 200, dtype=np.int32), rng.uniform(-1, 1, (100, 3)) + np.array([0, 0, 5.0])
    orc.add_landmarks(many_ids, many_p, 0.7)
    core.add_landmarks(many_ids, many_p, 0.7)
    assert core.N == 103 and np.array_equal(core.get_sigma(), orc.get_sigma())
    check_state(core, orc, 1e-15)


@pytest.mark.parametrize("chart", list(CHARTS))
def test_outlier_stats(chart):
    """eqf_outlier_stats vs VIOFilter::removeOutliers' per-landmark quantities (VIOFilter.cpp:304-334)."""
    N = 25
    rng, settings, orc, core, (xi0, Xs, ids, q0, Q, S) = make_pair(CHARTS[chart], N, seed=13)
    cam = default_camera()
    subset = rng.permutation(N)[:19]
    mid, y = synth_measurement(rng, cam, ids, q0, Q, noise_px=3.0, subset=subset)
    a_g, p_g, d_g = core.outlier_stats(cam, mid, y)
    a_o, p_o = orc.outlier_stats(cam, mid, y)
    np.testing.assert_allclose(a_g, a_o, rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(p_g, p_o, rtol=1e-9, atol=1e-11)
    _, _, p_e = orc.state_estimate()
    np.testing.assert_allclose(d_g, (p_e**2).sum(1), rtol=1e-13)


@pytest.mark.parametrize("chart", list(CHARTS))
@pytest.mark.parametrize("N", [1, 25, 200])
def test_output_cov_all(chart, N):
    """eqf_output_cov_all (what a binding with the reference's unchanged VIOFilter.cpp serves getOutputCovById from) against the oracle's
    getOutputCovById (VIO_eqf.cpp:196-211) for every landmark, after a propagation (Sigma not block diagonal any more)."""
    rng, settings, orc, core, (xi0, Xs, ids, q0, Q, S) = make_pair(CHARTS[chart], N, seed=40 + N)
    cam = default_camera()
    imu = random_imu(rng)
    core.integrate_riccati_fast(imu, 0.02, settings.input_gain_diag12(), settings.state_gain_diag8())
    orc.integrate_riccati_fast(imu, 0.02)
    g, o = core.output_cov_all(cam), orc.output_cov_all(cam)
    assert g.shape == (N, 2, 2)
    np.testing.assert_allclose(g, o, rtol=1e-10, atol=1e-12 * np.abs(o).max())
    assert np.allclose(g, np.transpose(g, (0, 2, 1)), rtol=0, atol=1e-13 * np.abs(g).max())  # v01 and v10 are separate sums


def test_output_cov_all_after_the_landmark_set_changed():
    """The sequence of the reference-side binding, stateEstimate(); removeLandmarkById() / addNewLandmarks(); getOutputCovById() (VIOFilter.cpp:304-334 around :258-302):
    the covariances that were computed along with the state estimate belong to the OLD landmark set and must not be served afterwards (they were)."""
    N = 30
    rng, settings, orc, core, (xi0, Xs, ids, q0, Q, S) = make_pair(CHARTS["invdepth"], N, seed=77)
    cam = default_camera()
    imu = random_imu(rng)
    core.integrate_riccati_fast(imu, 0.02, settings.input_gain_diag12(), settings.state_gain_diag8())
    orc.integrate_riccati_fast(imu, 0.02)
    core.output_cov_all(cam)  # (the hint: from here on a state estimate computes the covariances as well)
    core.integrate_riccati_fast(imu, 0.01, settings.input_gain_diag12(), settings.state_gain_diag8())
    orc.integrate_riccati_fast(imu, 0.01)
    for step in range(3):
        orc.integrate_observer(imu, 0.01, True)  # (the landmark estimates change: the next state estimate asks the device, and computes the covariances along with it)
        core.integrate_observer(imu[None, :], np.array([0.01]), True)
        core.state_estimate()  # fills the cache for the current landmark set
        if step == 0:  # remove two landmarks in the middle
            for idx in (11, 4):
                orc.remove_landmark_by_index(idx)
            core.remove_landmarks(np.array([4, 11], dtype=np.int32))
        elif step == 1:  # append three
            nid, npnt = np.arange(500, 503, dtype=np.int32), rng.uniform(-1, 1, (3, 3)) + np.array([0, 0, 4.0])
            orc.add_landmarks(nid, npnt, 0.6)
            core.add_landmarks(nid, npnt, 0.6)
        else:  # remove the last one and append another in one go
            orc.remove_landmark_by_index(core.N - 1)
            core.remove_landmarks(np.array([core.N - 1], dtype=np.int32))
            nid, npnt = np.array([900], dtype=np.int32), np.array([[0.2, -0.1, 3.0]])
            orc.add_landmarks(nid, npnt, 0.6)
            core.add_landmarks(nid, npnt, 0.6)
        g, o = core.output_cov_all(cam), orc.output_cov_all(cam)
        assert g.shape == o.shape == (core.N, 2, 2)
        np.testing.assert_allclose(g, o, rtol=1e-10, atol=1e-12 * np.abs(o).max())


def test_update_rejects_unknown_and_unsorted_ids():
    rng, settings, orc, core, (xi0, Xs, ids, q0, Q, S) = make_pair(CHARTS["euclid"], 6, seed=3)
    cam = default_camera()
    mid, y = synth_measurement(rng, cam, ids, q0, Q)
    with pytest.raises(EqfError):
        core.vision_update(cam, mid[::-1].copy(), y, 4.0)
    bad = mid.copy()
    bad[-1] = 99999
    with pytest.raises(EqfError):
        core.vision_update(cam, bad, y, 4.0)
    # empty measurement is a no-op (VIO_eqf.cpp:108-109)
    core.vision_update(cam, np.zeros(0, np.int32), np.zeros(0), 4.0)
    check_sigma(core, orc, 0.0)


@pytest.mark.parametrize("N", [200, 500])
def test_headline_size_properties(N):
    """BASELINE.json sizes (N = 200: n = 621, m = 400, the metric; N = 500: n = 1521, m = 1000, the stress config): oracle parity on one frame (the oracle needs ~1 s / ~15 s here)
    plus size-independent properties: Sigma stays symmetric, the update never increases any variance, and the
    posterior satisfies the Joseph identity Sigma+ = (I - K C) Sigma within roundoff via trace consistency."""
    chart = CHARTS["invdepth"]
    rng, settings, orc, core, (xi0, Xs, ids, q0, Q, S) = make_pair(chart, N, seed=2024, sigma="init", fastRiccati=1, useDiscreteInnovationLift=0,
                                                                   initialPointVariance=9.0, measurementNoise=1.93)
    cam = euroc_camera()
    imu = random_imu(rng)
    orc.integrate_riccati_fast(imu, 0.05)
    core.integrate_riccati_fast(imu, 0.05, settings.input_gain_diag12(), settings.state_gain_diag8())
    S_prior = check_sigma(core, orc)
    mid, y = synth_measurement(rng, cam, ids, q0, Q, noise_px=1.0)
    orc.vision_update(cam, mid, y)
    core.vision_update(cam, mid, y, settings.measurementNoise**2, True, False)
    S_post = check_sigma(core, orc)
    check_state(core, orc)
    assert np.max(np.abs(S_post - S_post.T)) <= 1e-12 * np.max(np.abs(S_post))
    assert np.all(np.diag(S_post) <= np.diag(S_prior) * (1 + 1e-12))
    assert np.all(np.linalg.eigvalsh(0.5 * (S_post + S_post.T)) > 0)


@pytest.mark.parametrize("chart", list(CHARTS))
@pytest.mark.parametrize("N", [0, 3, 20, 45])
def test_compute_nees(chart, N):
    """eqf_compute_nees (chain factorisation of Sigma, n odd or even) vs VIO_eqf::computeNEES (VIO_eqf.cpp:153-170,
    dense LU inverse). The 'true' state holds more landmarks than the filter, in a different order."""
    rng, settings, orc, core, (xi0, Xs, ids, q0, Q, S) = make_pair(CHARTS[chart], N, seed=900 + N, cap=max(N, 4))
    s_e, ids_e, p_e = orc.state_estimate()
    true_sensor = s_e.copy()
    true_sensor[0:6] += rng.normal(size=6) * 0.01
    true_sensor[10:13] += rng.normal(size=3) * 0.05
    true_sensor[13:16] += rng.normal(size=3) * 0.05
    extra = 4
    true_ids = np.concatenate([ids_e, 100000 + np.arange(extra)]).astype(np.int32)
    true_p = np.concatenate([p_e * (1.0 + 0.02 * rng.normal(size=(N, 1))) + rng.normal(size=(N, 3)) * 0.05, rng.normal(size=(extra, 3)) + [0, 0, 5]])
    perm = rng.permutation(N + extra)
    true_ids, true_p = true_ids[perm], true_p[perm]
    nees_o = orc.compute_nees(true_sensor, true_ids, true_p)
    nees_g = core.compute_nees(true_sensor, true_ids, true_p)
    assert nees_o > 0
    assert abs(nees_g - nees_o) <= 1e-9 * nees_o
    # Sigma and X untouched
    check_sigma(core, orc, 0.0)
    with pytest.raises(EqfError):
        core.compute_nees(true_sensor, true_ids[:1], true_p[:1]) if N > 1 else (_ for _ in ()).throw(EqfError(-3, "n/a"))


@pytest.mark.parametrize("chart", list(CHARTS))
def test_speculative_tail_equals_stats_plus_update(chart):
    """eqf_stats_then_update: with no outlier candidate the state equals eqf_outlier_stats + eqf_vision_update bit for bit;
    with a threshold that one landmark exceeds the device cancels the queued update: nothing is modified, the statistics are
    returned, and the classic calls still work afterwards (oracle parity)."""
    from eqvio_amd.capi import OPT_SPECULATIVE

    N = 19
    rng, settings, orc, core, (xi0, Xs, ids, q0, Q, S) = make_pair(CHARTS[chart], N, seed=77, useDiscreteInnovationLift=0)
    core.set_option(OPT_SPECULATIVE, 2)  # always queue the tail (the default backs off for a frame after a cancellation, see below)
    twin = EqfCore(N, CHARTS[chart])
    twin.set_state(xi0, Xs, ids, q0, Q)
    twin.set_sigma(S)
    cam = default_camera()
    mid, y = synth_measurement(rng, cam, ids, q0, Q, noise_px=1.0, subset=rng.permutation(N)[:15])
    var = settings.measurementNoise**2
    # classic pair on the twin
    a0, p0, d0 = twin.outlier_stats(cam, mid, y)
    twin.vision_update(cam, mid, y, var, True, False)
    # speculative call, thresholds far away
    upd, a1, p1, d1 = core.stats_then_update(cam, mid, y, 1e8, 1e8, var, True, False)
    assert upd == 1
    assert np.array_equal(a0, a1) and np.array_equal(p0, p1) and np.array_equal(d0, d1)
    assert np.array_equal(core.get_sigma(), twin.get_sigma())
    for u, v in zip(core.get_state(), twin.get_state()):
        assert np.array_equal(u, v)
    orc.vision_update(cam, mid, y)
    check_sigma(core, orc)
    check_state(core, orc)
    # second frame: a threshold just below the largest probabilistic error cancels the tail on the device
    mid2, y2 = synth_measurement(rng, cam, ids, q0, Q, noise_px=1.0, subset=rng.permutation(N)[:12])
    S_before, st_before = core.get_sigma(), core.get_state()
    a_ref, p_ref, _ = twin.outlier_stats(cam, mid2, y2)
    thr = 0.999 * np.max(p_ref)
    upd, a2, p2, d2 = core.stats_then_update(cam, mid2, y2, 1e8, thr, var, True, False)
    assert upd == 0
    assert np.array_equal(a2, a_ref) and np.array_equal(p2, p_ref)
    assert np.array_equal(core.get_sigma(), S_before)
    for u, v in zip(core.get_state(), st_before):
        assert np.array_equal(u, v)
    # ... and exactly at the largest error (not exceeded: '>' in VIOFilter.cpp:324) it goes through
    upd, _, _, _ = core.stats_then_update(cam, mid2, y2, 1e8, np.max(p_ref), var, True, False)
    assert upd == 1
    # an id that is not in the state: not applicable, nothing computed
    S_now = core.get_sigma()
    upd, _, _, _ = core.stats_then_update(cam, np.array([10**6], np.int32), np.array([1.0, 2.0]), 1e8, 1e8, var, True, False)
    assert upd == -1 and np.array_equal(core.get_sigma(), S_now)
    twin.vision_update(cam, mid2, y2, var, True, False)
    assert np.array_equal(core.get_sigma(), twin.get_sigma())
    # default mode (1): a cancelled tail is followed by a statistics-only call (back-off), a frame without a candidate ends it
    core.set_option(OPT_SPECULATIVE, 1)
    mid3, y3 = synth_measurement(rng, cam, ids, q0, Q, noise_px=1.0, subset=rng.permutation(N)[:12])
    a_ref, p_ref, _ = twin.outlier_stats(cam, mid3, y3)
    upd, _, _, _ = core.stats_then_update(cam, mid3, y3, 1e8, 0.999 * np.max(p_ref), var, True, False)
    assert upd == 0  # cancelled on the device
    upd, a4, p4, _ = core.stats_then_update(cam, mid3, y3, 1e8, 1e8, var, True, False)
    assert upd == 0 and np.array_equal(p4, p_ref)  # backed off: statistics only, no candidate -> back-off over
    upd, _, _, _ = core.stats_then_update(cam, mid3, y3, 1e8, 1e8, var, True, False)
    assert upd == 1
    twin.vision_update(cam, mid3, y3, var, True, False)
    assert np.array_equal(core.get_sigma(), twin.get_sigma())


@pytest.mark.parametrize("N,M,cap,chart,star", [(40, 33, 11, "invdepth", True), (200, 180, 11, "invdepth", True), (256, 256, 11, "invdepth", True), (300, 257, 11, "invdepth", True),
                                              (512, 470, 11, "invdepth", True), (40, 33, 10**6, "invdepth", True), (200, 180, 10**6, "invdepth", True), (256, 256, 10**6, "invdepth", True),
                                              (300, 257, 10**6, "invdepth", True), (512, 470, 10**6, "invdepth", True), (200, 180, 10**6, "euclid", True),
                                              (200, 180, 10**6, "invdepth", False), (200, 180, 10**6, "euclid", False), (300, 200, 10**6, "invdepth", True), (500, 256, 11, "invdepth", True)])
def test_outlier_decision_in_one_workgroup_equals_the_two_launches(N, M, cap, chart, star):
    """eqf_stats_select_update up to 512 landmarks: statistics, VIOFilter::removeOutliers' decision (VIOFilter.cpp:304-364) and the masking of the discarded measurements
    are ONE launch of one workgroup (k_stats_select: statistics and output blocks on different waves, whole-wave ranking). Against the two launches
    (EQF_OPT_SELECT_ONE_WORKGROUP = 0: k_outlier_stats, k_select_outliers): same statistics, same discarded landmarks, same Sigma+ and state, bit for bit; and the
    discarded set is the reference's: absolute outliers first by absErr, then probabilistic ones by probErr, capped."""
    import ctypes as C

    from eqvio_amd.capi import OPT_LIVE_COLUMNS_FIRST, OPT_SELECT_ONE_WORKGROUP, OPT_SPECULATIVE

    CH = CHARTS[chart]
    rng, settings, orc, core, (xi0, Xs, ids, q0, Q, S) = make_pair(CH, N, seed=500 + N, cap=N)
    core.set_option(OPT_LIVE_COLUMNS_FIRST, 0)  # measurement order kept: the same Z as the two launches build
    twin = EqfCoreStats(xi0, Xs, ids, q0, Q, S, N, CH)
    twin.set_option(OPT_SELECT_ONE_WORKGROUP, 0)
    third = EqfCoreStats(xi0, Xs, ids, q0, Q, S, N, CH)  # default: the live columns in front, the factorisation ends behind them (up to 16 panels)
    cam = default_camera()
    mid, y = synth_measurement(rng, cam, ids, q0, Q, noise_px=1.0, subset=rng.permutation(N)[:M])
    bad = rng.choice(M, 9, replace=False)
    y.reshape(-1, 2)[bad] += rng.normal(size=(9, 2)) * 30.0  # gross outliers on top of the probabilistic ones
    y.reshape(-1, 2)[bad[1]] = y.reshape(-1, 2)[bad[0]]
    var = settings.measurementNoise**2
    a_ref, p_ref, _ = EqfCoreStats(xi0, Xs, ids, q0, Q, S, N, CH).outlier_stats(cam, mid, y)
    thr_abs = np.sort(a_ref[a_ref >= 0])[-6]  # five absolute outliers
    thr_prob = np.median(p_ref[a_ref >= 0])  # half of the rest are probabilistic candidates
    res = []
    for c in (core, twin):
        c.set_option(OPT_SPECULATIVE, 0)  # straight to the masked pipeline
        res.append(c.stats_select_update(cam, mid, y, thr_abs, thr_prob, cap, var, star, False))
    (u1, a1, p1, d1, r1), (u0, a0, p0, d0, r0) = res
    assert u1 == 1 and u0 == 1
    assert np.array_equal(a1, a0) and np.array_equal(p1, p0) and np.array_equal(d1, d0)
    assert np.array_equal(a1, a_ref) and np.array_equal(p1, p_ref)
    assert np.array_equal(r1, r0)
    # the reference's order, restated on the returned statistics
    meas = a1 >= 0
    absl = [i for i in np.argsort(-a1, kind="stable") if meas[i] and a1[i] > thr_abs]
    probl = [i for i in np.argsort(-p1, kind="stable") if meas[i] and not a1[i] > thr_abs and p1[i] > thr_prob]
    assert len(absl) == 5 and len(absl) + len(probl) > 11
    assert sorted((absl + probl)[:cap]) == list(r1)  # (without the cap: about half of the measured landmarks)
    assert core.N == N - len(r1) and twin.N == N - len(r1)
    assert np.array_equal(core.get_sigma(), twin.get_sigma())
    for u, v in zip(core.get_state(), twin.get_state()):
        assert np.array_equal(u, v)
    # EQF_OPT_LIVE_COLUMNS_FIRST: another column order of the same Z, the dead columns behind the last factorised panel - the same update up to rounding
    third.set_option(OPT_SPECULATIVE, 0)
    u2, a2, p2, d2, r2 = third.stats_select_update(cam, mid, y, thr_abs, thr_prob, cap, var, star, False)
    assert u2 == 1 and np.array_equal(a2, a1) and np.array_equal(p2, p1) and np.array_equal(r2, r1)
    used = C.c_long()
    assert third.lib.eqf_live_columns_stats(third.h, C.byref(used), 0) == 0
    assert used.value == (1 if 3 <= (2 * M + 31) // 32 <= 16 else 0)
    assert rel_fro(third.get_sigma(), core.get_sigma()) <= 1e-12
    for u, v in zip(third.get_state(), core.get_state()):
        np.testing.assert_allclose(u, v, rtol=1e-12, atol=1e-13)
    # ... and a look-ahead launch that gives up (time-out 0) is redone on the launch chain from the same live-first output blocks: every column, the dead ones too - the
    # bits of the launch that ended behind the live ones (W is exactly zero in a dead column either way)
    from eqvio_amd.capi import OPT_LA_TIMEOUT_US

    fourth = EqfCoreStats(xi0, Xs, ids, q0, Q, S, N, CH)
    fourth.set_option(OPT_SPECULATIVE, 0)
    fourth.set_option(OPT_LA_TIMEOUT_US, 0)
    u3, a3, p3, d3, r3 = fourth.stats_select_update(cam, mid, y, thr_abs, thr_prob, cap, var, star, False)
    assert u3 == 1 and np.array_equal(r3, r1)
    la, fb = C.c_long(), C.c_long()
    assert fourth.lib.eqf_lookahead_stats(fourth.h, C.byref(la), C.byref(fb), 0) == 0
    if used.value:
        assert la.value == 1 and fb.value == 1
        assert np.array_equal(fourth.get_sigma(), third.get_sigma())
        for u, v in zip(fourth.get_state(), third.get_state()):  # (Gamma is summed in another order on the chain: eqf_hip.h, eqf_lookahead_stats)
            np.testing.assert_allclose(u, v, rtol=1e-12, atol=1e-13)


def EqfCoreStats(xi0, Xs, ids, q0, Q, S, N, chart=COORD_INVDEPTH):
    c = EqfCore(N, chart)
    c.set_state(xi0, Xs, ids, q0, Q)
    c.set_sigma(S)
    return c


@pytest.mark.parametrize("chart", list(CHARTS))
@pytest.mark.parametrize("k,discrete", [(1, True), (10, True), (30, False), (45, True), (0, True)])
def test_propagate_fast_equals_riccati_plus_observer(chart, k, discrete):
    """eqf_propagate_fast = integrateRiccatiStateFast at the current X followed by the k observer steps (VIOFilter.cpp:134-192,
    fast branch); only the queueing order on the device differs (assembly inside the propagation kernel, observer blocks writing the
    second landmark buffer). k = 30 / 45 need two / three kernel-argument chunks of 20 steps; k = 0: no observer blocks at all."""
    N = 17
    rng, settings, orc, core, (xi0, Xs, ids, q0, Q, S) = make_pair(CHARTS[chart], N, seed=91)
    twin = EqfCore(N, CHARTS[chart])
    twin.set_state(xi0, Xs, ids, q0, Q)
    twin.set_sigma(S)
    imus = np.stack([random_imu(rng, bias_vel=True) for _ in range(k)]) if k else np.zeros((0, 13))
    dts = rng.uniform(0.002, 0.008, k)
    mean = (imus * dts[:, None]).sum(0) / dts.sum() if k else random_imu(rng, bias_vel=True)
    dt_total = float(dts.sum()) if k else 0.05
    Qd, Pd = settings.input_gain_diag12(), settings.state_gain_diag8()
    twin.integrate_riccati_fast(mean, dt_total, Qd, Pd)
    if k:
        twin.integrate_observer(imus, dts, discrete)
    core.propagate_fast(mean, dt_total, Qd, Pd, imus, dts, discrete)
    assert np.array_equal(core.get_sigma(), twin.get_sigma())
    for u, v in zip(core.get_state(), twin.get_state()):
        assert np.array_equal(u, v)
    orc.integrate_riccati_fast(mean, dt_total)
    for s_ in range(k):
        orc.integrate_observer(imus[s_], dts[s_], discrete)
    check_sigma(core, orc, 1e-12)
    check_state(core, orc)


@pytest.mark.parametrize("chart", list(CHARTS))
def test_staged_measurement_is_bit_identical_and_falls_back_when_stale(chart):
    """eqf_stage_measurement before the propagation: eqf_stats_then_update then reads the measurement from HBM (copied by a block of
    the propagation kernel) and must produce exactly what it produces without the hint; a hint that no longer matches (other
    pixels, other ids, landmark set changed in between) is ignored."""
    N = 21
    rng, settings, orc, core, (xi0, Xs, ids, q0, Q, S) = make_pair(CHARTS[chart], N, seed=91, useDiscreteInnovationLift=0)
    twin = EqfCore(N, CHARTS[chart])
    twin.set_state(xi0, Xs, ids, q0, Q)
    twin.set_sigma(S)
    cam = default_camera()
    var = settings.measurementNoise**2
    Qd, Pd = settings.input_gain_diag12(), settings.state_gain_diag8()
    k = 5
    imus = np.stack([random_imu(rng, stamp=0.005 * i) for i in range(k)])
    dts = rng.uniform(0.002, 0.006, k)
    mean = imus.mean(axis=0)

    def frame(c, hint_ids, hint_y, mid, y):
        if hint_ids is not None:
            c.stage_measurement(hint_ids, hint_y)
        c.propagate_fast(mean, float(dts.sum()), Qd, Pd, imus, dts, True)
        return c.stats_then_update(cam, mid, y, 1e8, 1e8, var, True, False)

    def same(a, b):
        assert np.array_equal(a.get_sigma(), b.get_sigma())
        for u, v in zip(a.get_state(), b.get_state()):
            assert np.array_equal(u, v)

    # 1. matching hint
    mid, y = synth_measurement(rng, cam, ids, q0, Q, noise_px=1.0, subset=rng.permutation(N)[:17])
    r0 = frame(core, mid, y, mid, y)
    r1 = frame(twin, None, None, mid, y)
    assert r0[0] == 1 and r1[0] == 1
    for u, v in zip(r0[1:], r1[1:]):
        assert np.array_equal(u, v)
    same(core, twin)
    # 2. hint with other pixels / other ids: ignored
    mid2, y2 = synth_measurement(rng, cam, ids, q0, Q, noise_px=1.0, subset=rng.permutation(N)[:15])
    frame(core, mid2, y2 + 0.5, mid2, y2)
    frame(twin, None, None, mid2, y2)
    same(core, twin)
    frame(core, mid, y, mid2, y2)
    frame(twin, None, None, mid2, y2)
    same(core, twin)
    # 3. landmark set changed between hint and update (indices shift): ignored
    order = np.argsort(ids)
    gone = int(order[0])
    keep = np.array([i for i in ids if i != ids[gone]], np.int32)
    mid3, y3 = synth_measurement(rng, cam, ids, q0, Q, noise_px=1.0)
    sel = np.isin(mid3, keep)
    mid3k, y3k = mid3[sel], y3.reshape(-1, 2)[sel].reshape(-1)
    core.stage_measurement(mid3k, y3k)
    core.propagate_fast(mean, float(dts.sum()), Qd, Pd, imus, dts, True)
    twin.propagate_fast(mean, float(dts.sum()), Qd, Pd, imus, dts, True)
    core.remove_landmarks(np.array([gone], np.int32))
    twin.remove_landmarks(np.array([gone], np.int32))
    a = core.stats_then_update(cam, mid3k, y3k, 1e8, 1e8, var, True, False)
    b = twin.stats_then_update(cam, mid3k, y3k, 1e8, 1e8, var, True, False)
    assert a[0] == 1 and b[0] == 1
    same(core, twin)
    # 4. a hint with an id that is not in the state is not an error and stages nothing
    core.stage_measurement(np.array([999999], np.int32), np.array([1.0, 2.0]))


@pytest.mark.parametrize("chart", list(CHARTS))
@pytest.mark.parametrize("between", ["observer", "dense_riccati"])
def test_output_blocks_of_the_propagation_kernel_are_dropped_when_the_state_moves_on(chart, between):
    """EQF_OPT_MEASURE_IN_PROPAGATE leaves C / yTilde evaluated at the Q_i the propagation kernel ended with. Anything that moves the state between that kernel and
    the update - further observer steps (eqf_integrate_observer), another propagation in any Riccati mode - must drop them: the update then evaluates the output blocks
    itself (eqf_measure_in_propagate_stats stays 0 for that frame) and follows the oracle (a stale cache gave a silently wrong update)."""
    import ctypes as C

    N = 40
    rng, settings, orc, core, (xi0, Xs, ids, q0, Q, S) = make_pair(CHARTS[chart], N, seed=17, useDiscreteInnovationLift=0)
    cam = default_camera()
    var = settings.measurementNoise**2
    Qd, Pd = settings.input_gain_diag12(), settings.state_gain_diag8()
    k = 5
    used = C.c_long()
    for frame in range(3):
        imus = np.stack([random_imu(rng, stamp=0.005 * i) for i in range(k)])
        dts = rng.uniform(0.002, 0.006, k)
        mean = imus.mean(axis=0)
        extra = np.stack([random_imu(rng, stamp=0.1 + 0.005 * i) for i in range(3)])
        edts = rng.uniform(0.002, 0.006, 3)
        st = core.get_state()
        mid, y = synth_measurement(rng, cam, ids, st[3], st[4], noise_px=0.5)
        core.stage_measurement(mid, y)
        core.propagate_fast(mean, float(dts.sum()), Qd, Pd, imus, dts, True)
        orc.integrate_riccati_fast(mean, float(dts.sum()))
        for s_ in range(k):
            orc.integrate_observer(imus[s_], dts[s_], True)
        if frame > 0:  # frame 0 names the camera; from frame 1 on the propagation kernel evaluates the output blocks - and what follows must invalidate them
            if between == "observer":
                core.integrate_observer(extra, edts, True)
                for s_ in range(3):
                    orc.integrate_observer(extra[s_], edts[s_], True)
            else:
                core.set_option(OPT_RICCATI_DENSE, 1)
                core.integrate_riccati_fast(extra[0], float(edts[0]), Qd, Pd)
                core.set_option(OPT_RICCATI_DENSE, 0)
                orc.integrate_riccati_fast(extra[0], float(edts[0]))
        upd, *_ = core.stats_then_update(cam, mid, y, 1e9, 1e9, var, True, False)
        assert upd == 1
        orc.vision_update(cam, mid, y)
        check_sigma(core, orc)
        check_state(core, orc)
        assert core.lib.eqf_measure_in_propagate_stats(core.h, C.byref(used), 0) == 0 and used.value == 0, frame


def test_nees_returns_a_number_when_sigma_is_not_numerically_spd():
    """The reference inverts Sigma by partial-pivot LU and returns a number whatever Sigma is (VIO_eqf.cpp:166-168); the device's
    Cholesky-type chain meets a non-positive pivot when Sigma is positive definite only up to rounding and must then fall back to
    an elimination with partial pivoting (on the device) instead of reporting EQF_E_NOT_SPD."""
    rng = np.random.default_rng(77)
    N = 20
    n = 21 + 3 * N
    chart = COORD_INVDEPTH
    settings = settings_for(chart)
    xi0, Xs, ids, q0, Q = reasonable_state(rng, N)
    V, _ = np.linalg.qr(rng.normal(size=(n, n)))
    lam = np.exp(rng.uniform(np.log(1e-3), np.log(10.0), n))
    lam[3] = -1e-9  # slightly indefinite: what rounding leaves of a zero eigenvalue
    S = (V * lam) @ V.T
    S = 0.5 * (S + S.T)
    core = EqfCore(N, chart)
    core.set_state(xi0, Xs, ids, q0, Q)
    core.set_sigma(S)
    orc = OracleFilter(settings)
    orc.set_eqf(xi0, Xs, ids, q0, Q, S)
    es, eids, ep = orc.state_estimate()
    ts = es.copy()
    ts[0:6] += rng.normal(size=6) * 1e-3
    ts[13:16] += rng.normal(size=3) * 1e-2
    tp = ep + rng.normal(size=ep.shape) * 1e-2
    assert core.nees_lu_fallbacks() == 0
    nees = core.compute_nees(ts, eids, tp)
    assert core.nees_lu_fallbacks() == 1
    ref = orc.compute_nees(ts, eids, tp)
    assert np.isfinite(nees) and abs(nees - ref) <= 1e-6 * abs(ref), (nees, ref)
    # an SPD matrix takes the factorisation path and leaves the counter alone
    lam[3] = 1e-3
    S2 = (V * lam) @ V.T
    core.set_sigma(0.5 * (S2 + S2.T))
    orc.set_eqf(xi0, Xs, ids, q0, Q, 0.5 * (S2 + S2.T))
    n2 = core.compute_nees(ts, eids, tp)
    assert core.nees_lu_fallbacks() == 1 and abs(n2 - orc.compute_nees(ts, eids, tp)) <= 1e-9 * abs(n2)
    # odd and even dimensions both work (the chain pads odd n)
    for N2 in (7, 8):
        n2_ = 21 + 3 * N2
        xi0b, Xsb, idsb, q0b, Qb = reasonable_state(rng, N2)
        Vb, _ = np.linalg.qr(rng.normal(size=(n2_, n2_)))
        lb = np.exp(rng.uniform(np.log(1e-2), np.log(5.0), n2_))
        lb[0] = -1e-8
        Sb = (Vb * lb) @ Vb.T
        Sb = 0.5 * (Sb + Sb.T)
        cb = EqfCore(N2, chart)
        cb.set_state(xi0b, Xsb, idsb, q0b, Qb)
        cb.set_sigma(Sb)
        ob = OracleFilter(settings)
        ob.set_eqf(xi0b, Xsb, idsb, q0b, Qb, Sb)
        esb, eidb, epb = ob.state_estimate()
        tpb = epb + rng.normal(size=epb.shape) * 1e-2
        v = cb.compute_nees(esb, eidb, tpb)
        r = ob.compute_nees(esb, eidb, tpb)
        assert cb.nees_lu_fallbacks() == 1 and abs(v - r) <= 1e-6 * abs(r), (N2, v, r)


@pytest.mark.parametrize("N,M", [(200, 200), (50, 50), (40, 40), (60, 33), (224, 224), (130, 97), (256, 250), (300, 270), (384, 384), (400, 390), (500, 500), (512, 512)])
def test_lookahead_factorisation_is_bit_identical_to_the_launch_chain(N, M):
    """EQF_OPT_LOOKAHEAD: the factorisation of [S ; T ; y^T] as ONE persistent kernel (owner workgroup on the pivot chain, one workgroup
    per block row following behind, hand-offs as write-through 8 KB tiles + one sequence-numbered flag per tile) against one launch per panel. Same tile arithmetic in
    the same order: W and Sigma must not change by a bit; Gamma is summed in another (fixed) order, so it and the lifted state agree to
    rounding. Sizes: 13 panels (the headline), 4, 3 (smallest eligible), a last panel of 2 columns, 14 panels (largest of the small
    instantiation), M < N with a ragged last panel, the 16-panel instantiation, and the two large ones (17 .. 24 and 25 .. 32 panels: N = 300 .. 512,
    the stress configuration N = 500 among them). A second update on the same context
    meets the first one's tiles and flags in the hand-off buffers (the sequence number in the flags tells them apart)."""
    from eqvio_amd.capi import OPT_LA_HOME, OPT_LOOKAHEAD

    rng, settings, orc, core, (xi0, Xs, ids, q0, Q, S) = make_pair(CHARTS["invdepth"], N, seed=N + M, useDiscreteInnovationLift=0)
    cam = default_camera()
    mid, y = synth_measurement(rng, cam, ids, q0, Q, noise_px=1.0, subset=np.sort(rng.permutation(N)[:M]))
    rows, m = 2 * M + 21 + 3 * N + 1, 2 * M
    outs = []
    cores = []
    # the look-ahead kernel in its classic placement and (round 5, up to 16 panels) in the HOME placement - owner and S half-rows on one XCD, their hand-offs through its
    # L2 (2 = also while other contexts share the device, as they do in this test)
    for la, home in ((0, 0), (1, 0), (1, 2)):
        c = EqfCore(N, CHARTS["invdepth"])
        c.set_state(xi0, Xs, ids, q0, Q)
        c.set_sigma(S)
        c.set_option(OPT_LOOKAHEAD, la)
        c.set_option(OPT_LA_HOME, home)
        c.vision_update(cam, mid, y, settings.measurementNoise**2, True, False)
        outs.append((c.get_sigma(), c.get_state(), c.last_gamma(), c.debug_get_W(rows, m)[m:]))
        cores.append(c)
    for S1, st1, g1, W1 in outs[1:]:
        assert np.array_equal(W1, outs[0][3])
        assert np.array_equal(S1, outs[0][0])
        assert np.linalg.norm(g1 - outs[0][2]) <= 1e-12 * np.linalg.norm(outs[0][2])
        for u, v in zip(st1, outs[0][1]):
            assert np.allclose(u, v, rtol=1e-12, atol=1e-13)
    assert np.array_equal(outs[1][2], outs[2][2])  # Gamma is deterministic
    # second frame on every context (propagation in between keeps the problem well posed)
    imu = random_imu(rng)
    y2 = y + rng.normal(size=y.shape) * 0.5
    S2 = []
    for c in cores:
        c.integrate_riccati_fast(imu, 0.05, settings.input_gain_diag12(), settings.state_gain_diag8())
        c.vision_update(cam, mid, y2, settings.measurementNoise**2, True, False)
        S2.append(c.get_sigma())
    assert rel_fro(S2[1], S2[0]) <= 1e-11 and np.array_equal(S2[1], S2[2])
    if N <= 60:
        orc.vision_update(cam, mid, y)
        assert rel_fro(outs[1][0], orc.get_sigma()) <= 1e-9


@pytest.mark.parametrize("chart", ["euclid", "invdepth"])
@pytest.mark.parametrize("N,drop,knew", [(200, [3, 17, 100, 199], 5), (60, [], 3), (48, [0, 47], 16), (20, list(range(1, 20)), 30), (300, [0, 150, 299], 9), (16, [], 1), (61, [5, 60], 4)])
def test_held_landmarks_pass_through_the_propagation(chart, N, drop, knew):
    """eqf_add_landmarks_held: the frame's new landmarks appended IN FRONT of the propagation (the reference appends them behind it, VIOFilter.cpp:217 behind :196) and
    passed through it untouched - created by the propagation kernel itself (rows of Sigma, planes, output blocks; no append pass). Against the reference's order
    (remove, propagate, append) on a second context: Sigma, the state and the following update must agree bit for bit - with and without removals in the same frame, a
    new landmark that fills a tile exactly, more new landmarks than old ones, above 256 landmarks; with an entry point in between (the held landmarks become an ordinary
    append and still pass through); and the refusals (another variance, ordinary append while landmarks are held, the other propagation entry points)."""
    import ctypes as C

    from eqvio_amd.capi import EqfError, load_eqf_lib

    lib = load_eqf_lib()
    rng, settings, orc, core, (xi0, Xs, ids, q0, Q, S) = make_pair(CHARTS[chart], N, seed=7 * N + knew, cap=N + knew + 8, useDiscreteInnovationLift=0)
    cam = default_camera()
    nsteps = 45 if N == 61 else 6  # 45: more IMU samples than one chunk of observer steps holds (a dropped camera frame): the further chunks are launches of their own
    imus = [random_imu(rng) for _ in range(nsteps)]
    dts = [0.003] * nsteps
    mean = np.mean(imus, axis=0)
    new_ids = (np.arange(knew) * 13 + int(np.max(ids)) + 5).astype(np.int32)
    new_p = rng.uniform(-1, 1, (knew, 3)) + np.array([0, 0, 5.0])
    var = 1.7
    Qg, Pg = settings.input_gain_diag12(), settings.state_gain_diag8()

    def fresh():
        c = EqfCore(N + knew + 8, CHARTS[chart])
        c.set_state(xi0, Xs, ids, q0, Q)
        c.set_sigma(S)
        return c

    # reference order
    ref = fresh()
    if drop:
        ref.remove_landmarks(np.array(drop, np.int32))
    ref.propagate_fast(mean, sum(dts), Qg, Pg, imus, dts, True)
    ref.add_landmarks(new_ids, new_p, var)
    # held, created by the propagation kernel
    a = fresh()
    if drop:
        a.remove_landmarks(np.array(drop, np.int32))
    assert a.add_landmarks_held(new_ids, new_p, var)
    a.propagate_fast(mean, sum(dts), Qg, Pg, imus, dts, True)
    used = C.c_long()
    assert lib.eqf_hold_stats(a.h, C.byref(used), 0) == 0 and used.value == 1
    # held, with an entry point in between (they are appended by an ordinary pass, and still pass through)
    b = fresh()
    if drop:
        b.remove_landmarks(np.array(drop, np.int32))
    assert b.add_landmarks_held(new_ids[:1], new_p[:1], var)
    if knew > 1:
        assert b.add_landmarks_held(new_ids[1:], new_p[1:], var)  # a second call joins the first
    # refusals while landmarks are held
    assert not b.add_landmarks_held(np.array([99991], np.int32), np.array([[0.0, 0.0, 4.0]]), var + 1.0)  # another variance
    with pytest.raises(EqfError):
        b.add_landmarks(np.array([99992], np.int32), np.array([[0.0, 0.0, 4.0]]), var)
    with pytest.raises(EqfError):
        b.integrate_riccati_fast(imus[0], 0.01, Qg, Pg)
    assert b.get_sigma().shape == (21 + 3 * (N - len(drop) + knew),) * 2  # an entry point: ordinary append
    b.propagate_fast(mean, sum(dts), Qg, Pg, imus, dts, True)
    assert lib.eqf_hold_stats(b.h, C.byref(used), 0) == 0 and used.value == 0
    Sr, str_ = ref.get_sigma(), ref.get_state()
    for c in (a, b):
        assert np.array_equal(c.get_sigma(), Sr)
        for u, v in zip(c.get_state(), str_):
            assert np.array_equal(np.asarray(u), np.asarray(v))
    # the update that follows reads everything the propagation left (Sigma, elements, origin points, chart constants)
    mid, y = synth_measurement(rng, cam, str_[2], str_[3], str_[4], noise_px=1.0)
    for c in (ref, a, b):
        c.vision_update(cam, mid, y, settings.measurementNoise**2, True, False)
    for c in (a, b):
        assert np.array_equal(c.get_sigma(), ref.get_sigma())
        for u, v in zip(c.get_state(), ref.get_state()):
            assert np.array_equal(np.asarray(u), np.asarray(v))
    # ... and a second propagation reads what only a propagation reads of a landmark: its chart constants (computed where the landmark was created)
    for c in (ref, a, b):
        c.propagate_fast(mean, sum(dts), Qg, Pg, imus, dts, True)
    for c in (a, b):
        assert np.array_equal(c.get_sigma(), ref.get_sigma())
        for u, v in zip(c.get_state(), ref.get_state()):
            assert np.array_equal(np.asarray(u), np.asarray(v))
    # a held landmark that is removed again is gone; the others still pass through
    if knew >= 2:
        d, r2 = fresh(), fresh()
        assert d.add_landmarks_held(new_ids, new_p, var)
        d.remove_landmarks(np.array([N], np.int32))  # the first held one
        d.propagate_fast(mean, sum(dts), Qg, Pg, imus, dts, True)
        r2.propagate_fast(mean, sum(dts), Qg, Pg, imus, dts, True)
        r2.add_landmarks(new_ids[1:], new_p[1:], var)
        assert np.array_equal(d.get_sigma(), r2.get_sigma())
        for u, v in zip(d.get_state(), r2.get_state()):
            assert np.array_equal(np.asarray(u), np.asarray(v))


def test_remove_unmeasured_landmarks_and_the_id_table_under_turnover():
    """eqf_remove_unmeasured_landmarks (VIOFilter::removeOldLandmarks, VIOFilter.cpp:280-302, in one call) and the sorted (id, index) table behind it, which
    eqf_add_landmarks / eqf_remove_landmarks keep up to date instead of re-sorting: 60 rounds of random turnover with ids in NO particular order in the state (and a
    stretch with ascending ids), every round checked against a plain numpy restatement - the indices removed, the ids left, and the id -> landmark mapping of a
    measurement (through the outlier statistics: a measured landmark has absErr >= 0, an unmeasured one -1)."""
    import ctypes as C

    from eqvio_amd.capi import load_eqf_lib

    lib = load_eqf_lib()
    rng = np.random.default_rng(2024)
    cam = default_camera()
    for ascending in (False, True):
        N = 40
        rng2, settings, orc, core, (xi0, Xs, ids, q0, Q, S) = make_pair(CHARTS["invdepth"], N, seed=11, cap=96)
        ids = (np.sort(ids) if ascending else ids).astype(np.int32)
        core.set_state(xi0, Xs, ids, q0, Q)
        core.set_sigma(S)
        cur = list(ids)
        next_id = int(max(cur)) + 1
        for rnd in range(60):
            # a measurement of a random subset (ascending ids, as a VisionMeasurement's are)
            keep = sorted(int(v) for v in rng.choice(cur, size=max(1, len(cur) - int(rng.integers(0, 6))), replace=False))
            meas = np.array(keep, np.int32)
            removed = np.zeros(len(cur) + 1, np.int32)
            n = C.c_int()
            assert lib.eqf_remove_unmeasured_landmarks(core.h, meas.ctypes.data_as(C.POINTER(C.c_int)), len(meas), removed.ctypes.data_as(C.POINTER(C.c_int)), C.byref(n)) == 0
            expect = [i for i, v in enumerate(cur) if v not in set(keep)]
            assert list(removed[: n.value]) == expect, (ascending, rnd)
            cur = [v for v in cur if v in set(keep)]
            assert list(core.get_state()[2]) == cur
            # append a few landmarks with ids that do not ascend (or do, in the second pass)
            k = int(rng.integers(0, 5))
            if k:
                new_ids = np.arange(next_id, next_id + k, dtype=np.int32)
                next_id += k
                if not ascending:
                    new_ids = (new_ids * 7919) % 100003 + 1000  # scattered, distinct
                    new_ids = np.array([v for v in new_ids if v not in set(cur)], np.int32)
                    k = len(new_ids)
                if k:
                    core.add_landmarks(new_ids, rng.uniform(-1, 1, (k, 3)) + np.array([0, 0, 5.0]), 1.0)
                    cur += [int(v) for v in new_ids]
            # the mapping of a measurement of every second landmark
            st = core.get_state()
            sub = np.sort(np.array(cur)[:: 2]).astype(np.int32)
            pos = {v: i for i, v in enumerate(cur)}
            y = np.zeros((len(sub), 2))
            for j, v in enumerate(sub):
                p = st[3][pos[int(v)]]
                y[j] = [cam.fx * p[0] / p[2] + cam.cx, cam.fy * p[1] / p[2] + cam.cy]
            absErr, probErr, depth2 = core.outlier_stats(cam, sub, y.reshape(-1))
            measured = set(int(v) for v in sub)
            for i, v in enumerate(cur):
                assert (absErr[i] != -1.0) == (v in measured), (ascending, rnd, i)
    # ids that do not ascend are refused (the caller takes its general route)
    bad = np.array([5, 3], np.int32)
    n = C.c_int()
    removed = np.zeros(200, np.int32)
    assert lib.eqf_remove_unmeasured_landmarks(core.h, bad.ctypes.data_as(C.POINTER(C.c_int)), 2, removed.ctypes.data_as(C.POINTER(C.c_int)), C.byref(n)) == -3


@pytest.mark.parametrize("chart", ["euclid", "invdepth"])
@pytest.mark.parametrize("N,drop", [(200, [3, 17, 18, 19, 100, 199]), (60, [0]), (60, [59]), (47, list(range(20, 44))), (300, [0, 1, 2, 63, 64, 65, 127, 128, 150, 299]), (33, list(range(1, 33))), (512, list(range(0, 512, 3)))])
def test_removed_landmarks_leave_inside_the_propagation_kernel(chart, N, drop):
    """EQF_OPT_GATHER_IN_PROPAGATE: a record of removals only is applied by eqf_propagate_fast's kernel itself (it reads the old positions, writes the new ones) instead of
    a compaction pass in front of it. Against the same calls with the option off: Sigma, the landmark elements, the origin points and the next update must agree bit for bit;
    first / last / a block of 24 / all but one landmark removed; above 256 landmarks (several tiles per workgroup); the record is NOT taken over when something was appended
    or when no observer steps ride along (then the ordinary pass runs, same result)."""
    import ctypes as C

    from eqvio_amd.capi import OPT_GATHER_IN_PROPAGATE, load_eqf_lib

    lib = load_eqf_lib()
    rng, settings, orc, core, (xi0, Xs, ids, q0, Q, S) = make_pair(CHARTS[chart], N, seed=5 * N + len(drop), useDiscreteInnovationLift=0)
    cam = default_camera()
    imus = [random_imu(rng) for _ in range(5)]
    dts = [0.004] * 5
    mean = np.mean(imus, axis=0)
    keep = np.array([i for i in range(N) if i not in set(drop)])
    outs = []
    for opt in (0, 1):
        c = EqfCore(N, CHARTS[chart])
        c.set_state(xi0, Xs, ids, q0, Q)
        c.set_sigma(S)
        c.set_option(OPT_GATHER_IN_PROPAGATE, opt)
        c.remove_landmarks(np.array(drop, np.int32))
        c.propagate_fast(mean, sum(dts), settings.input_gain_diag12(), settings.state_gain_diag8(), imus, dts, True)
        used = C.c_long()
        assert lib.eqf_gather_stats(c.h, C.byref(used), 0) == 0
        assert used.value == (1 if opt else 0), (opt, used.value)
        trail = [c.get_sigma(), c.get_state()]
        outs.append((c, trail))
    (c0, t0), (c1, t1) = outs
    assert np.array_equal(t0[0], t1[0])
    for u, v in zip(t0[1], t1[1]):
        assert np.array_equal(np.asarray(u), np.asarray(v))
    assert t0[0].shape == (21 + 3 * len(keep),) * 2
    # the same measurement on both contexts: the update reads the buffers the propagation left current (Sigma, landmark elements, origin points, chart constants)
    ids_k, q0_k, Q_k = t1[1][2], t1[1][3], t1[1][4]
    mid, y = synth_measurement(rng, cam, ids_k, q0_k, Q_k, noise_px=1.0)
    for c in (c0, c1):
        c.vision_update(cam, mid, y, settings.measurementNoise**2, True, False)
    assert np.array_equal(c0.get_sigma(), c1.get_sigma())
    for u, v in zip(c0.get_state(), c1.get_state()):
        assert np.array_equal(np.asarray(u), np.asarray(v))
    # a record with an appended landmark, and a propagation without observer steps: the ordinary pass, same results on both
    if len(keep) >= 3:
        new_id = np.array([int(np.max(ids)) + 7], np.int32)
        new_p = np.array([[0.1, -0.2, 4.0]])
        for c in (c0, c1):
            c.remove_landmarks(np.array([1], np.int32))
            c.add_landmarks(new_id, new_p, 1.5)
            c.propagate_fast(mean, sum(dts), settings.input_gain_diag12(), settings.state_gain_diag8(), imus, dts, True)
            c.remove_landmarks(np.array([0], np.int32))
            c.integrate_riccati_fast(imus[0], 0.01, settings.input_gain_diag12(), settings.state_gain_diag8())
        used = C.c_long()
        assert lib.eqf_gather_stats(c1.h, C.byref(used), 0) == 0 and used.value == 1
        assert np.array_equal(c0.get_sigma(), c1.get_sigma())
        for u, v in zip(c0.get_state(), c1.get_state()):
            assert np.array_equal(np.asarray(u), np.asarray(v))


@pytest.mark.parametrize("N,M", [(330, 330), (400, 371), (500, 500), (512, 512)])
def test_large_state_kernel_forms_are_bit_identical(N, M):
    """Round 5, above 256 landmarks: the propagation kernel's workgroups take several tiles of a block row each (EQF_OPT_TILES_PER_WORKGROUP; both assembly forms: the
    stand-alone Riccati step reads the assembled A / B terms, eqf_propagate_fast assembles them in the kernel). Every entry is the same sum in the same order as with one
    tile per workgroup: Sigma must not change by a bit over two frames, whatever the number of tiles per workgroup."""
    from eqvio_amd.capi import OPT_TILES_PER_WORKGROUP

    rng, settings, orc, core, (xi0, Xs, ids, q0, Q, S) = make_pair(CHARTS["invdepth"], N, seed=3 * N + M, useDiscreteInnovationLift=0)
    cam = default_camera()
    mid, y = synth_measurement(rng, cam, ids, q0, Q, noise_px=1.0, subset=np.sort(rng.permutation(N)[:M]))
    imus = [random_imu(rng) for _ in range(4)]
    dts = [0.005] * 4
    mean = np.mean(imus, axis=0)
    y2 = y + rng.normal(size=y.shape) * 0.5
    outs = []
    for tpw in (0, 1, 2, 5):
        c = EqfCore(N, CHARTS["invdepth"])
        c.set_state(xi0, Xs, ids, q0, Q)
        c.set_sigma(S)
        c.set_option(OPT_TILES_PER_WORKGROUP, tpw)
        trail = []
        c.integrate_riccati_fast(imus[0], 0.02, settings.input_gain_diag12(), settings.state_gain_diag8())
        trail.append(c.get_sigma())
        c.vision_update(cam, mid, y, settings.measurementNoise**2, True, False)
        trail.append(c.get_sigma())
        c.propagate_fast(mean, sum(dts), settings.input_gain_diag12(), settings.state_gain_diag8(), imus, dts, True)
        trail.append(c.get_sigma())
        c.vision_update(cam, mid, y2, settings.measurementNoise**2, True, False)
        trail.append(c.get_sigma())
        outs.append(trail)
    for trail in outs[1:]:
        for k, (a, b) in enumerate(zip(trail, outs[0])):
            assert np.array_equal(a, b), (N, k, np.abs(a - b).max())
    assert np.all(np.isfinite(outs[0][-1])) and np.array_equal(outs[0][-1], outs[0][-1].T)


def test_lookahead_factorisation_soak():
    """The hand-offs of the persistent kernel under repetition: fresh contexts (zeroed buffers, sequence 1) and one context reused
    (every word of the previous launch still in place), every result compared bit by bit with the launch chain's."""
    from eqvio_amd.capi import OPT_LA_HOME, OPT_LOOKAHEAD

    # (300, 270) and (500, 500): the 17 .. 32-panel instantiation, whose half-rows run a look-ahead of their own (la_row2: LDS double buffers by panel parity,
    # pair counters, flags raised under a later round trip) - fewer repetitions, they are 10 x the work
    for N, M, reps, fresh in ((200, 200, 61, 20), (40, 40, 61, 20), (60, 33, 61, 20), (300, 270, 25, 8), (500, 500, 25, 8)):
        rng, settings, orc, core, (xi0, Xs, ids, q0, Q, S) = make_pair(CHARTS["invdepth"], N, seed=N + M, useDiscreteInnovationLift=0)
        cam = default_camera()
        mid, y = synth_measurement(rng, cam, ids, q0, Q, noise_px=1.0, subset=np.sort(rng.permutation(N)[:M]))
        ref = None
        kept = EqfCore(N, CHARTS["invdepth"])
        for it in range(reps):
            c = EqfCore(N, CHARTS["invdepth"]) if it <= fresh else kept
            c.set_state(xi0, Xs, ids, q0, Q)
            c.set_sigma(S)
            c.set_option(OPT_LOOKAHEAD, 0 if it == 0 else 1)
            c.set_option(OPT_LA_HOME, 2 if it % 2 else 0)  # HOME placement (up to 16 panels) and classic placement in turns
            c.vision_update(cam, mid, y, settings.measurementNoise**2, True, False)
            Sg = c.get_sigma()
            if it == 0:
                ref = Sg
            else:
                assert np.array_equal(Sg, ref), (N, it, np.abs(Sg - ref).max())


@pytest.mark.parametrize("chart", list(CHARTS))
@pytest.mark.parametrize("N", [0, 1, 9, 40])
def test_riccati_discrete(chart, N):
    """eqf_integrate_riccati_discrete (row a7) against the oracle's integrateRiccatiStateDiscrete: same central differences (h = cbrt(eps)) of
    a0Discrete, 43 sensor-level evaluations on the host + one lane per landmark on the device. Tolerance = the differencing's rounding noise."""
    rng, settings, orc, core, _ = make_pair(CHARTS[chart], max(N, 0), seed=300 + N, cap=max(N, 1)) if N > 0 else make_pair(CHARTS[chart], 0, seed=300, cap=1)
    for rep in range(2):
        imu = random_imu(rng, bias_vel=True)
        dt = float(rng.uniform(0.004, 0.02))
        core.integrate_riccati_discrete(imu, dt, settings.input_gain_diag12(), settings.state_gain_diag8())
        orc.integrate_riccati_discrete(imu, dt)
        Sg, So = core.get_sigma(), orc.get_sigma()
        assert rel_fro(Sg, So) <= 5e-9, rel_fro(Sg, So)
        core.integrate_observer(imu[None, :], np.array([dt]), True)
        orc.integrate_observer(imu, dt, True)
