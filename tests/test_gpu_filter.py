"""Filter-level parity: the host VIOFilter mirror driving the HIP core (include/eqvio_filter.h) against the oracle's
VIOFilter on identical IMU + feature-track inputs, free running, with landmarks entering and leaving the view
(removeOldLandmarks / addNewLandmarks / removeOutliers decisions must coincide; SURVEY.md §7 "hard parts")."""
import numpy as np
import pytest

from eqvio_amd.capi import COORD_EUCLIDEAN, COORD_INVDEPTH, Settings, VIOFilter
from oracle_binding import OracleFilter, se3_log_dist
from eqvio_amd.simworld import SimWorld
from util import rel_fro, teacher_force

pytestmark = pytest.mark.gpu
TOL = 1e-9


def sim_settings(chart, **kw):
    s = Settings.defaults()
    s.coordinateChoice = chart
    s.fastRiccati = 1
    s.useDiscreteInnovationLift = 0
    s.useMedianDepth = 1
    s.initialSceneDepth = 4.0
    s.initialPointVariance = 4.0
    s.measurementNoise = 1.5
    s.cameraOffset[:] = [0.5, -0.5, 0.5, -0.5, 0, 0, 0]
    for k, v in kw.items():
        setattr(s, k, v)
    return s


def compare(flt, orc, tol=TOL):
    s_g, ids_g, p_g = flt.state_estimate()
    s_o, ids_o, p_o = orc.state_estimate()
    assert np.array_equal(ids_g, ids_o), (ids_g, ids_o)
    assert se3_log_dist(s_g[6:13], s_o[6:13]) <= tol * max(1.0, np.linalg.norm(s_o[10:13]))
    assert se3_log_dist(s_g[16:23], s_o[16:23]) <= tol
    assert np.max(np.abs(s_g[13:16] - s_o[13:16])) <= tol * max(1.0, np.max(np.abs(s_o[13:16])))
    assert np.max(np.abs(s_g[0:6] - s_o[0:6])) <= tol
    if len(p_o):
        assert np.max(np.linalg.norm(p_g - p_o, axis=1) / np.maximum(1.0, np.linalg.norm(p_o, axis=1))) <= tol
    Sg, So = flt.get_sigma(), orc.get_sigma()
    assert Sg.shape == So.shape
    assert rel_fro(Sg, So) <= tol, rel_fro(Sg, So)


@pytest.mark.parametrize("chart", [COORD_EUCLIDEAN, COORD_INVDEPTH])
@pytest.mark.parametrize("median_depth", [1, 0])
def test_wave_world_free_running(chart, median_depth):
    """main_sim-like run (src/main_sim.cpp:128-184 without augmentLandmarkStates): features enter and leave."""
    world = SimWorld(seed=3, num_points=1500, max_features=30, trajectory="wave", noise_px=0.3)
    settings = sim_settings(chart, useMedianDepth=median_depth)
    ids0, _ = world.vision(0.0)
    sensor, ids, p = world.true_state(0.0, ids0)
    orc = OracleFilter(settings, sensor, ids, p, 0.0)
    flt = VIOFilter(settings, max_landmarks=64, sensor=sensor, ids=ids, p=p, time=0.0)
    turnover = 0
    prev = set(ids0.tolist())
    for imus, stamp, mid, y in world.frames(40):
        for s in range(len(imus)):
            orc.process_imu(imus[s])
            flt.process_imu(imus[s])
        orc.process_vision(stamp, world.cam, mid, y)
        flt.process_vision(stamp, world.cam, mid, y)
        turnover += len(set(mid.tolist()) ^ prev)
        prev = set(mid.tolist())
        assert flt.get_time() == orc.get_time() == stamp
        compare(flt, orc)
    assert turnover > 10  # the test really exercised landmark removal and addition


def test_uninitialised_filter_self_initialises_from_imu():
    """VIOFilter(settings) + first IMU sample sets attitude from gravity (VIOFilter.cpp:31-41, 65-78), N grows from 0."""
    world = SimWorld(seed=5, num_points=800, max_features=12, trajectory="hover", noise_px=0.2)
    settings = sim_settings(COORD_INVDEPTH)
    orc = OracleFilter(settings)
    flt = VIOFilter(settings, max_landmarks=32)
    assert not flt.is_initialised()
    # a vision frame before any IMU is silently ignored (VIOFilter.cpp:198-199)
    ids0, y0 = world.vision(0.0)
    flt.process_vision(0.0, world.cam, ids0, y0)
    orc.process_vision(0.0, world.cam, ids0, y0)
    assert flt.state_estimate()[1].size == 0
    for imus, stamp, mid, y in world.frames(8, t0=0.01):
        for s in range(len(imus)):
            orc.process_imu(imus[s])
            flt.process_imu(imus[s])
        orc.process_vision(stamp, world.cam, mid, y)
        flt.process_vision(stamp, world.cam, mid, y)
        compare(flt, orc)
    assert flt.is_initialised() and flt.state_estimate()[1].size == 12


def test_outlier_rejection_decisions_match():
    """Tight thresholds so that removeOutliers (VIOFilter.cpp:304-364) actually fires; the kept sets must be identical."""
    world = SimWorld(seed=9, num_points=600, max_features=25, trajectory="hover", noise_px=0.4)
    settings = sim_settings(COORD_INVDEPTH, outlierThresholdAbs=6.0, outlierThresholdProb=4.0, featureRetention=0.5, initialPointVariance=0.05)
    ids0, _ = world.vision(0.0)
    sensor, ids, p = world.true_state(0.0, ids0)
    orc = OracleFilter(settings, sensor, ids, p, 0.0)
    flt = VIOFilter(settings, max_landmarks=64, sensor=sensor, ids=ids, p=p, time=0.0)
    rng = np.random.default_rng(0)
    removed_any = False
    for imus, stamp, mid, y in world.frames(10):
        y = y.copy()
        bad = rng.choice(len(mid), 3, replace=False)
        y.reshape(-1, 2)[bad] += rng.normal(size=(3, 2)) * 25.0  # gross outliers
        for s in range(len(imus)):
            orc.process_imu(imus[s])
            flt.process_imu(imus[s])
        orc.process_vision(stamp, world.cam, mid, y)
        flt.process_vision(stamp, world.cam, mid, y)
        compare(flt, orc)
        removed_any |= flt.state_estimate()[1].size < len(mid)
    assert removed_any


@pytest.mark.parametrize("chart", [COORD_EUCLIDEAN, COORD_INVDEPTH])
def test_outlier_decision_on_the_device_matches_the_reference_order(chart):
    """With a fixed initial depth (both shipped dataset configurations) nothing of a frame needs the host between the outlier statistics and the
    update, and once the speculative tail has been cancelled (an outlier candidate in the frame) the filter queues statistics -> decision on the
    device (k_select_outliers: VIOFilter.cpp:304-364, the cap of (1 - featureRetention) * #features included) -> update with the discarded
    landmarks' measurements masked -> removal of those landmarks, with ONE host wait per frame. The reference erases first and updates then;
    both orders give the same state (an unmeasured landmark can be marginalised before or after the update): frame by frame against the oracle,
    gross outliers in every frame, features entering and leaving, retention 0.9 so that the cap decides in some frames."""
    import ctypes as C

    from eqvio_amd.capi import load_eqf_lib

    world = SimWorld(seed=17, num_points=1500, max_features=40, trajectory="wave", noise_px=0.4)
    settings = sim_settings(chart, useMedianDepth=0, outlierThresholdAbs=6.0, outlierThresholdProb=4.0, featureRetention=0.9, initialPointVariance=0.05)
    ids0, _ = world.vision(0.0)
    sensor, ids, p = world.true_state(0.0, ids0)
    orc = OracleFilter(settings, sensor, ids, p, 0.0)
    flt = VIOFilter(settings, max_landmarks=64, sensor=sensor, ids=ids, p=p, time=0.0)
    rng = np.random.default_rng(1)
    capped = 0
    for f, (imus, stamp, mid, y) in enumerate(world.frames(30)):
        y = y.copy()
        n_bad = 2 + (f % 5)  # (1 - 0.9) * 40 = 4: frames with 5 or 6 gross outliers exceed the cap
        bad = rng.choice(len(mid), n_bad, replace=False)
        y.reshape(-1, 2)[bad] += rng.normal(size=(n_bad, 2)) * 25.0
        capped += n_bad > int((1.0 - 0.9) * len(mid))
        for s in range(len(imus)):
            orc.process_imu(imus[s])
            flt.process_imu(imus[s])
        orc.process_vision(stamp, world.cam, mid, y)
        flt.process_vision(stamp, world.cam, mid, y)
        compare(flt, orc)
    frames, discarded = C.c_long(), C.c_long()
    assert load_eqf_lib().eqf_selection_stats(flt.core_handle(), C.byref(frames), C.byref(discarded), 0) == 0
    assert frames.value >= 20 and discarded.value >= 40 and capped >= 5  # the device took the decision in (almost) every frame


@pytest.mark.parametrize("chart", [COORD_EUCLIDEAN, COORD_INVDEPTH])
def test_template_config_accurate_riccati(chart):
    """20 features, the template's fastRiccati: false (EQVIO_config_template.yaml eqf block):
    one accurate Riccati step per IMU sample, interleaved with the observer steps (VIOFilter.cpp:128-139)."""
    world = SimWorld(seed=5, num_points=1000, max_features=20, trajectory="wave", noise_px=0.3)
    settings = sim_settings(chart, fastRiccati=0)
    ids0, _ = world.vision(0.0)
    sensor, ids, p = world.true_state(0.0, ids0)
    orc = OracleFilter(settings, sensor, ids, p, 0.0)
    flt = VIOFilter(settings, max_landmarks=32, sensor=sensor, ids=ids, p=p, time=0.0)
    for imus, stamp, mid, y in world.frames(25):
        for s in range(len(imus)):
            orc.process_imu(imus[s])
            flt.process_imu(imus[s])
        orc.process_vision(stamp, world.cam, mid, y)
        flt.process_vision(stamp, world.cam, mid, y)
        compare(flt, orc)


def test_filter_grows_past_its_initial_capacity():
    """Settings::maxLandmarks is an initial size, not a limit (the reference's filter has none). With removeLostLandmarks = 0 (a supported
    reference setting, VIOFilter.cpp:207-211) the landmark count only grows: a filter created for 16 landmarks ends the run with several
    times that, frame by frame in step with the oracle (no frame abandoned half done on a capacity error)."""
    world = SimWorld(seed=5, num_points=1500, max_features=24, trajectory="wave", noise_px=0.3)
    settings = sim_settings(COORD_INVDEPTH, removeLostLandmarks=0)
    ids0, _ = world.vision(0.0)
    sensor, ids, p = world.true_state(0.0, ids0[:10])
    orc = OracleFilter(settings, sensor, ids, p, 0.0)
    flt = VIOFilter(settings, max_landmarks=16, sensor=sensor, ids=ids, p=p, time=0.0)
    for imus, stamp, mid, y in world.frames(40):
        for s in range(len(imus)):
            orc.process_imu(imus[s])
            flt.process_imu(imus[s])
        orc.process_vision(stamp, world.cam, mid, y)
        flt.process_vision(stamp, world.cam, mid, y)
        compare(flt, orc)
    assert len(flt.state_estimate()[1]) > 40


@pytest.mark.parametrize("chart", [COORD_EUCLIDEAN, COORD_INVDEPTH, 2])  # 2 = COORD_NORMAL: numerically differentiated twice over in the oracle
def test_discrete_state_matrix_filter_run(chart):
    """Row a7: useDiscreteStateMatrix (integrateRiccatiStateDiscrete, VIO_eqf.cpp:93-103; the mode the reference's own statistical test runs,
    test_FilterStatistics.cpp:110,132) against the oracle's filter, TEACHER FORCED: every frame starts from the oracle's state, so the bound below is one
    frame's. It is not 1e-9 and cannot be: A_d comes from central differences with h = cbrt(eps) = 6e-6 on both sides (EqFMatrices.cpp:24-41), whose
    rounding noise eps / h = 4e-11 per entry of A_d is the REFERENCE's own (two evaluations of the reference's differencing - another compiler's FMA
    contraction is enough - differ by it; tests/test_indep_restatement.py measures the same distance between the two CPU restatements), and A_d Sigma A_d^T
    over the frame's ~10 IMU steps carries it into Sigma. Measured per frame: Sigma 6e-9 / 3e-9 (Euclidean / InvDepth), 2e-8 with the Normal chart, whose
    chart maps the reference differentiates numerically a second time (VIOState.cpp:391-401)."""
    world = SimWorld(seed=13, num_points=500, max_features=14, trajectory="wave", noise_px=0.3)
    settings = sim_settings(chart, fastRiccati=0, useDiscreteStateMatrix=1)
    ids0, _ = world.vision(0.0)
    sensor, ids, p = world.true_state(0.0, ids0)
    orc = OracleFilter(settings, sensor, ids, p, 0.0)
    flt = VIOFilter(settings, max_landmarks=48, sensor=sensor, ids=ids, p=p, time=0.0)
    for imus, stamp, mid, y in world.frames(6):
        for s in range(len(imus)):
            orc.process_imu(imus[s])
            flt.process_imu(imus[s])
        orc.process_vision(stamp, world.cam, mid, y)
        flt.process_vision(stamp, world.cam, mid, y)
        compare(flt, orc, 3e-8 if chart == 2 else 1e-8)
        teacher_force(flt, orc)


def test_feature_predictions_match():
    """getFeaturePredictions (VIOFilter.cpp:247-252) -> VIO_eqf::predictState (VIO_eqf.cpp:139-151) -> integrateSystemFunction
    (VIOState.cpp:28-68): the state estimate pushed through the buffered IMU samples up to a stamp, then projected. The
    prediction is requested between frames (as the tracker front end would, main_opt.cpp:200-203)."""
    world = SimWorld(seed=9, num_points=800, max_features=25, trajectory="wave", noise_px=0.2)
    settings = sim_settings(COORD_INVDEPTH, useFeaturePredictions=1)
    ids0, _ = world.vision(0.0)
    sensor, ids, p = world.true_state(0.0, ids0)
    orc = OracleFilter(settings, sensor, ids, p, 0.0)
    flt = VIOFilter(settings, max_landmarks=64, sensor=sensor, ids=ids, p=p, time=0.0)
    n_checked = 0
    for imus, stamp, mid, y in world.frames(12):
        for s in range(len(imus)):
            orc.process_imu(imus[s])
            flt.process_imu(imus[s])
        # the IMU buffer now reaches the next image stamp: predict the features there, before processing it
        ig, yg = flt.get_feature_predictions(world.cam, stamp)
        io, yo = orc.get_feature_predictions(world.cam, stamp)
        assert np.array_equal(ig, io) and len(ig) > 0
        np.testing.assert_allclose(yg, yo, rtol=0, atol=1e-7)  # pixels: 1e-9 relative of O(100) px coordinates
        n_checked += len(ig)
        orc.process_vision(stamp, world.cam, mid, y)
        flt.process_vision(stamp, world.cam, mid, y)
    assert n_checked > 100
    off = VIOFilter(sim_settings(COORD_INVDEPTH), max_landmarks=64, sensor=sensor, ids=ids, p=p, time=0.0)
    assert len(off.get_feature_predictions(world.cam, 0.1)[0]) == 0  # disabled by default, like the reference


def test_remove_invalid_landmarks_fires():
    """VIO_eqf::removeInvalidLandmarks (VIO_eqf.cpp:213-223): landmarks whose SOT(3) scale left (1e-8, 1e8] are dropped after the update.
    The scale never gets there in a healthy run, so the test plants it: two landmarks out of range (1e-9, 2e8), two far from one
    but valid (1e-4, 1e4). A landmark's pixel depends on its direction only, so the update itself stays ordinary; ids, the compacted
    Sigma and the state must follow the oracle's filter, which runs the reference's own loop. Scales of 1e8 stretch the dynamic range of
    C and Sigma in the frame they take part in; a second oracle in the other dense arithmetic ("efficient" vs "as written") measures
    the floor that leaves to any two fp64 evaluations, and the device is held to that floor."""
    import ctypes as C

    from eqvio_amd.capi import load_eqf_lib
    from oracle_binding import ARITH_AS_WRITTEN, ARITH_EFFICIENT

    world = SimWorld(seed=21, num_points=900, max_features=24, trajectory="hover", noise_px=0.3)
    settings = sim_settings(COORD_INVDEPTH)
    ids0, _ = world.vision(0.0)
    sensor, ids, p = world.true_state(0.0, ids0)
    orc = OracleFilter(settings, sensor, ids, p, 0.0)
    orc2 = OracleFilter(settings, sensor, ids, p, 0.0)
    orc.set_arithmetic(ARITH_AS_WRITTEN)
    orc2.set_arithmetic(ARITH_EFFICIENT)
    flt = VIOFilter(settings, max_landmarks=64, sensor=sensor, ids=ids, p=p, time=0.0)
    lib = load_eqf_lib()
    removed = None
    for f, (imus, stamp, mid, y) in enumerate(world.frames(8)):
        for s in range(len(imus)):
            orc.process_imu(imus[s])
            orc2.process_imu(imus[s])
            flt.process_imu(imus[s])
        if f == 3:
            for o in (orc, orc2):
                xi0, Xs, lid, q0, Q = o.get_eqf()
                n_before = len(lid)
                Q[2, 4], Q[5, 4], Q[7, 4], Q[11, 4] = 1e-9, 2e8, 1e-4, 1e4
                removed = {int(lid[2]), int(lid[5])}
                o.set_eqf(xi0, Xs, lid, q0, Q, o.get_sigma(), o.get_time())
            gx, gX, gid, gq0, gQ = flt.get_eqf()
            assert np.array_equal(gid, lid)
            gQ[:, 4] = Q[:, 4]
            f64p = C.POINTER(C.c_double)
            a = [np.ascontiguousarray(v, dtype=np.float64) for v in (gx, gX, gq0, gQ)]
            gid = np.ascontiguousarray(gid, dtype=np.int32)
            assert lib.eqf_set_state(flt.core_handle(), a[0].ctypes.data_as(f64p), a[1].ctypes.data_as(f64p), gid.ctypes.data_as(C.POINTER(C.c_int)), a[2].ctypes.data_as(f64p),
                                     a[3].ctypes.data_as(f64p), len(gid)) == 0
        for o in (orc, orc2):
            o.process_vision(stamp, world.cam, mid, y)
        flt.process_vision(stamp, world.cam, mid, y)
        if f == 3:
            kept = set(flt.state_estimate()[1].tolist())
            assert removed.isdisjoint(kept) and len(kept) == n_before - 2
            assert set(orc.state_estimate()[1].tolist()) == kept
            assert flt.sigma_dim() == 21 + 3 * (n_before - 2)
        if f < 3:
            compare(flt, orc)
        else:
            assert np.array_equal(flt.state_estimate()[1], orc.state_estimate()[1])
            floor = rel_fro(orc2.get_sigma(), orc.get_sigma())
            err = rel_fro(flt.get_sigma(), orc.get_sigma())
            assert err <= max(TOL, 2.0 * floor), (f, err, floor)
            s_g, _, p_g = flt.state_estimate()
            s_o, _, p_o = orc.state_estimate()
            s_2, _, p_2 = orc2.state_estimate()
            fl = max(np.max(np.abs(s_2 - s_o)), np.max(np.abs(p_2 - p_o) / np.maximum(1.0, np.abs(p_o))))
            assert max(np.max(np.abs(s_g - s_o)), np.max(np.abs(p_g - p_o) / np.maximum(1.0, np.abs(p_o)))) <= max(TOL, 2.0 * fl)
    assert removed is not None


def test_remove_invalid_landmarks_boundaries_exact():
    """The interval is (1e-8, 1e8]: a <= 1e-8 and a > 1e8 are invalid (VIO_eqf.cpp:216). Core level, no update in between: decisions on the
    boundary values themselves and their fp64 neighbours, Sigma compaction bit-exact."""
    from eqvio_amd.capi import EqfCore
    from util import random_spd, reasonable_state

    rng = np.random.default_rng(4)
    N = 9
    xi0, Xs, ids, q0, Q = reasonable_state(rng, N, shuffle_ids=True)
    Q[1, 4] = 1e-8                       # invalid (<=)
    Q[2, 4] = np.nextafter(1e-8, 1.0)    # valid
    Q[4, 4] = 1e8                        # valid (the interval is closed there)
    Q[6, 4] = np.nextafter(1e8, np.inf)  # invalid
    Q[7, 4] = 0.0                        # invalid
    S = random_spd(rng, 21 + 3 * N)
    core = EqfCore(N, COORD_EUCLIDEAN)
    core.set_state(xi0, Xs, ids, q0, Q)
    core.set_sigma(S)
    assert core.remove_invalid_landmarks() == 3
    _, _, ids2, q02, Q2 = core.get_state()
    keep = np.array([0, 2, 3, 4, 5, 8])
    assert np.array_equal(ids2, ids[keep]) and np.array_equal(q02, q0[keep]) and np.array_equal(Q2[:, 4], Q[keep, 4])
    rows = np.concatenate([np.arange(21)] + [21 + 3 * k + np.arange(3) for k in keep])
    assert np.array_equal(core.get_sigma(), S[np.ix_(rows, rows)])
    assert core.remove_invalid_landmarks() == 0
