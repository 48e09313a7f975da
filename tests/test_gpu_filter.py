"""Filter-level parity: the host VIOFilter mirror driving the HIP core (include/eqvio_filter.h) against the oracle's
VIOFilter on identical IMU + feature-track inputs, free running, with landmarks entering and leaving the view
(removeOldLandmarks / addNewLandmarks / removeOutliers decisions must coincide; SURVEY.md §7 "hard parts")."""
import numpy as np
import pytest

from eqvio_amd.capi import COORD_EUCLIDEAN, COORD_INVDEPTH, Settings, VIOFilter
from oracle_binding import OracleFilter, se3_log_dist
from simworld import SimWorld
from util import rel_fro

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


def test_discrete_state_matrix_is_refused_loudly():
    """Row a7 (integrateRiccatiStateDiscrete) is oracle-only: the product refuses it instead of falling back."""
    world = SimWorld(seed=1, num_points=100, max_features=5, trajectory="hover")
    settings = sim_settings(COORD_EUCLIDEAN, fastRiccati=0, useDiscreteStateMatrix=1)
    ids0, y0 = world.vision(0.0)
    sensor, ids, p = world.true_state(0.0, ids0)
    flt = VIOFilter(settings, max_landmarks=16, sensor=sensor, ids=ids, p=p, time=0.0)
    flt.process_imu(world.imu(0.0))
    with pytest.raises(RuntimeError):
        flt.process_vision(0.05, world.cam, ids0, y0)


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
