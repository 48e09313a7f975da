"""Free-running forms of the filter-level parity runs (SURVEY.md section 8(d) "Parity definition": teacher-forced AND free-running with decision-flip
detection). The teacher-forced forms (tests/test_gpu_filter_headline.py, flat 1e-9 per frame) compare one frame's arithmetic at a time. Here the device
filter is never reset: a last-bit difference of frame 1 is carried - and, where the configuration is ill conditioned, amplified - through every later
frame, in the device exactly as in a second CPU evaluation of the reference. What two fp64 evaluations of the same run can agree to is therefore MEASURED in
the same test (a second oracle in the other dense arithmetic) and the device is held to max(1e-9, 2 x that distance); every landmark bookkeeping / outlier
decision must coincide with the reference's (a flipped decision fails the test at once: `parity` asserts identical landmark sets)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from eqvio_amd.capi import PreparedFrames, SimSettings, SimulationDataServer, VIOFilter  # noqa: E402
from oracle_binding import OracleFilter  # noqa: E402
from test_gpu_filter_headline import _frame_mix_world, counters  # noqa: E402

pytestmark = pytest.mark.gpu
TOL = 1e-9


def test_frame_mix_shipped_thresholds_N200_free_running():
    """bench.frame_mix's second workload: wave world (about 9 of 200 tracked features change per frame), the shipped EuRoC outlier thresholds / retention /
    point variance (EQVIO_config_EuRoC_stationary.yaml:26-32). Almost every frame has an outlier candidate: the speculative tail is cancelled, speculation
    backs off and the frames take statistics -> k_select_outliers -> masked update -> k_reshape, at N ~ 200, in step with the oracle's reference order
    (src/VIOFilter.cpp:304-364)."""
    N, s, world, frames, sensor, ids, p = _frame_mix_world()
    from oracle_binding import ARITH_AS_WRITTEN, ARITH_EFFICIENT
    from run_configs import parity

    flt = VIOFilter(s, max_landmarks=N + 120, sensor=sensor, ids=ids, p=p, time=0.0)
    # Point variance 129.9 against 1.93 px of pixel noise: cond(S) is large and the freshly initialised landmarks move by metres in their first update.
    # What two fp64 evaluations of the same frame can agree to is measured, not assumed: a second oracle in the other dense arithmetic ("efficient":
    # Cholesky; the first is "as written": LU inverse, K evaluated twice, VIO_eqf.cpp:116-131). The device is held to max(1e-9, 2 x that floor).
    orc = OracleFilter(s, sensor, ids, p, 0.0)
    orc2 = OracleFilter(s, sensor, ids, p, 0.0)
    orc.set_arithmetic(ARITH_AS_WRITTEN)
    orc2.set_arithmetic(ARITH_EFFICIENT)
    prepared = PreparedFrames(world.cam, *bench.flatten_frames(frames))
    dims, worst, worst_floor = [], 0.0, 0.0
    for f, (imus, stamp, mid, y) in enumerate(frames[:8]):
        assert flt.run_prepared(prepared, f, 1) == 1
        for o in (orc, orc2):
            for k_ in range(len(imus)):
                o.process_imu(imus[k_])
            o.process_vision(stamp, world.cam, mid, y)
        es, eS = parity(flt, orc)  # asserts identical landmark sets: every outlier decision of the device matches the reference order
        fs_, fS_ = parity(orc2, orc)
        assert es <= max(TOL, 2.0 * fs_) and eS <= max(TOL, 2.0 * fS_), (f, es, fs_, eS, fS_)
        worst, worst_floor = max(worst, es, eS), max(worst_floor, fs_, fS_)
        dims.append((flt.sigma_dim() - 21) // 3)
    assert worst <= 1e-7, (worst, worst_floor)
    k = counters(flt)
    assert k["sel_frames"] >= 3 and k["sel_discarded"] >= 3, k  # the device took the outlier decision at this size
    assert k["la_launches"] >= 6 and k["la_fallbacks"] == 0, k
    assert min(dims) >= 100, dims


def test_config2_euroc_structured_sine_50_landmarks_free_running():
    """BASELINE.json configs[1] stand-in (tests/run_configs.py config 2, promoted): the C++ SimulationDataServer on the sine trajectory, 50 tracked
    features, the shipped EuRoC settings' structure with simulator-consistent values, the filter adding and dropping landmarks by itself
    (main_opt-like), >= 100 frames in lockstep with the oracle."""
    from run_configs import euroc_settings, parity, sim_consistent

    fs = sim_consistent(euroc_settings(), measurementNoise=1.0)
    sim = SimSettings.defaults(duration=6.0, trajectory="sine", numPoints=4000, wallDistance=3.0, numWalls=6, randomSeed=1, maxFeatures=50, outputNoise=1, inputNoise=0)
    srv = SimulationDataServer(sim, fs)
    fs.cameraOffset[:] = srv.camera_offset()
    s0, ids0, p0 = srv.true_state(0.0, True)
    from oracle_binding import ARITH_AS_WRITTEN, ARITH_EFFICIENT

    flt = VIOFilter(fs, max_landmarks=2 * sim.maxFeatures + 64, sensor=s0, ids=ids0[:0], p=p0[:0], time=0.0)
    orc = OracleFilter(fs, s0, ids0[:0], p0[:0], 0.0)
    orc2 = OracleFilter(fs, s0, ids0[:0], p0[:0], 0.0)  # the other dense arithmetic: what a free-running fp64 filter can agree to after 100+ frames
    orc.set_arithmetic(ARITH_AS_WRITTEN)
    orc2.set_arithmetic(ARITH_EFFICIENT)
    frames, worst_state, worst_sigma, floor_state, floor_sigma, seen = 0, 0.0, 0.0, 0.0, 0.0, set()
    while srv.next_measurement_type() != srv.NONE:
        if srv.next_measurement_type() == srv.IMU:
            imu = srv.get_imu()
            flt.process_imu(imu)
            orc.process_imu(imu)
            orc2.process_imu(imu)
            continue
        stamp, ids, y = srv.get_vision()
        flt.process_vision(stamp, srv.cam, ids, y)
        orc.process_vision(stamp, srv.cam, ids, y)
        orc2.process_vision(stamp, srv.cam, ids, y)
        es, eS = parity(flt, orc)  # asserts identical landmark sets
        fs_, fS_ = parity(orc2, orc)
        worst_state, worst_sigma = max(worst_state, es), max(worst_sigma, eS)
        floor_state, floor_sigma = max(floor_state, fs_), max(floor_sigma, fS_)
        seen |= set(ids.tolist())
        frames += 1
    assert frames >= 100 and len(seen) > 60, (frames, len(seen))  # landmarks really entered and left
    print(f"config 2 stand-in, {frames} frames: device vs oracle state {worst_state:.2e} Sigma {worst_sigma:.2e}; oracle vs oracle {floor_state:.2e} / {floor_sigma:.2e}")
    assert worst_state <= max(TOL, 2.0 * floor_state) and worst_sigma <= max(TOL, 2.0 * floor_sigma), (worst_state, floor_state, worst_sigma, floor_sigma)
    assert worst_state <= 1e-8 and worst_sigma <= 1e-9
    assert counters(flt)["la_launches"] >= 90


@pytest.mark.parametrize("chart", [0, 1])  # Euclidean, InvDepth (the Normal chart's own free-running run: tests/test_gpu_normal_chart.py)
def test_discrete_state_matrix_free_running(chart):
    """useDiscreteStateMatrix (integrateRiccatiStateDiscrete, VIO_eqf.cpp:93-103) with device and oracle each on their OWN state for 20 frames - the
    teacher-forced form (tests/test_gpu_filter.py::test_discrete_state_matrix_filter_run) bounds one frame's error and cannot see drift. The bound is not 1e-9: A_d is a
    central difference with h = 6e-6 on both sides, whose rounding noise (eps / h = 4e-11 per entry, amplified by A_d Sigma A_d^T over ~10 IMU steps per frame) is the
    reference's own; measured 6e-9 per frame teacher forced. Free running it may accumulate - it must not grow faster than linearly in the frames: 2e-7 after 20."""
    from eqvio_amd.capi import VIOFilter
    from eqvio_amd.simworld import SimWorld
    from test_gpu_filter import compare, sim_settings

    world = SimWorld(seed=13, num_points=500, max_features=14, trajectory="wave", noise_px=0.3)
    settings = sim_settings(chart, fastRiccati=0, useDiscreteStateMatrix=1)
    ids0, _ = world.vision(0.0)
    sensor, ids, p = world.true_state(0.0, ids0)
    orc = OracleFilter(settings, sensor, ids, p, 0.0)
    flt = VIOFilter(settings, max_landmarks=48, sensor=sensor, ids=ids, p=p, time=0.0)
    for k, (imus, stamp, mid, y) in enumerate(world.frames(20)):
        for s in range(len(imus)):
            orc.process_imu(imus[s])
            flt.process_imu(imus[s])
        orc.process_vision(stamp, world.cam, mid, y)
        flt.process_vision(stamp, world.cam, mid, y)
        compare(flt, orc, 1e-8 * (k + 1))
    flt.close()
