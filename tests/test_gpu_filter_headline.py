"""The EXACT path bench.py times, in front of the oracle at the benchmarked sizes.

bench.py drives `VIOFilter` (the host mirror) through eqvio_filter_run_prepared -> eqf_stage_measurement + eqf_propagate_fast (k_propagate_main<double, true>: fused
assembly + observer blocks + measurement staging) + eqf_stats_then_update (k_build_Z<double, true>: measurement fusion + statistics row + speculative cancel,
k_chol_lookahead, k_lift, k_syrk_sub). The kernel-level tests reach those instantiations at N <= 19 only; here they meet the oracle at N = 200 and N = 500 on
bench.build_workload's own hover world with bench.eurocish_settings(), through the same C-ABI call bench.py uses, frame by frame (src/VIOFilter.cpp:194-241).
Also here: the `frame_mix` wave world with the shipped EuRoC thresholds at N ~ 200 (k_select_outliers + masked update + k_reshape at size) and BASELINE
config 2 (EuRoC-structured settings, sine trajectory, 50 landmarks, >= 100 frames, promoted from tests/run_configs.py), both TEACHER FORCED: every frame
starts from the oracle's state and is held to a flat 1e-9. Their free-running forms (which also measure how a configuration's conditioning amplifies
last-bit differences over a run) live in tests/test_gpu_free_running.py."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from eqvio_amd.capi import PreparedFrames, SimSettings, SimulationDataServer, VIOFilter, load_eqf_lib  # noqa: E402
from oracle_binding import OracleFilter  # noqa: E402
from eqvio_amd.simworld import SimWorld  # noqa: E402
from test_gpu_filter import compare  # noqa: E402
from util import teacher_force  # noqa: E402

pytestmark = pytest.mark.gpu
TOL = 1e-9


def counters(flt):
    lib = load_eqf_lib()
    a, b, c, d, e, f, g = (C.c_long() for _ in range(7))
    assert lib.eqf_speculation_stats(flt.core_handle(), C.byref(a), C.byref(b), C.byref(c), 0) == 0
    assert lib.eqf_lookahead_stats(flt.core_handle(), C.byref(d), C.byref(e), 0) == 0
    assert lib.eqf_selection_stats(flt.core_handle(), C.byref(f), C.byref(g), 0) == 0
    return dict(calls=a.value, queued=b.value, cancelled=c.value, la_launches=d.value, la_fallbacks=e.value, sel_frames=f.value, sel_discarded=g.value)


def run_lockstep(world, frames, settings, N, n_frames, tol=TOL, max_landmarks=None):
    """bench.rank_pass's filter and input containers; one eqvio_filter_run_prepared call per frame so that the oracle can be compared in between."""
    mk = lambda s, sensor, ids, p, t: VIOFilter(s, max_landmarks=max_landmarks or N, sensor=sensor, ids=ids, p=p, time=t)  # noqa: E731
    flt = bench.make_filter(world, settings, N, None, frames, mk)
    orc = bench.make_filter(world, settings, N, None, frames, lambda s, sensor, ids, p, t: OracleFilter(s, sensor, ids, p, t))
    prepared = PreparedFrames(world.cam, *bench.flatten_frames(frames[:n_frames]))
    worst = 0.0
    for f in range(n_frames):
        imus, stamp, mid, y = frames[f]
        assert flt.run_prepared(prepared, f, 1) == 1
        for s in range(len(imus)):
            orc.process_imu(imus[s])
        orc.process_vision(stamp, world.cam, mid, y)
        assert flt.get_time() == orc.get_time() == stamp
        compare(flt, orc, tol)
    return flt, orc, worst


def test_headline_path_N200_against_the_oracle():
    """The workload of the bench line itself (seed of rank 0), 6 frames in lockstep. Every frame must have gone through the one-round-trip path: tail
    queued speculatively and never cancelled, factorisation on the look-ahead kernel."""
    N = 200
    import gc

    gc.collect()  # (filters of earlier tests that are still registered on the device would switch the HOME placement off)
    world, frames = bench.build_workload(seed=100, n_frames=7, N=N)
    flt, orc, _ = run_lockstep(world, frames, bench.eurocish_settings(), N, 6)
    k = counters(flt)
    assert k["calls"] == k["queued"] == 6 and k["cancelled"] == 0, k
    assert k["la_launches"] == 6 and k["la_fallbacks"] == 0, k
    assert flt.sigma_dim() == 21 + 3 * N
    # the HOME placement leans on an empirically learnt block -> XCD mapping; a driver that deals the blocks differently sends every launch back to the
    # classic placement (or the chain) silently. With the device to itself - this filter is the only registered context - every look-ahead launch must have been a HOME one.
    lib = load_eqf_lib()
    hx, hl = C.c_int(), C.c_long()
    assert lib.eqf_lookahead_home(flt.core_handle(), C.byref(hx), C.byref(hl)) == 0
    if lib.eqf_device_to_itself(flt.core_handle()) == 1 and hx.value >= 0:
        assert hl.value == k["la_launches"], (hx.value, hl.value, k)
    # EQF_OPT_EARLY_DOORBELL (round 6): every one of these updates succeeds, so the host must have taken each from the look-ahead kernel's own doorbell or - when the lift's
    # rang before the host looked - from the lift's; at this size (the lift a kernel boundary + 4 us behind) nearly all come early
    er = C.c_long()
    assert lib.eqf_early_doorbell_stats(flt.core_handle(), C.byref(er), 0) == 0
    assert 0 <= er.value <= 6


def test_headline_path_N500_against_the_oracle():
    """BASELINE config 3 through the filter (bench.py --landmarks 500): the 32-panel ring instantiation of the look-ahead kernel, two frames."""
    N = 500
    world, frames = bench.build_workload(seed=100, n_frames=3, N=N)
    flt, orc, _ = run_lockstep(world, frames, bench.eurocish_settings(), N, 2)
    k = counters(flt)
    assert k["calls"] == k["queued"] == 2 and k["cancelled"] == 0 and k["la_launches"] == 2 and k["la_fallbacks"] == 0, k


def test_headline_path_N50_against_the_oracle():
    """BASELINE config 2's size through the same path (bench.py --landmarks 50): 4 panels, the smallest look-ahead instantiation in use."""
    N = 50
    world, frames = bench.build_workload(seed=100, n_frames=13, N=N)
    flt, orc, _ = run_lockstep(world, frames, bench.eurocish_settings(), N, 12)
    k = counters(flt)
    assert k["queued"] == 12 and k["cancelled"] == 0 and k["la_launches"] == 12, k


def _frame_mix_world():
    """bench.frame_mix's second workload: wave world (about 9 of 200 tracked features change per frame), the shipped EuRoC outlier thresholds / retention /
    point variance (EQVIO_config_EuRoC_stationary.yaml:26-32)."""
    N = 200
    s = bench.eurocish_settings()
    s.outlierThresholdAbs, s.outlierThresholdProb, s.featureRetention, s.initialPointVariance = 4.852186665580312, 0.03229809583062128, 0.18594708334486176, 129.90415638150924
    world = SimWorld(seed=321, num_points=2500, max_features=N, trajectory="wave", noise_px=0.5)
    frames = list(world.frames(9))
    ids0 = frames[0][2]
    sensor, ids, p = world.true_state(0.0, ids0)
    p = p * (1.0 + 0.05 * np.random.default_rng(7).normal(size=(len(ids), 1)))
    return N, s, world, frames, sensor, ids, p


def test_frame_mix_shipped_thresholds_N200_teacher_forced():
    """Almost every frame of this workload has an outlier candidate: the speculative tail is cancelled, speculation backs off and the frames take statistics ->
    k_select_outliers -> masked update -> k_reshape, at N ~ 200, in step with the oracle's reference order (src/VIOFilter.cpp:304-364). TEACHER FORCED
    (SURVEY.md section 8(d)): every frame starts from the oracle's (xi0, X, Sigma), so that one frame's arithmetic is compared at a time - flat 1e-9, no
    conditioning allowance. The free-running form of the same run: tests/test_gpu_free_running.py."""
    N, s, world, frames, sensor, ids, p = _frame_mix_world()
    from run_configs import parity

    flt = VIOFilter(s, max_landmarks=N + 120, sensor=sensor, ids=ids, p=p, time=0.0)
    orc = OracleFilter(s, sensor, ids, p, 0.0)  # the reference's arithmetic as written: LU inverse, K evaluated twice (VIO_eqf.cpp:116-131)
    prepared = PreparedFrames(world.cam, *bench.flatten_frames(frames))
    dims = []
    for f, (imus, stamp, mid, y) in enumerate(frames[:8]):
        assert flt.run_prepared(prepared, f, 1) == 1
        for k_ in range(len(imus)):
            orc.process_imu(imus[k_])
        orc.process_vision(stamp, world.cam, mid, y)
        es, eS = parity(flt, orc)  # asserts identical landmark sets: every outlier decision of the device matches the reference order
        assert es <= TOL and eS <= TOL, (f, es, eS)
        teacher_force(flt, orc)
        dims.append((flt.sigma_dim() - 21) // 3)
    k = counters(flt)
    assert k["sel_frames"] >= 3 and k["sel_discarded"] >= 3, k  # the device took the outlier decision at this size
    assert k["la_launches"] >= 6 and k["la_fallbacks"] == 0, k
    assert min(dims) >= 100, dims


def test_config2_euroc_structured_sine_50_landmarks_teacher_forced():
    """BASELINE.json configs[1] stand-in (tests/run_configs.py config 2, promoted): the C++ SimulationDataServer on the sine trajectory, 50 tracked
    features, the shipped EuRoC settings' structure with simulator-consistent values, the filter adding and dropping landmarks by itself
    (main_opt-like), >= 100 frames, each one started from the oracle's state (teacher forced): flat 1e-9."""
    from run_configs import euroc_settings, parity, sim_consistent

    fs = sim_consistent(euroc_settings(), measurementNoise=1.0)
    sim = SimSettings.defaults(duration=6.0, trajectory="sine", numPoints=4000, wallDistance=3.0, numWalls=6, randomSeed=1, maxFeatures=50, outputNoise=1, inputNoise=0)
    srv = SimulationDataServer(sim, fs)
    fs.cameraOffset[:] = srv.camera_offset()
    s0, ids0, p0 = srv.true_state(0.0, True)
    flt = VIOFilter(fs, max_landmarks=2 * sim.maxFeatures + 64, sensor=s0, ids=ids0[:0], p=p0[:0], time=0.0)
    orc = OracleFilter(fs, s0, ids0[:0], p0[:0], 0.0)
    frames, worst_state, worst_sigma, seen = 0, 0.0, 0.0, set()
    while srv.next_measurement_type() != srv.NONE:
        if srv.next_measurement_type() == srv.IMU:
            imu = srv.get_imu()
            flt.process_imu(imu)
            orc.process_imu(imu)
            continue
        stamp, ids, y = srv.get_vision()
        flt.process_vision(stamp, srv.cam, ids, y)
        orc.process_vision(stamp, srv.cam, ids, y)
        es, eS = parity(flt, orc)  # asserts identical landmark sets
        worst_state, worst_sigma = max(worst_state, es), max(worst_sigma, eS)
        teacher_force(flt, orc)
        seen |= set(ids.tolist())
        frames += 1
    assert frames >= 100 and len(seen) > 60, (frames, len(seen))  # landmarks really entered and left
    print(f"config 2 stand-in, {frames} frames teacher forced: device vs oracle state {worst_state:.2e} Sigma {worst_sigma:.2e}")
    assert worst_state <= TOL and worst_sigma <= TOL, (worst_state, worst_sigma)
    assert counters(flt)["la_launches"] >= 90


def test_measurement_edited_in_place_after_it_was_built():
    """ADVICE r2 (measurement cache) + round 3: the staging of the measurement ahead of the propagation (eqf_stage_measurement) reads the host mirror's cached
    flat copy WITHOUT the validating walk over the std::map (a hint; 1.7 us off the host path to the propagation's launch). A caller who writes new pixel
    values into the public map of an already built VisionMeasurement (eqvio_frames_edit_pixel) must still get exactly what a measurement built with those
    values gives: the device-side staged copy is compared with the (validated) measurement of the update call and dropped."""
    N = 60
    world, frames = bench.build_workload(seed=7, n_frames=6, N=N)
    settings = bench.eurocish_settings()
    mk = lambda s, sensor, ids, p, t: VIOFilter(s, max_landmarks=N, sensor=sensor, ids=ids, p=p, time=t)  # noqa: E731
    edited = [list(f) for f in frames]
    rng = np.random.default_rng(5)
    changes = []
    for f in (2, 3, 4):
        y = np.array(edited[f][3], dtype=np.float64).reshape(-1, 2).copy()
        for k in rng.permutation(N)[:5]:
            y[k] += rng.normal(size=2) * 0.7
            changes.append((f, int(k), float(y[k, 0]), float(y[k, 1])))
        edited[f][3] = y.reshape(-1)
    fresh = bench.make_filter(world, settings, N, None, frames, mk)
    inplace = bench.make_filter(world, settings, N, None, frames, mk)
    from eqvio_amd.capi import OPT_MEASURE_IN_PROPAGATE

    for f in (fresh, inplace): # bit for bit: both filters on routes that evaluate the output blocks in the update's own kernels (see test_output_blocks_from_the_propagation_kernel)
        assert load_eqf_lib().eqf_set_option(f.core_handle(), OPT_MEASURE_IN_PROPAGATE, 0) == 0
    pf_fresh = PreparedFrames(world.cam, *bench.flatten_frames([tuple(f) for f in edited[:5]]))
    pf_edit = PreparedFrames(world.cam, *bench.flatten_frames(frames[:5]))  # built (flat copies cached) with the OLD pixels ...
    for f, k, u, v in changes:
        pf_edit.edit_pixel(f, k, u, v)  # ... then edited through the map only
    for f in range(5):
        assert fresh.run_prepared(pf_fresh, f, 1) == 1 and inplace.run_prepared(pf_edit, f, 1) == 1
        (sa, ia, pa), (sb, ib, pb) = fresh.state_estimate(), inplace.state_estimate()
        assert np.array_equal(sa, sb) and np.array_equal(ia, ib) and np.array_equal(pa, pb), f
        assert np.array_equal(fresh.get_sigma(), inplace.get_sigma()), f
    fresh.close()
    inplace.close()


def test_feature_replaced_in_place_in_an_otherwise_quiet_frame():
    """Round 5: in a frame whose (cached, unvalidated) ids are exactly the ones the last update mapped, VIOFilter::processVisionData decides "nobody lost, nobody new"
    on the cache's word, launches the propagation and validates the measurement against its std::map BESIDE that kernel. A caller who replaced a feature in the public
    map of an already built measurement (erase + insert under another id: the size stays, eqvio_frames_edit_id) is then found out behind the launch, and the lost and
    the new landmark are dealt with there - the reference's own order. Against a filter that is given measurements built with the replaced ids from the start (whose
    frames take the turnover route: removal inside the propagation kernel, held new landmark): identical state and Sigma to rounding (the two orders are the same
    arithmetic; the output blocks come from different kernels, see test_output_blocks_from_the_propagation_kernel) - and identical landmark sets, frame by frame."""
    N = 60
    world, frames = bench.build_workload(seed=9, n_frames=7, N=N)
    settings = bench.eurocish_settings()
    mk = lambda s, sensor, ids, p, t: VIOFilter(s, max_landmarks=N + 8, sensor=sensor, ids=ids, p=p, time=t)  # noqa: E731
    edited = [list(f) for f in frames]
    swaps = []
    next_id = int(max(int(np.max(f[2])) for f in frames)) + 100
    for f, k in ((3, 7), (5, 0), (6, 59)):
        ids = np.array(edited[f][2]).copy()
        y = np.array(edited[f][3], dtype=np.float64).reshape(-1, 2).copy()
        # erase + insert: the new id is the largest, so the pixel moves to the end of the (ascending) flat arrays
        px = y[k].copy()
        ids = np.concatenate([np.delete(ids, k), [next_id]])
        y = np.vstack([np.delete(y, k, axis=0), px])
        edited[f][2], edited[f][3] = ids.astype(np.int32), y.reshape(-1)
        swaps.append((f, k, next_id))
        next_id += 1
    fresh = bench.make_filter(world, settings, N, None, frames, mk)
    inplace = bench.make_filter(world, settings, N, None, frames, mk)
    pf_fresh = PreparedFrames(world.cam, *bench.flatten_frames([tuple(f) for f in edited[:7]]))
    pf_edit = PreparedFrames(world.cam, *bench.flatten_frames(frames[:7]))  # built (flat copies cached) with the OLD ids ...
    for f, k, nid in swaps:  # ... then edited through the map only
        pf_edit.edit_id(f, k, nid)
    for f in range(7):
        assert fresh.run_prepared(pf_fresh, f, 1) == 1 and inplace.run_prepared(pf_edit, f, 1) == 1
        (sa, ia, pa), (sb, ib, pb) = fresh.state_estimate(), inplace.state_estimate()
        assert np.array_equal(ia, ib), f
        assert np.allclose(sa, sb, rtol=0, atol=1e-11) and np.allclose(pa, pb, rtol=0, atol=1e-10), f
        Sa, Sb = fresh.get_sigma(), inplace.get_sigma()
        assert np.linalg.norm(Sa - Sb) <= 1e-10 * np.linalg.norm(Sa), f
    fresh.close()
    inplace.close()


@pytest.mark.parametrize("N", [200, 60])
def test_measurement_and_z_inside_the_lookahead_kernel_change_nothing(N):
    """EQF_OPT_Z_IN_LOOKAHEAD in the speculative frame tail (bench.py's path): 0 = k_build_Z in front of the look-ahead kernel, 2 = the look-ahead kernel
    evaluates the C blocks, builds its rows of Z, eliminates the first tile and computes the outlier statistics itself (forced also at N = 200, where the
    default keeps k_build_Z). Same expressions entry by entry: state, Sigma and the statistics must be identical bit for bit, frame after frame."""
    from eqvio_amd.capi import OPT_Z_IN_LOOKAHEAD

    lib = load_eqf_lib()
    world, frames = bench.build_workload(seed=21, n_frames=7, N=N)
    settings = bench.eurocish_settings()
    mk = lambda s, sensor, ids, p, t: VIOFilter(s, max_landmarks=N, sensor=sensor, ids=ids, p=p, time=t)  # noqa: E731
    from eqvio_amd.capi import OPT_MEASURE_IN_PROPAGATE

    flts = []
    for val in (0, 2):
        f = bench.make_filter(world, settings, N, None, frames, mk)
        assert lib.eqf_set_option(f.core_handle(), OPT_Z_IN_LOOKAHEAD, val) == 0
        # (the third route, output blocks evaluated by the propagation kernel, is compiled in another context and agrees to rounding only: its own test below)
        assert lib.eqf_set_option(f.core_handle(), OPT_MEASURE_IN_PROPAGATE, 0) == 0
        flts.append(f)
    pf = PreparedFrames(world.cam, *bench.flatten_frames(frames[:6]))
    for k in range(6):
        for f in flts:
            assert f.run_prepared(pf, k, 1) == 1
        (sa, ia, pa), (sb, ib, pb) = flts[0].state_estimate(), flts[1].state_estimate()
        assert np.array_equal(sa, sb) and np.array_equal(ia, ib) and np.array_equal(pa, pb), k
        assert np.array_equal(flts[0].get_sigma(), flts[1].get_sigma()), k
    ca, cb = counters(flts[0]), counters(flts[1])
    assert ca["queued"] == cb["queued"] == 6 and ca["cancelled"] == cb["cancelled"] == 0 and cb["la_launches"] == 6 and cb["la_fallbacks"] == 0
    for f in flts:
        f.close()


@pytest.mark.parametrize("N", [200, 60])
def test_output_blocks_from_the_propagation_kernel(N):
    """EQF_OPT_MEASURE_IN_PROPAGATE (round 4, default): the observer blocks of the propagation kernel evaluate the output blocks of the staged measurement, the look-ahead kernel
    builds Z from them (no k_build_Z launch up to 16 panels). Same function and inputs as the update's own evaluation, but compiled in another kernel: the compiler contracts
    the expressions into fused multiply-adds differently there (with -ffp-contract=off the routes agree bit for bit), so state and Sigma agree to rounding, not bitwise.
    Every frame must take the route (counter), and a measurement that was edited after it was staged must fall back and give what a freshly built one gives."""
    from eqvio_amd.capi import OPT_MEASURE_IN_PROPAGATE

    lib = load_eqf_lib()
    world, frames = bench.build_workload(seed=21, n_frames=9, N=N)
    settings = bench.eurocish_settings()
    mk = lambda s, sensor, ids, p, t: VIOFilter(s, max_landmarks=N, sensor=sensor, ids=ids, p=p, time=t)  # noqa: E731
    on, off = bench.make_filter(world, settings, N, None, frames, mk), bench.make_filter(world, settings, N, None, frames, mk)
    assert lib.eqf_set_option(off.core_handle(), OPT_MEASURE_IN_PROPAGATE, 0) == 0
    pf = PreparedFrames(world.cam, *bench.flatten_frames(frames[:8]))
    for k in range(8):
        assert on.run_prepared(pf, k, 1) == 1 and off.run_prepared(pf, k, 1) == 1
        (sa, ia, pa), (sb, ib, pb) = on.state_estimate(), off.state_estimate()
        assert np.array_equal(ia, ib)
        assert np.allclose(sa, sb, rtol=1e-12, atol=1e-13) and np.allclose(pa, pb, rtol=1e-12, atol=1e-13), k
        Sa, Sb = on.get_sigma(), off.get_sigma()
        assert np.max(np.abs(Sa - Sb)) <= 1e-12 * np.max(np.abs(Sb)), k
    used = C.c_long()
    assert lib.eqf_measure_in_propagate_stats(on.core_handle(), C.byref(used), 0) == 0 and used.value == 7  # every frame but the first (no update call had named a camera yet)
    assert lib.eqf_measure_in_propagate_stats(off.core_handle(), C.byref(used), 0) == 0 and used.value == 0
    on.close()
    off.close()


@pytest.mark.parametrize("N,lookahead", [(200, 1), (60, 1), (60, 0), (5, 1)])
def test_lift_and_covariance_update_in_one_launch_change_nothing(N, lookahead):
    """EQF_OPT_LIFT_WITH_SYRK (round 4, default): the landmark lift / result packet / doorbell and Sigma <- Sigma - W W^T (VIO_eqf.cpp:130-131) are one launch instead of two in
    a row - behind the look-ahead kernel, behind the launch chain (whose lift sums Gamma's partial vectors) and at a size without a factorisation kernel of its own.
    Same arithmetic per thread: state, Sigma and estimates identical bit for bit, frame after frame."""
    from eqvio_amd.capi import OPT_LIFT_WITH_SYRK, OPT_LOOKAHEAD

    lib = load_eqf_lib()
    world, frames = bench.build_workload(seed=23, n_frames=7, N=N)
    settings = bench.eurocish_settings()
    mk = lambda s, sensor, ids, p, t: VIOFilter(s, max_landmarks=N, sensor=sensor, ids=ids, p=p, time=t)  # noqa: E731
    flts = []
    for val in (0, 1):
        f = bench.make_filter(world, settings, N, None, frames, mk)
        assert lib.eqf_set_option(f.core_handle(), OPT_LIFT_WITH_SYRK, val) == 0
        assert lib.eqf_set_option(f.core_handle(), OPT_LOOKAHEAD, lookahead) == 0
        flts.append(f)
    pf = PreparedFrames(world.cam, *bench.flatten_frames(frames[:6]))
    for k in range(6):
        for f in flts:
            assert f.run_prepared(pf, k, 1) == 1
        (sa, ia, pa), (sb, ib, pb) = flts[0].state_estimate(), flts[1].state_estimate()
        assert np.array_equal(sa, sb) and np.array_equal(ia, ib) and np.array_equal(pa, pb), k
        assert np.array_equal(flts[0].get_sigma(), flts[1].get_sigma()), k
    for f in flts:
        f.close()


def test_output_blocks_wait_for_the_last_observer_step():
    """More IMU samples between two frames than one propagation launch carries (kMaxSteps = 24; 600 Hz IMU at 20 Hz camera = 30): the landmarks' group elements are final only
    after a further k_observer launch, so the propagation kernel must NOT evaluate the output blocks (they would belong to an intermediate state). Teacher forced against the oracle."""
    from run_configs import parity

    N = 60
    world = SimWorld(seed=5, num_points=N, max_features=N, trajectory="hover", imu_freq=600.0, image_freq=20.0, noise_px=0.5)
    frames = list(world.frames(6))
    assert max(len(f[0]) for f in frames) > 24
    settings = bench.eurocish_settings()
    ids0 = frames[0][2]
    sensor, ids, p = world.true_state(0.0, ids0)
    p = p * (1.0 + 0.05 * np.random.default_rng(7).normal(size=(len(ids), 1)))
    flt = VIOFilter(settings, max_landmarks=N, sensor=sensor, ids=ids, p=p, time=0.0)
    orc = OracleFilter(settings, sensor, ids, p, 0.0)
    prepared = PreparedFrames(world.cam, *bench.flatten_frames(frames))
    for f, (imus, stamp, mid, y) in enumerate(frames):
        assert flt.run_prepared(prepared, f, 1) == 1
        for k_ in range(len(imus)):
            orc.process_imu(imus[k_])
        orc.process_vision(stamp, world.cam, mid, y)
        es, eS = parity(flt, orc)
        assert es <= TOL and eS <= TOL, (f, es, eS)
        teacher_force(flt, orc)
    used = C.c_long()
    assert load_eqf_lib().eqf_measure_in_propagate_stats(flt.core_handle(), C.byref(used), 0) == 0 and used.value == 0
    flt.close()


@pytest.mark.parametrize("N,Mmeas,shuffled", [(272, 272, False), (300, 289, True), (500, 500, False), (512, 497, True)])
def test_z_inside_the_lookahead_kernel_above_16_panels(N, Mmeas, shuffled):
    """Round 6: EQF_OPT_Z_IN_LOOKAHEAD also with 17 .. 32 panels (la_build_rows2: the half-rows of the la_row2 form - split ones included - build their tile ranges of Z
    themselves, no k_build_Z launch). Option value 2 keeps round 5's behaviour (k_build_Z above 16 panels). Through eqf_vision_update (output blocks from the
    measurement kernel: the same expressions entry by entry, Sigma+ and the state must be identical BIT FOR BIT) and through the staged eqf_stats_then_update (output
    blocks from the propagation kernel, statistics from the kernel's own workgroup: rounding level, test_output_blocks_from_the_propagation_kernel says why). Ragged last
    panels, fewer measurements than landmarks, identity and loaded index maps; every launch must have built Z inside (counter) and none may have stalled."""
    from eqvio_amd.capi import COORD_INVDEPTH, OPT_Z_IN_LOOKAHEAD, EqfCore
    from util import default_camera, random_imu, random_spd, reasonable_state, settings_for, synth_measurement

    rng = np.random.default_rng(1000 + N)
    s = settings_for(COORD_INVDEPTH, fastRiccati=1)
    xi0, Xs, ids, q0, Q = reasonable_state(rng, N, shuffle_ids=shuffled)
    S0 = random_spd(rng, 21 + 3 * N)
    cam = default_camera()
    Qd, Pd = s.input_gain_diag12(), s.state_gain_diag8()
    imus = np.stack([random_imu(rng, stamp=0.005 * i) for i in range(4)])
    mean = random_imu(rng)
    sub = None if Mmeas == N else np.sort(rng.permutation(N)[:Mmeas])
    mid, y = synth_measurement(rng, cam, ids, q0, Q, noise_px=0.5, subset=sub)
    outs = {}
    for route in ("update", "staged"):
        for zb in (2, 1):
            core = EqfCore(N, COORD_INVDEPTH)
            core.set_option(OPT_Z_IN_LOOKAHEAD, zb)
            core.set_state(xi0, Xs, ids, q0, Q)
            core.set_sigma(S0)
            stats = None
            if route == "update":
                core.vision_update(cam, mid, y, 4.0, True, True)
            else:
                for rep in range(2):  # the second frame finds the camera of the first: output blocks from the propagation kernel
                    core.stage_measurement(mid, y)
                    core.propagate_fast(mean, 0.02, Qd, Pd, imus, np.full(4, 0.005), True)
                    upd, *stats = core.stats_then_update(cam, mid, y, 1e9, 1e9, 4.0, True, True)
                    assert upd == 1
            la, fb, zbl = C.c_long(), C.c_long(), C.c_long()
            assert core.lib.eqf_lookahead_stats(core.h, C.byref(la), C.byref(fb), 0) == 0
            assert la.value == (1 if route == "update" else 2) and fb.value == 0
            assert core.lib.eqf_z_in_lookahead_stats(core.h, C.byref(zbl), 0) == 0
            assert zbl.value == (1 if zb == 1 else 0), (route, zb, zbl.value)  # (staged: the first frame has no camera yet and takes k_build_Z)
            outs[route, zb] = (core.get_state(), core.get_sigma(), stats)
            core.close()
    (sa, Sa, _), (sb, Sb, _) = outs["update", 2], outs["update", 1]
    assert all(np.array_equal(x, z) for x, z in zip(sa, sb)) and np.array_equal(Sa, Sb)
    (sa, Sa, ta), (sb, Sb, tb) = outs["staged", 2], outs["staged", 1]
    assert all(np.allclose(x, z, rtol=1e-11, atol=1e-13) for x, z in zip(sa, sb)) and np.allclose(Sa, Sb, rtol=1e-10, atol=1e-12 * np.abs(Sa).max())
    for u, v in zip(ta, tb):
        assert np.allclose(u, v, rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("shuffled", [False, True])
def test_index_map_of_the_z_building_prologue(shuffled):
    """The look-ahead kernel's prologue builds Z from Sigma's landmark blocks; since round 4 it does not load the measurement -> landmark index map when the host
    has seen that measurement j belongs to landmark j for every j (the regular frame). Both forms - landmarks in ascending id order (identity) and in shuffled order (the map
    is loaded) - against the k_build_Z route, through eqf_vision_update (output blocks from the measurement kernel) and through the staged eqf_stats_then_update (output blocks
    from the propagation kernel, compared at rounding level: test_output_blocks_from_the_propagation_kernel says why)."""
    from eqvio_amd.capi import COORD_INVDEPTH, OPT_MEASURE_IN_PROPAGATE, OPT_Z_IN_LOOKAHEAD, EqfCore
    from util import default_camera, random_imu, random_spd, reasonable_state, settings_for, synth_measurement

    N = 72
    rng = np.random.default_rng(7)
    s = settings_for(COORD_INVDEPTH, fastRiccati=1)
    xi0, Xs, ids, q0, Q = reasonable_state(rng, N, shuffle_ids=shuffled)
    assert shuffled == bool(np.any(np.diff(ids) < 0))
    S0 = random_spd(rng, 21 + 3 * N)
    cam = default_camera()
    Qd, Pd = s.input_gain_diag12(), s.state_gain_diag8()
    imus = np.stack([random_imu(rng, stamp=0.005 * i) for i in range(4)])
    mean = random_imu(rng)
    mid, y = synth_measurement(rng, cam, ids, q0, Q, noise_px=0.5)
    outs = {}
    for route in ("update", "staged"):
        for zb in (0, 1):
            core = EqfCore(N, COORD_INVDEPTH)
            core.set_option(OPT_Z_IN_LOOKAHEAD, zb)
            core.set_state(xi0, Xs, ids, q0, Q)
            core.set_sigma(S0)
            if route == "update":
                core.vision_update(cam, mid, y, 4.0, True, True)
            else:
                for rep in range(2):  # the second frame finds the camera of the first: output blocks from the propagation kernel
                    core.stage_measurement(mid, y)
                    core.propagate_fast(mean, 0.02, Qd, Pd, imus, np.full(4, 0.005), True)
                    upd, *_ = core.stats_then_update(cam, mid, y, 1e9, 1e9, 4.0, True, True)
                    assert upd == 1
                used = C.c_long()
                assert core.lib.eqf_measure_in_propagate_stats(core.h, C.byref(used), 0) == 0
                assert used.value == (1 if zb else 0)
            outs[route, zb] = (core.get_state(), core.get_sigma())
            core.close()
    (sa, Sa), (sb, Sb) = outs["update", 0], outs["update", 1]
    assert all(np.array_equal(x, z) for x, z in zip(sa, sb)) and np.array_equal(Sa, Sb)
    (sa, Sa), (sb, Sb) = outs["staged", 0], outs["staged", 1]
    assert all(np.allclose(x, z, rtol=1e-11, atol=1e-13) for x, z in zip(sa, sb)) and np.allclose(Sa, Sb, rtol=1e-10, atol=1e-12 * np.abs(Sa).max())
