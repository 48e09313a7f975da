"""Edge cases of the C-ABI the reference's control flow relies on (VIO_eqf.cpp:105-135, 172-245) and the error contract of
include/eqf_hip.h: empty and ragged measurements, empty landmark set, capacity, loud failures (never a silent fallback)."""
import numpy as np
import pytest

from eqvio_amd.capi import OPT_CHECK_FINITE, EqfCore, EqfError
from test_gpu_parity import check_sigma, check_state, make_pair
from oracle_binding import OracleFilter
from util import CHARTS, default_camera, random_imu, random_spd, reasonable_state, rel_fro, settings_for, synth_measurement

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("chart", list(CHARTS))
def test_empty_measurement_is_a_no_op(chart):
    """performVisionUpdate returns immediately on an empty measurement (VIO_eqf.cpp:108-109)."""
    rng, settings, orc, core, (xi0, Xs, ids, q0, Q, S) = make_pair(CHARTS[chart], 7, seed=1)
    cam = default_camera()
    core.vision_update(cam, np.zeros(0, np.int32), np.zeros(0), settings.measurementNoise**2, True, False)
    assert np.array_equal(core.get_sigma(), S)
    check_state(core, orc)


@pytest.mark.parametrize("chart", list(CHARTS))
def test_no_landmarks_propagates_the_sensor_block_only(chart):
    """N = 0: n = 21; propagation and observer steps work, outlier statistics of an empty state are empty."""
    rng, settings, orc, core, _ = make_pair(CHARTS[chart], 0, seed=2, cap=8)
    imu = random_imu(rng, bias_vel=True)
    for f, g in ((orc.integrate_riccati_fast, lambda: core.integrate_riccati_fast(imu, 0.05, settings.input_gain_diag12(), settings.state_gain_diag8())),
                 (orc.integrate_riccati_accurate, lambda: core.integrate_riccati_accurate(imu, 0.05, settings.input_gain_diag12(), settings.state_gain_diag8()))):
        f(imu, 0.05)
        g()
        check_sigma(core, orc, 1e-11)
    orc.integrate_observer(imu, 0.05, True)
    core.integrate_observer(imu[None, :], np.array([0.05]), True)
    check_state(core, orc)
    a, p, d = core.outlier_stats(default_camera(), np.zeros(0, np.int32), np.zeros(0))
    assert len(a) == len(p) == len(d) == 0
    assert core.get_sigma().shape == (21, 21)


def test_single_landmark_single_measurement():
    """The smallest update: N = M = 1 (m = 2: one 2x2 pivot block, identity-padded tile)."""
    rng, settings, orc, core, (xi0, Xs, ids, q0, Q, S) = make_pair(CHARTS["invdepth"], 1, seed=3, cap=4, sigma="diag")
    cam = default_camera()
    mid, y = synth_measurement(rng, cam, ids, q0, Q, noise_px=1.0)
    orc.vision_update(cam, mid, y)
    core.vision_update(cam, mid, y, settings.measurementNoise**2, True, False)
    check_sigma(core, orc)
    check_state(core, orc)


def test_capacity_grows_on_demand():
    """max_landmarks of eqf_create is an initial capacity, not a limit (the reference has none: VIO_eqf.cpp:225-245 resizes Sigma). A context
    created for 16 landmarks takes 14 + 2 (exactly full) + 30 more, keeps state, Sigma, options and counters across the growth, and the
    propagation + update that follow agree with the oracle."""
    from eqvio_amd.capi import OPT_LOOKAHEAD, OPT_TIMING
    from oracle_binding import OracleFilter

    N, cap = 14, 16
    chart = CHARTS["invdepth"]
    rng = np.random.default_rng(4)
    settings = settings_for(chart, useDiscreteInnovationLift=0)
    xi0, Xs, ids, q0, Q = reasonable_state(rng, N)
    S = np.diag(settings.initial_cov_diag(N)) + 1e-3 * random_spd(rng, 21 + 3 * N)
    orc = OracleFilter(settings)
    orc.set_eqf(xi0, Xs, ids, q0, Q, S)
    core = EqfCore(cap, chart)
    core.set_option(OPT_LOOKAHEAD, 0)
    core.set_option(OPT_TIMING, 1)
    core.set_state(xi0, Xs, ids, q0, Q)
    core.set_sigma(S)
    for new_ids, var in ((np.array([100, 101], np.int32), 1.0), (np.arange(200, 230, dtype=np.int32), 0.4)):
        p = rng.uniform(-1, 1, (len(new_ids), 3)) + [0, 0, 5]
        orc.add_landmarks(new_ids, p, var)
        core.add_landmarks(new_ids, p, var)
    assert core.N == 46
    assert np.array_equal(core.get_sigma(), orc.get_sigma())  # the move to the larger buffers copies bits
    check_state(core, orc, 1e-15)
    imu = random_imu(rng)
    orc.integrate_riccati_fast(imu, 0.01)
    core.integrate_riccati_fast(imu, 0.01, settings.input_gain_diag12(), settings.state_gain_diag8())
    orc.integrate_observer(imu, 0.01, True)
    core.integrate_observer(imu[None, :], np.array([0.01]), True)
    _, _, ids_all, q0_all, Q_all = orc.get_eqf()
    cam = default_camera()
    mid, y = synth_measurement(rng, cam, ids_all, q0_all, Q_all, noise_px=1.0)
    orc.vision_update(cam, mid, y)
    core.vision_update(cam, mid, y, settings.measurementNoise**2, True, False)
    check_sigma(core, orc)
    check_state(core, orc)
    # an option set before the growth is still in force: with the look-ahead kernel off the update ran as the launch chain
    names = {name for name, _ in core.kernel_times()}
    assert "k_chol_step" in names and "k_chol_lookahead" not in names
    # set_state with more landmarks than the current capacity grows as well
    xi0b, Xsb, idsb, q0b, Qb = reasonable_state(rng, 20)
    small = EqfCore(16, CHARTS["euclid"])
    small.set_state(xi0b, Xsb, idsb, q0b, Qb)
    assert small.N == 20 and np.array_equal(small.get_state()[2], idsb)


def test_indefinite_innovation_covariance_is_reported():
    """The reference's LU inverse never fails; the factorisation here reports a non-positive pivot (EQF_E_NOT_SPD)
    instead of producing garbage - and reports it with the filter untouched: Sigma, X (sensor part and landmarks) are what they were
    before the call, for the one-panel chain (N = 5), the look-ahead kernel (N = 60) and the launch chain (N = 60, look-ahead off)."""
    from eqvio_amd.capi import OPT_LOOKAHEAD

    for N, la in ((5, 1), (60, 1), (60, 0)):
        rng, settings, orc, core, (xi0, Xs, ids, q0, Q, S) = make_pair(CHARTS["euclid"], N, seed=5)
        core.set_option(OPT_LOOKAHEAD, la)
        core.set_sigma(-S)  # S_innov = C (-Sigma) C^T + R is indefinite for this Sigma
        before = core.get_state()
        cam = default_camera()
        mid, y = synth_measurement(rng, cam, ids, q0, Q, noise_px=1.0)
        with pytest.raises(EqfError) as e:
            core.vision_update(cam, mid, y, 1e-6, True, False)
            core.synchronize()
        assert e.value.code == -2  # EQF_E_NOT_SPD
        assert np.array_equal(core.get_sigma(), -S)
        for a, b in zip(core.get_state(), before):
            assert np.array_equal(a, b)
        # and the context is usable afterwards: the same update on a proper Sigma matches the oracle
        core.set_sigma(S)
        orc.vision_update(cam, mid, y)
        core.vision_update(cam, mid, y, settings.measurementNoise**2, True, False)
        check_sigma(core, orc)


def test_non_finite_input_is_caught_when_checking_is_on():
    """assert(!Sigma.hasNaN()) of the reference (VIO_eqf.cpp:70, 132) = EQF_OPT_CHECK_FINITE."""
    rng, settings, orc, core, (xi0, Xs, ids, q0, Q, S) = make_pair(CHARTS["euclid"], 4, seed=6)
    core.set_option(OPT_CHECK_FINITE, 1)
    S2 = S.copy()
    S2[3, 3] = np.nan
    core.set_sigma(S2)
    with pytest.raises(EqfError) as e:
        core.integrate_riccati_fast(random_imu(rng), 0.05, settings.input_gain_diag12(), settings.state_gain_diag8())
    assert e.value.code == -1  # EQF_E_NONFINITE


def test_non_finite_sigma_is_caught_after_an_update_when_checking_is_on():
    """A NaN that the update itself never reads (variance of an unmeasured landmark) survives Sigma -= W W^T: only the
    finite check (VIO_eqf.cpp:132) can report it."""
    rng, settings, orc, core, (xi0, Xs, ids, q0, Q, S) = make_pair(CHARTS["invdepth"], 5, seed=8)
    cam = default_camera()
    order = np.argsort(ids)
    unmeasured = order[-1]
    mid, y = synth_measurement(rng, cam, ids, q0, Q, noise_px=1.0, subset=order[:-1])
    S2 = S.copy()
    l = 21 + 3 * unmeasured
    S2[l, l] = np.nan
    core.set_sigma(S2)
    core.vision_update(cam, mid, y, 1.0, True, False)  # unchecked: goes through (the reference would only assert in debug builds)
    core.set_sigma(S2)
    core.set_option(OPT_CHECK_FINITE, 1)
    with pytest.raises(EqfError) as e:
        core.vision_update(cam, mid, y, 1.0, True, False)
    assert e.value.code == -1  # EQF_E_NONFINITE


def test_bad_arguments():
    rng, settings, orc, core, (xi0, Xs, ids, q0, Q, S) = make_pair(CHARTS["euclid"], 4, seed=7)
    with pytest.raises(EqfError):
        core.set_sigma(np.eye(5))  # wrong dimension
    with pytest.raises(EqfError):
        core.integrate_riccati_accurate(random_imu(rng), 0.0, settings.input_gain_diag12(), settings.state_gain_diag8())  # dt must be > 0
    with pytest.raises(EqfError):
        core.remove_landmarks(np.array([9], np.int32))  # index out of range
    cam = default_camera()
    mid, y = synth_measurement(rng, cam, ids, q0, Q, noise_px=1.0)
    with pytest.raises(EqfError):
        core.vision_update(cam, np.concatenate([mid, [999]]).astype(np.int32), np.concatenate([y, [1.0, 2.0]]), 1.0, True, False)  # id not in the state
    check_sigma(core, orc)  # nothing of the above changed the state


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_random_bookkeeping_sequences_match_the_oracle(seed):
    """Landmark bookkeeping is recorded on the host and applied lazily in one kernel (k_reshape) when the device state is next needed.
    Random interleavings of remove / add / remove-what-was-just-added / reads / propagation / updates / capacity growth against the oracle,
    which executes every call at once: Sigma and the landmark arrays must be bit-identical after pure bookkeeping and agree to 1e-9 once
    arithmetic ran in between."""
    rng = np.random.default_rng(1000 + seed)
    chart = CHARTS["invdepth"]
    settings = settings_for(chart, useDiscreteInnovationLift=0)
    N0 = 12
    xi0, Xs, ids, q0, Q = reasonable_state(rng, N0)
    S = np.diag(settings.initial_cov_diag(N0)) + 1e-3 * random_spd(rng, 21 + 3 * N0)
    orc = OracleFilter(settings)
    orc.set_eqf(xi0, Xs, ids, q0, Q, S)
    core = EqfCore(16, chart)  # small on purpose: the sequence outgrows it
    core.set_state(xi0, Xs, ids, q0, Q)
    core.set_sigma(S)
    cam = default_camera()
    next_id = 1000
    exact = True  # no arithmetic since the last exact comparison
    for step in range(120):
        op = rng.choice(["remove", "add", "add_remove", "read", "propagate", "update", "estimate"], p=[0.2, 0.2, 0.1, 0.15, 0.1, 0.15, 0.1])
        n_now = core.N
        if op == "remove" and n_now > 3:
            k = int(rng.integers(1, min(4, n_now - 2)))
            idx = np.sort(rng.choice(n_now, k, replace=False))
            for i in idx[::-1]:
                orc.remove_landmark_by_index(int(i))
            core.remove_landmarks(idx)
        elif op == "add" and n_now < 60:
            k = int(rng.integers(1, 6))
            new_ids = np.arange(next_id, next_id + k, dtype=np.int32)
            next_id += k
            p = rng.uniform(-1, 1, (k, 3)) + [0, 0, 5]
            var = float(rng.uniform(0.1, 2.0))
            orc.add_landmarks(new_ids, p, var)
            core.add_landmarks(new_ids, p, var)
        elif op == "add_remove" and 3 < n_now < 60:
            new_ids = np.arange(next_id, next_id + 3, dtype=np.int32)
            next_id += 3
            p = rng.uniform(-1, 1, (3, 3)) + [0, 0, 5]
            orc.add_landmarks(new_ids, p, 0.5)
            core.add_landmarks(new_ids, p, 0.5)
            for i in (n_now + 1, 0):  # one of the landmarks that exist only in the pending record, and an old one
                orc.remove_landmark_by_index(i)
            core.remove_landmarks(np.array([0, n_now + 1]))
        elif op == "read":
            Sg, So = core.get_sigma(), orc.get_sigma()
            assert Sg.shape == So.shape
            if exact:
                assert np.array_equal(Sg, So)
                check_state(core, orc, 1e-15)
            else:
                assert rel_fro(Sg, So) <= 1e-9
                check_state(core, orc)
        elif op == "propagate":
            imu = random_imu(rng)
            orc.integrate_riccati_fast(imu, 0.01)
            core.integrate_riccati_fast(imu, 0.01, settings.input_gain_diag12(), settings.state_gain_diag8())
            orc.integrate_observer(imu, 0.01, True)
            core.integrate_observer(imu[None, :], np.array([0.01]), True)
            exact = False
        elif op == "update" and n_now > 0:
            _, _, ids_all, q0_all, Q_all = orc.get_eqf()
            sub = np.sort(rng.choice(n_now, max(1, n_now // 2), replace=False))
            mid, y = synth_measurement(rng, cam, ids_all, q0_all, Q_all, noise_px=1.0, subset=sub)
            orc.vision_update(cam, mid, y)
            core.vision_update(cam, mid, y, settings.measurementNoise**2, True, False)
            exact = False
        elif op == "estimate":
            s_g, ids_g, p_g = core.state_estimate()
            s_o, ids_o, p_o = orc.state_estimate()
            assert np.array_equal(ids_g, ids_o)
            assert np.max(np.linalg.norm(p_g - p_o, axis=1) / np.maximum(1.0, np.linalg.norm(p_o, axis=1))) <= (0.0 if exact else 1e-9)
        assert core.N == len(orc.get_eqf()[2])
    assert rel_fro(core.get_sigma(), orc.get_sigma()) <= 1e-9
    check_state(core, orc)


def test_options_and_counters_survive_capacity_growth():
    """growing the capacity rebuilds the context (eqf_hip.hip: grow_capacity) and must carry EVERY option of eqf_set_option and every counter
    over. Walks the option enum through eqf_get_option (so that a new option cannot be forgotten without this test failing: every id declared in
    include/eqf_hip.h that the library knows is set to a non-default value before the growth and read back after it), and checks that the look-ahead
    counters of eqf_lookahead_stats do not restart."""
    import ctypes as C
    import os
    import re

    from eqvio_amd.capi import OPT_LA_TIMEOUT_US, OPT_LOOKAHEAD, OPT_SIGMA_FP32, OPT_TRACE, OPT_Z_IN_LOOKAHEAD

    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "eqf_hip.h")).read()
    enum = hdr[hdr.index("/* options for eqf_set_option */"):]
    enum = enum[: enum.index("};")]
    opt_ids = sorted({int(v) for v in re.findall(r"^\s+EQF_OPT_[A-Z0-9_]+\s*=\s*(\d+)", enum, re.M)} | {100})
    assert len(opt_ids) >= 12, opt_ids
    rng = np.random.default_rng(9)
    N = 40
    chart = CHARTS["invdepth"]
    settings = settings_for(chart, useDiscreteInnovationLift=0)
    xi0, Xs, ids, q0, Q = reasonable_state(rng, N)
    core = EqfCore(N, chart)
    core.set_state(xi0, Xs, ids, q0, Q)
    core.set_sigma(np.diag(settings.initial_cov_diag(N)))
    cam = default_camera()
    mid, y = synth_measurement(rng, cam, ids, q0, Q, noise_px=1.0)
    core.vision_update(cam, mid, y, settings.measurementNoise**2, True, False)  # 3 panels: one look-ahead launch on the books
    a, b = C.c_long(), C.c_long()
    assert core.lib.eqf_lookahead_stats(core.h, C.byref(a), C.byref(b), 0) == 0 and (a.value, b.value) == (1, 0)
    # a non-default value for every option (values that are legal together: the dense Riccati excludes the float store, id 3 = 1 is its fp64 model)
    wanted = {}
    for oid in opt_ids:
        cur = core.get_option(oid)
        new = {OPT_LA_TIMEOUT_US: 12345, OPT_SIGMA_FP32: 1, OPT_TRACE: 1}.get(oid, 0 if cur else 1)
        core.set_option(oid, new)
        wanted[oid] = core.get_option(oid)
        assert wanted[oid] == new, oid
    p = rng.uniform(-1, 1, (70, 3)) + [0, 0, 5]
    core.add_landmarks(np.arange(1000, 1070, dtype=np.int32), p, 0.5)  # 110 landmarks > capacity 40: the context is rebuilt
    assert core.N == 110
    for oid in opt_ids:
        assert core.get_option(oid) == wanted[oid], (oid, core.get_option(oid), wanted[oid])
    assert core.lib.eqf_lookahead_stats(core.h, C.byref(a), C.byref(b), 0) == 0 and (a.value, b.value) == (1, 0)
    assert OPT_LOOKAHEAD in opt_ids and OPT_Z_IN_LOOKAHEAD in opt_ids


def test_lookahead_selftest_runs_at_creation():
    """eqf_create factorises a fixed 96-column problem on the launch chain and on the persistent look-ahead kernel and compares W bit for bit
    (eqf_hip.hip: lookahead_selftest); a context whose capacity never reaches three panels skips it."""
    assert EqfCore(200, CHARTS["invdepth"]).lookahead_selftest() == 1
    assert EqfCore(48, CHARTS["euclid"]).lookahead_selftest() == 1
    assert EqfCore(16, CHARTS["euclid"]).lookahead_selftest() == 0
