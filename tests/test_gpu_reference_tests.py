"""The reference's OWN tests, run against the HIP path (missing #1 / next #7).

The reference holds no golden vectors; what it does hold is (i) the differential tests of the EqF matrices (test/test_EqFMatrices.cpp:60-179: A, B, C must be the
derivatives of the compositions of maps they are derived from) and (ii) the statistical tests of the filter (test/test_FilterStatistics.cpp:98-168, the only
reference tests that execute Sigma arithmetic: the mean NEES of 1000 particles must stay near 1 through propagation and update). Here those tests call the
DEVICE for everything the reference's tests call VIO_eqf / the coordinate suites for — eqf_debug_matrices_AB, eqf_debug_matrix_C, eqf_integrate_riccati_discrete,
eqf_integrate_observer, eqf_vision_update, eqf_compute_nees — and evaluate the test's own side (particles, system function, group action, charts, numerical
differentials, resampling) on the host in numpy. No filter-level oracle output enters an assertion: the geometric maps come from the numpy module
oracle/indep/eqvio_ref.py (test infrastructure; group action, charts, lifts, exponential — NOT its A / B / C / Riccati / update code), the expected values are
the reference tests' own: "equals the numerical differential to max(h, 10 h |entry|)" and "mean NEES within 0.1 / 1.0 / 0.1 / 0.5 of 1"."""
import os
import sys

import numpy as np
import pytest

from eqvio_amd.capi import COORD_EUCLIDEAN, COORD_INVDEPTH, COORD_NORMAL, Camera, EqfCore, Settings
from util import default_camera, so3_exp, unit_quat

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "indep"))
from eqvio_ref import Algebra, EqVIORef, F64  # noqa: E402
from eqvio_ref import Camera as RCam  # noqa: E402

pytestmark = pytest.mark.gpu
R = EqVIORef(F64())
CBRT_EPS = float(np.cbrt(np.finfo(np.float64).eps))
CBRT_EPS_F = float(np.cbrt(np.finfo(np.float32).eps))
TEST_REPS = 25  # test/CMakeLists.txt:33
NEAR_ZERO = 1e-12  # test/CMakeLists.txt:34
SUITES = {"euclid": COORD_EUCLIDEAN, "invdepth": COORD_INVDEPTH, "normal": COORD_NORMAL}


# ---- test/testing_utilities.cpp:24-124 on flat arrays (tests/util.py layout) ---------------------------------------------------------
def reasonable_state_element(rng, ids):
    xi0 = np.zeros(23)
    xi0[0:6] = rng.uniform(-1, 1, 6)
    xi0[6:10] = unit_quat(rng)
    xi0[10:13] = rng.uniform(-1, 1, 3)
    xi0[13:16] = rng.uniform(-1, 1, 3)
    xi0[16:20] = unit_quat(rng)
    xi0[20:23] = rng.uniform(-1, 1, 3)
    q0 = rng.uniform(-1, 1, (len(ids), 3)) * 10.0
    q0[:, 2] += 20.0
    return xi0, q0


def reasonable_group_element(rng, ids):
    Xs = np.zeros(23)
    Xs[0:6] = rng.uniform(-1, 1, 6) * 0.1
    A = R.se3_exp(rng.uniform(-1, 1, 6) * 0.1)
    B = R.se3_exp(rng.uniform(-1, 1, 6) * 0.1)
    Xs[6:10], Xs[10:13] = R.R_to_quat(A.R), A.x
    Xs[13:16] = rng.uniform(-1, 1, 3) * 0.1
    Xs[16:20], Xs[20:23] = R.R_to_quat(B.R), B.x
    Q = np.zeros((len(ids), 5))
    for i in range(len(ids)):
        Q[i, :4] = so3_exp(rng.uniform(-1, 1, 3) * 0.02)
        Q[i, 4] = 2.0 * rng.uniform() + 1.0
    return Xs, Q


def random_velocity_element(rng):
    v = np.zeros(13)
    v[1:13] = rng.uniform(-1, 1, 12)
    return v


def algebra_sub(a, b):
    return Algebra(a.u_beta - b.u_beta, a.U_A - b.U_A, a.U_B - b.U_B, a.u_w - b.u_w, a.ids, a.W - b.W)


def assert_matrix_equality(M1, M2, h):
    """assertMatrixEquality (test/testing_utilities.cpp:186-215): entrywise |M1 - M2| <= max(h, 10 h |M1|)."""
    assert M1.shape == M2.shape and np.all(np.isfinite(M1)) and np.all(np.isfinite(M2))
    bad = np.abs(M1 - M2) > np.maximum(h, 10.0 * h * np.abs(M1))
    assert not bad.any(), (np.argwhere(bad)[:5], np.abs(M1 - M2)[bad][:5])


def device_core(chart, xi0, Xs, ids, q0, Q):
    core = EqfCore(len(ids), chart)
    core.set_state(xi0, Xs, np.asarray(ids, np.int32), q0, Q)
    return core


# ---- EqFSuiteTest (test/test_EqFMatrices.cpp:60-179) ---------------------------------------------------------------------------------
@pytest.mark.parametrize("suite", ["euclid", "invdepth", "normal"])
def test_EqFSuiteTest_stateMatrixA(suite):
    """A0 from the DEVICE against the numerical differential of a0 = eps o phi_{X^-1} o phi_xi o exp o LambdaTilde_v o phi_X o eps^-1 (:60-98)."""
    rng = np.random.default_rng(0)
    ids = [0, 1, 2, 3, 4]
    for rep in range(TEST_REPS):
        xi0f, q0 = reasonable_state_element(rng, ids)
        Xsf, Q = reasonable_group_element(rng, ids)
        vel = random_velocity_element(rng)
        A0t, _ = device_core(SUITES[suite], xi0f, Xsf, ids, q0, Q).debug_matrices_AB(vel)
        xi0, X, imu = R.state_from_flat(xi0f, ids, q0), R.group_from_flat(Xsf, ids, Q), R.imu_from_flat(vel)
        xi_hat = R.state_action(X, xi0)
        Xinv = R.group_inv(X)

        def a0(eps):
            xi = R.state_action(X, R.state_chart_inv(suite, eps, xi0))
            lam = algebra_sub(R.lift_velocity(xi, imu), R.lift_velocity(xi_hat, imu))
            xi_e1 = R.state_action(Xinv, R.state_action(R.vio_exp(lam), xi_hat))
            return R.state_chart(suite, xi_e1, xi0)

        n = 21 + 3 * len(ids)
        assert np.linalg.norm(a0(np.zeros(n))) <= NEAR_ZERO
        assert_matrix_equality(A0t, R.numerical_differential(a0, np.zeros(n)), CBRT_EPS)


@pytest.mark.parametrize("suite", ["euclid", "invdepth", "normal"])
def test_EqFSuiteTest_inputMatrixB(suite):
    """Bt from the DEVICE against the numerical differential of b0 (velocity error -> state error, :100-137)."""
    rng = np.random.default_rng(0)
    ids = [0, 1, 2, 3, 4]
    for rep in range(TEST_REPS):
        xi0f, q0 = reasonable_state_element(rng, ids)
        Xsf, Q = reasonable_group_element(rng, ids)
        vel = random_velocity_element(rng)
        _, Bt = device_core(SUITES[suite], xi0f, Xsf, ids, q0, Q).debug_matrices_AB(vel)
        xi0, X, imu = R.state_from_flat(xi0f, ids, q0), R.group_from_flat(Xsf, ids, Q), R.imu_from_flat(vel)
        xi_hat = R.state_action(X, xi0)
        Xinv = R.group_inv(X)

        def b0(err):
            noisy = dict(imu, gyr=imu["gyr"] + err[0:3], acc=imu["acc"] + err[3:6], gyrBiasVel=imu["gyrBiasVel"] + err[6:9], accBiasVel=imu["accBiasVel"] + err[9:12])
            lam = algebra_sub(R.lift_velocity(xi_hat, noisy), R.lift_velocity(xi_hat, imu))
            return R.state_chart(suite, R.state_action(Xinv, R.state_action(R.vio_exp(lam), xi_hat)), xi0)

        assert np.linalg.norm(b0(np.zeros(12))) <= NEAR_ZERO
        assert_matrix_equality(Bt, R.numerical_differential(b0, np.zeros(12)), CBRT_EPS)


@pytest.mark.parametrize("suite", ["euclid", "invdepth", "normal"])
def test_EqFSuiteTest_outputMatrixC(suite):
    """Ct from the DEVICE (equivariant and plain form coincide at y = yHat, :153-154) against the numerical differential of ct = h o phi_XHat o eps^-1
    with the float step (:140-179). ids {5,0,1,2,3,4}: state order differs from the measurement's ascending-id row order."""
    rng = np.random.default_rng(0)
    ids = [5, 0, 1, 2, 3, 4]
    cam = default_camera()
    rcam = RCam(0, cam.fx, cam.fy, cam.cx, cam.cy)
    order = sorted(ids)
    for rep in range(TEST_REPS):
        xi0f, q0 = reasonable_state_element(rng, ids)
        Xsf, Q = reasonable_group_element(rng, ids)
        xi0, X = R.state_from_flat(xi0f, ids, q0), R.group_from_flat(Xsf, ids, Q)
        y_hat = R.measure(R.state_action(X, xi0), rcam)
        y_flat = np.concatenate([y_hat[i] for i in order])
        core = device_core(SUITES[suite], xi0f, Xsf, ids, q0, Q)
        Ct, ytil = core.debug_matrix_C(cam, np.array(order, np.int32), y_flat, True)
        Ct2, _ = core.debug_matrix_C(cam, np.array(order, np.int32), y_flat, False)
        assert_matrix_equality(Ct, Ct2, CBRT_EPS)
        assert np.abs(ytil).max() <= 1e-9  # the residual of yHat itself

        def ct(eps):
            y = R.measure(R.state_action(X, R.state_chart_inv(suite, eps, xi0)), rcam)
            return np.concatenate([y[i] - y_hat[i] for i in order])

        n = 21 + 3 * len(ids)
        # NEAR_ZERO on PIXEL values of several hundred: the host's own round trip eps^-1 -> phi -> h costs a few ulp of that (1.0e-12 observed for the
        # inverse-depth chart with numpy on the GPU box); the reference's check is the same number in Eigen. Scaled by the pixel magnitude here.
        assert np.linalg.norm(ct(np.zeros(n))) <= NEAR_ZERO * max(1.0, np.abs(y_flat).max() / 100.0)
        assert_matrix_equality(Ct, R.numerical_differential(ct, np.zeros(n), CBRT_EPS_F), CBRT_EPS_F)


# ---- FilterStatisticsTest (test/test_FilterStatistics.cpp) ---------------------------------------------------------------------------
NUM_PARTICLES = 1000  # :26


class Fixture:
    """The constructor of FilterStatisticsTest (:30-53): InvDepth, two landmarks, the variances of :32-41, filter = (xi0, Identity, Sigma0) ON THE DEVICE,
    1000 particles xi = phi(exp(liftInnovation(eps)), xi0), eps ~ N(0, Sigma0), on the host."""

    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)
        self.ids = [0, 1]
        s = Settings.defaults()
        s.coordinateChoice = COORD_INVDEPTH
        s.initialPointVariance = s.initialPointDepthVariance = 0.01**2
        s.initialBiasOmegaVariance = s.initialBiasAccelVariance = 0.01**2
        s.initialVelocityVariance = 0.1**2
        s.initialPositionVariance = 0.001**2
        self.settings = s
        xi0f, q0 = reasonable_state_element(self.rng, self.ids)
        self.xi0 = R.state_from_flat(xi0f, self.ids, q0)
        ident = np.zeros(23)
        ident[6] = ident[16] = 1.0
        Q = np.tile(np.array([1.0, 0, 0, 0, 1.0]), (2, 1))
        self.core = device_core(COORD_INVDEPTH, xi0f, ident, self.ids, q0, Q)
        self.sigma0_diag = s.initial_cov_diag(2)
        self.core.set_sigma(np.diag(self.sigma0_diag))
        self.particles = []
        for _ in range(NUM_PARTICLES):
            eps = np.sqrt(self.sigma0_diag) * self.rng.normal(size=27)  # sampleGaussianDistribution (Geometry.cpp:53-69) for a diagonal covariance
            self.particles.append(R.state_action(R.vio_exp(R.lift_innovation("invdepth", eps, self.xi0)), self.xi0))
        self.cam = Camera.pinhole(458.654, 457.296, 367.215, 248.375, 752, 480)
        self.rcam = RCam(0, 458.654, 457.296, 367.215, 248.375)

    def mean_nees(self):
        """computeMeanNEES (:74-82): VIO_eqf::computeNEES per particle — every one of them on the device."""
        ids = np.array(self.ids, np.int32)
        return float(np.mean([self.core.compute_nees(R.sensor_to_flat(p.sensor), ids, np.asarray(p.p, float)) for p in self.particles]))


def test_FilterStatisticsTest_initialDistribution():  # :98
    assert abs(Fixture(1).mean_nees() - 1.0) <= 0.1


def _propagate(f, true_vel, dt, tol):
    imu = R.imu_from_flat(true_vel)
    zq, zp = np.zeros(12), np.zeros(8)
    for rep in range(5):
        f.particles = [R.integrate_system(p, imu, dt) for p in f.particles]
        f.core.integrate_riccati_discrete(true_vel, dt, zq, zp)  # 0 * inputGain, 0 * stateGain (:110-111, :133-134)
        f.core.integrate_observer(true_vel[None, :], np.array([dt]), True)
        assert abs(f.mean_nees() - 1.0) <= tol, rep


def test_FilterStatisticsTest_trueInputDistribution():  # :100-117
    _propagate(Fixture(2), np.zeros(13), 0.2, 1.0)


def test_FilterStatisticsTest_inputDistribution():  # :119-140
    f = Fixture(3)
    vel = np.zeros(13)
    vel[1:7] = f.rng.uniform(-1, 1, 6)  # IMUVelocity(Matrix<double,6,1>::Random())
    _propagate(f, vel, 0.05, 0.1)


def test_FilterStatisticsTest_outputDistribution():  # :142-168
    f = Fixture(4)
    var = f.settings.measurementNoise**2  # constructOutputGainMatrix (VIOFilterSettings.h:203-206)
    y0 = R.measure(f.xi0, f.rcam)
    meas = {i: y0[i] + np.sqrt(var) * f.rng.normal(size=2) for i in f.ids}
    w = np.zeros(NUM_PARTICLES)
    for k, p in enumerate(f.particles):
        y = R.measure(p, f.rcam)
        err = np.concatenate([meas[i] - y[i] for i in f.ids])
        w[k] = np.exp(-0.5 * err @ err / var)
    w /= w.sum()
    # weightedResample (test/testing_utilities.h:55-74)
    res, j, total = [], 0, w[0]
    for k in range(NUM_PARTICLES):
        thr = (f.rng.uniform() + k) / NUM_PARTICLES
        while total < thr and j + 1 < NUM_PARTICLES:
            j += 1
            total += w[j]
        res.append(f.particles[j])
    f.particles = res
    f.core.vision_update(f.cam, np.array(f.ids, np.int32), np.concatenate([meas[i] for i in f.ids]), var, True, False)  # performVisionUpdate's defaults (VIO_eqf.h:107-109)
    assert abs(f.mean_nees() - 1.0) <= 0.5
