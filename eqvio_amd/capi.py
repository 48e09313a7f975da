"""ctypes binding of the C-ABI in include/eqf_hip.h (libeqf_hip.so) and include/eqvio_filter.h
(libeqvio_filter.so). Plumbing only: numpy arrays in, numpy arrays out, errors raised loudly.

There is no CPU fallback anywhere in this package: if the HIP library is missing or no gfx950 device is
present, construction raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.environ.get("EQVIO_AMD_LIB_DIR") or os.path.join(_HERE, "lib")  # the override: same-box A/B of two builds (scripts/ab_builds.sh)

COORD_EUCLIDEAN, COORD_INVDEPTH, COORD_NORMAL = 0, 1, 2
OPT_RICCATI_DENSE, OPT_CHECK_FINITE, OPT_SIGMA_FP32, OPT_DOORBELL, OPT_SPECULATIVE, OPT_EARLY_LIFT, OPT_TRACE, OPT_FUSED_ASSEMBLY, OPT_LOOKAHEAD, OPT_LA_TIMEOUT_US, OPT_Z_IN_LOOKAHEAD, OPT_LA_SPLIT_ROWS, OPT_MEASURE_IN_PROPAGATE, OPT_LIFT_WITH_SYRK, OPT_LA_HOME, OPT_TILES_PER_WORKGROUP, OPT_GATHER_IN_PROPAGATE, OPT_HOLD_NEW_LANDMARKS, OPT_SELECT_ONE_WORKGROUP, OPT_LIVE_COLUMNS_FIRST, OPT_EARLY_DOORBELL, OPT_TIMING = 1, 2, 3, 6, 7, 8, 9, 11, 12, 15, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 100

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)


class Camera(C.Structure):
    """eqvio_camera (include/eqvio_types.h)."""

    _fields_ = [
        ("model", C.c_int),
        ("width", C.c_int),
        ("height", C.c_int),
        ("fx", C.c_double),
        ("fy", C.c_double),
        ("cx", C.c_double),
        ("cy", C.c_double),
        ("dist", C.c_double * 5),
    ]

    @staticmethod
    def pinhole(fx, fy, cx, cy, width, height):
        cam = Camera()
        cam.model, cam.width, cam.height = 0, int(width), int(height)
        cam.fx, cam.fy, cam.cx, cam.cy = fx, fy, cx, cy
        return cam

    @staticmethod
    def radtan(fx, fy, cx, cy, width, height, k1, k2, p1, p2, k3=0.0):
        """GIFT::StandardCamera (ASLDatasetReader.cpp:90-94): OpenCV order k1 k2 p1 p2 k3."""
        cam = Camera.pinhole(fx, fy, cx, cy, width, height)
        cam.model = 1
        cam.dist[:] = [k1, k2, p1, p2, k3]
        return cam

    @staticmethod
    def equidistant(fx, fy, cx, cy, width, height, k1, k2, k3, k4):
        """GIFT::EquidistantCamera (UZHFPVDatasetReader.cpp:99-102): Kannala-Brandt k1..k4."""
        cam = Camera.pinhole(fx, fy, cx, cy, width, height)
        cam.model = 2
        cam.dist[:] = [k1, k2, k3, k4, 0.0]
        return cam

    # the product's camera functions on the host (include/eqvio_sim.h)
    def project(self, p):
        out = np.zeros(2)
        _load_sim_protos().eqvio_camera_project(C.byref(self), _dp(_f64(p)), _dp(out))
        return out

    def undistort(self, y):
        out = np.zeros(3)
        _load_sim_protos().eqvio_camera_undistort(C.byref(self), _dp(_f64(y)), _dp(out))
        return out

    def jacobian(self, p):
        out = np.zeros(6)
        _load_sim_protos().eqvio_camera_jacobian(C.byref(self), _dp(_f64(p)), _dp(out))
        return out.reshape(2, 3)


_SETTINGS_DOUBLES = [
    "biasOmegaProcessVariance", "biasAccelProcessVariance", "attitudeProcessVariance", "positionProcessVariance",
    "velocityProcessVariance", "cameraAttitudeProcessVariance", "cameraPositionProcessVariance", "pointProcessVariance",
    "velGyrNoise", "velAccNoise", "velGyrBiasWalk", "velAccBiasWalk",
    "measurementNoise", "outlierThresholdAbs", "outlierThresholdProb", "featureRetention",
    "initialAttitudeVariance", "initialPositionVariance", "initialVelocityVariance", "initialCameraAttitudeVariance",
    "initialCameraPositionVariance", "initialPointVariance", "initialPointDepthVariance", "initialBiasOmegaVariance",
    "initialBiasAccelVariance", "initialSceneDepth",
]
_SETTINGS_INTS = [
    "useDiscreteInnovationLift", "useDiscreteVelocityLift", "useDiscreteStateMatrix", "fastRiccati", "useMedianDepth",
    "useFeaturePredictions", "useEquivariantOutput", "removeLostLandmarks", "coordinateChoice",
]


class Settings(C.Structure):
    """eqvio_settings (include/eqvio_types.h) == VIOFilter::Settings (VIOFilterSettings.h:58-124)."""

    _fields_ = [(n, C.c_double) for n in _SETTINGS_DOUBLES] + [(n, C.c_int) for n in _SETTINGS_INTS] + [("cameraOffset", C.c_double * 7)]

    @staticmethod
    def defaults():
        """Defaults of VIOFilterSettings.h:59-99."""
        s = Settings()
        vals = dict(
            biasOmegaProcessVariance=0.001, biasAccelProcessVariance=0.001, attitudeProcessVariance=0.001, positionProcessVariance=0.001,
            velocityProcessVariance=0.001, cameraAttitudeProcessVariance=0.001, cameraPositionProcessVariance=0.001, pointProcessVariance=0.001,
            velGyrNoise=1e-4, velAccNoise=1e-3, velGyrBiasWalk=1e-5, velAccBiasWalk=1e-3,
            measurementNoise=2.0, outlierThresholdAbs=1e8, outlierThresholdProb=1e8, featureRetention=0.3,
            initialAttitudeVariance=1e-4, initialPositionVariance=1e-4, initialVelocityVariance=1e-2, initialCameraAttitudeVariance=1e-5,
            initialCameraPositionVariance=1e-4, initialPointVariance=1.0, initialPointDepthVariance=-1.0, initialBiasOmegaVariance=0.1,
            initialBiasAccelVariance=0.1, initialSceneDepth=1.0,
            useDiscreteInnovationLift=1, useDiscreteVelocityLift=1, useDiscreteStateMatrix=0, fastRiccati=0, useMedianDepth=1,
            useFeaturePredictions=0, useEquivariantOutput=1, removeLostLandmarks=1, coordinateChoice=COORD_EUCLIDEAN,
        )
        for k, v in vals.items():
            setattr(s, k, v)
        s.cameraOffset[:] = [1, 0, 0, 0, 0, 0, 0]
        return s

    def state_gain_diag8(self):
        """constructStateGainMatrix (VIOFilterSettings.h:176-190) as 7 sensor classes + point."""
        return np.array([self.biasOmegaProcessVariance, self.biasAccelProcessVariance, self.attitudeProcessVariance, self.positionProcessVariance,
                         self.velocityProcessVariance, self.cameraAttitudeProcessVariance, self.cameraPositionProcessVariance, self.pointProcessVariance])

    def input_gain_diag12(self):
        """constructInputGainMatrix (VIOFilterSettings.h:192-201)."""
        v = [self.velGyrNoise**2, self.velAccNoise**2, self.velGyrBiasWalk**2, self.velAccBiasWalk**2]
        return np.repeat(np.array(v), 3)

    def initial_cov_diag(self, N):
        """constructInitialStateCovariance (VIOFilterSettings.h:208-229)."""
        v = [self.initialBiasOmegaVariance, self.initialBiasAccelVariance, self.initialAttitudeVariance, self.initialPositionVariance,
             self.initialVelocityVariance, self.initialCameraAttitudeVariance, self.initialCameraPositionVariance]
        d = np.concatenate([np.repeat(np.array(v), 3), np.full(3 * N, self.initialPointVariance)])
        if self.initialPointDepthVariance > 0:
            d[21 + 2::3] = self.initialPointDepthVariance
        return d


def _dp(a):
    return a.ctypes.data_as(c_double_p)


def _ip(a):
    return a.ctypes.data_as(c_int_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class EqfError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"eqf_hip error {code}: {msg}")
        self.code = code


_lib = None


def load_eqf_lib():
    """Load libeqf_hip.so and declare every prototype of include/eqf_hip.h. Raises if the library is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.path.join(LIB_DIR, "libeqf_hip.so")
    if not os.path.exists(path):
        raise RuntimeError(f"{path} not found: run `python -c 'import __graft_entry__ as g; g.build()'` — the EqF path has no CPU fallback")
    lib = C.CDLL(path)
    vp = C.c_void_p
    P = C.POINTER
    protos = {
        "eqf_error_string": (C.c_char_p, [C.c_int]),
        "eqf_kernel_name": (C.c_char_p, [C.c_int]),
        "eqf_create": (C.c_int, [P(vp), C.c_int, C.c_int, C.c_int]),
        "eqf_destroy": (None, [vp]),
        "eqf_set_option": (C.c_int, [vp, C.c_int, C.c_int]),
        "eqf_get_option": (C.c_int, [vp, C.c_int, C.POINTER(C.c_int)]),
        "eqf_lookahead_selftest": (C.c_int, [vp]),
        "eqf_lookahead_home": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_long)]),
        "eqf_device_to_itself": (C.c_int, [vp]),
        "eqf_early_doorbell_stats": (C.c_int, [vp, C.POINTER(C.c_long), C.c_int]),
        "eqf_synchronize": (C.c_int, [vp]),
        "eqf_num_landmarks": (C.c_int, [vp]),
        "eqf_get_ids": (C.c_int, [vp, c_int_p, C.c_int]),
        "eqf_stream": (vp, [vp]),
        "eqf_set_state": (C.c_int, [vp, c_double_p, c_double_p, c_int_p, c_double_p, c_double_p, C.c_int]),
        "eqf_get_state": (C.c_int, [vp, c_double_p, c_double_p, c_int_p, c_double_p, c_double_p, C.c_int]),
        "eqf_set_sigma": (C.c_int, [vp, c_double_p, C.c_int]),
        "eqf_set_sigma_diag": (C.c_int, [vp, c_double_p, C.c_int]),
        "eqf_get_sigma": (C.c_int, [vp, c_double_p, C.c_int]),
        "eqf_get_sigma_block": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, c_double_p]),
        "eqf_state_estimate": (C.c_int, [vp, c_double_p, c_int_p, c_double_p, C.c_int]),
        "eqf_add_landmarks": (C.c_int, [vp, c_int_p, c_double_p, C.c_int, C.c_double]),
        "eqf_remove_landmarks": (C.c_int, [vp, c_int_p, C.c_int]),
        "eqf_remove_unmeasured_landmarks": (C.c_int, [vp, c_int_p, C.c_int, c_int_p, c_int_p]),
        "eqf_find_unknown_ids": (C.c_int, [vp, c_int_p, C.c_int, c_int_p, c_int_p]),
        "eqf_same_as_mapped": (C.c_int, [vp, c_int_p, C.c_int]),
        "eqf_update_unsettled": (C.c_int, [vp]),
        "eqf_remove_invalid_at_update": (C.c_int, [vp]),
        "eqf_add_landmarks_held": (C.c_int, [vp, c_int_p, c_double_p, C.c_int, C.c_double]),
        "eqf_hold_supported": (C.c_int, [vp]),
        "eqf_hold_stats": (C.c_int, [vp, C.POINTER(C.c_long), C.c_int]),
        "eqf_live_columns_stats": (C.c_int, [vp, C.POINTER(C.c_long), C.c_int]),
        "eqf_own_hardware_queue": (C.c_int, [vp]),
        "eqf_remove_invalid_landmarks": (C.c_int, [vp]),
        "eqf_integrate_riccati_fast": (C.c_int, [vp, c_double_p, C.c_double, c_double_p, c_double_p]),
        "eqf_integrate_riccati_accurate": (C.c_int, [vp, c_double_p, C.c_double, c_double_p, c_double_p]),
        "eqf_integrate_riccati_discrete": (C.c_int, [vp, c_double_p, C.c_double, c_double_p, c_double_p]),
        "eqf_integrate_observer": (C.c_int, [vp, c_double_p, c_double_p, C.c_int, C.c_int]),
        "eqf_outlier_stats": (C.c_int, [vp, P(Camera), c_int_p, c_double_p, C.c_int, c_double_p, c_double_p, c_double_p]),
        "eqf_propagate_fast": (C.c_int, [vp, c_double_p, C.c_double, c_double_p, c_double_p, c_double_p, c_double_p, C.c_int, C.c_int]),
        "eqf_stage_measurement": (C.c_int, [vp, c_int_p, c_double_p, C.c_int]),
        "eqf_trace_read": (C.c_int, [vp, C.POINTER(C.c_ulonglong), C.POINTER(C.c_longlong), C.POINTER(C.c_uint)]),
        "eqf_host_wait_stats": (C.c_int, [vp, C.POINTER(C.c_long), c_double_p, C.c_int]),
        "eqf_stats_then_update": (C.c_int, [vp, C.POINTER(Camera), c_int_p, c_double_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, c_double_p, c_double_p,
                                  c_double_p, c_int_p]),
        "eqf_stats_select_update": (C.c_int, [vp, C.POINTER(Camera), c_int_p, c_double_p, C.c_int, C.c_double, C.c_double, C.c_int, C.c_double, C.c_int, C.c_int, c_double_p,
                                    c_double_p, c_double_p, c_int_p, c_int_p, c_int_p]),
        "eqf_vision_update": (C.c_int, [vp, P(Camera), c_int_p, c_double_p, C.c_int, C.c_double, C.c_int, C.c_int]),
        "eqf_last_gamma": (C.c_int, [vp, c_double_p, C.c_int]),
        "eqf_compute_nees": (C.c_int, [vp, c_double_p, c_int_p, c_double_p, C.c_int, c_double_p]),
        "eqf_nees_lu_fallbacks": (C.c_int, [vp, C.POINTER(C.c_long)]),
        "eqf_speculation_stats": (C.c_int, [vp, C.POINTER(C.c_long), C.POINTER(C.c_long), C.POINTER(C.c_long), C.c_int]),
        "eqf_measure_in_propagate_stats": (C.c_int, [vp, C.POINTER(C.c_long), C.c_int]),
        "eqf_z_in_lookahead_stats": (C.c_int, [vp, C.POINTER(C.c_long), C.c_int]),
        "eqf_gather_stats": (C.c_int, [vp, C.POINTER(C.c_long), C.c_int]),
        "eqf_selection_stats": (C.c_int, [vp, C.POINTER(C.c_long), C.POINTER(C.c_long), C.c_int]),
        "eqf_output_cov_all": (C.c_int, [vp, P(Camera), c_double_p]),
        "eqf_lookahead_stats": (C.c_int, [vp, C.POINTER(C.c_long), C.POINTER(C.c_long), C.c_int]),
        "eqf_debug_matrices_AB": (C.c_int, [vp, c_double_p, c_double_p, c_double_p]),
        "eqf_debug_get_W": (C.c_int, [vp, c_double_p, C.c_int, C.c_int]),
        "eqf_debug_syrk_order": (C.c_int, [C.c_int, c_int_p]),
        "eqf_debug_lookahead_stamps": (C.c_int, [vp, C.POINTER(C.c_ulonglong)]),
        "eqf_debug_matrix_C": (C.c_int, [vp, P(Camera), c_int_p, c_double_p, C.c_int, C.c_int, c_double_p, c_double_p]),
        "eqf_mfma_f64_peak": (C.c_int, [vp, c_double_p]),
        "eqf_mfma_f64_peak_clock": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "eqf_last_kernel_times": (C.c_int, [vp, c_int_p, P(C.c_float), C.c_int]),
    }
    for name, (res, args) in protos.items():
        if os.environ.get("EQVIO_AMD_LIB_DIR") and name in ("eqf_lookahead_home", "eqf_device_to_itself", "eqf_early_doorbell_stats", "eqf_update_unsettled", "eqf_remove_invalid_at_update", "eqf_gather_stats", "eqf_remove_unmeasured_landmarks", "eqf_find_unknown_ids", "eqf_add_landmarks_held", "eqf_hold_supported", "eqf_hold_stats", "eqf_same_as_mapped", "eqf_live_columns_stats", "eqf_own_hardware_queue") and not hasattr(lib, name):
            continue  # same-box A/B against the libraries of an older commit (scripts/ab_builds.sh): entry points that commit did not have yet
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    lib._declared = sorted(protos)
    _lib = lib
    return lib


class EqfCore:
    """One eqf_ctx == one reference VIO_eqf (include/eqvio/mathematical/VIO_eqf.h:34-134) on a gfx950 device."""

    def __init__(self, max_landmarks, coordinate_choice=COORD_EUCLIDEAN, device=0):
        self.lib = load_eqf_lib()
        self.h = C.c_void_p()
        self._chk0(self.lib.eqf_create(C.byref(self.h), device, max_landmarks, coordinate_choice))
        self._cap0 = max_landmarks + 16

    def _chk0(self, rc):
        if rc != 0:
            raise EqfError(rc, self.lib.eqf_error_string(rc).decode())

    def close(self):
        if self.h:
            self.lib.eqf_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def N(self):
        return self.lib.eqf_num_landmarks(self.h)

    @property
    def n(self):
        return 21 + 3 * self.N

    @property
    def cap(self):  # size of the output arrays: the context grows on demand (eqf_add_landmarks / eqf_set_state), so ask it
        return max(self._cap0, self.N + 16)

    def set_option(self, opt, val):
        self._chk0(self.lib.eqf_set_option(self.h, opt, val))

    def get_option(self, opt):
        v = C.c_int()
        self._chk0(self.lib.eqf_get_option(self.h, opt, C.byref(v)))
        return v.value

    def lookahead_selftest(self):
        return self.lib.eqf_lookahead_selftest(self.h)

    def synchronize(self):
        self._chk0(self.lib.eqf_synchronize(self.h))

    def stream(self):
        return self.lib.eqf_stream(self.h)

    def set_state(self, xi0_sensor, X_sensor, ids, q0, Q):
        xi0_sensor, X_sensor, ids, q0, Q = _f64(xi0_sensor), _f64(X_sensor), _i32(ids), _f64(q0), _f64(Q)
        self._chk0(self.lib.eqf_set_state(self.h, _dp(xi0_sensor), _dp(X_sensor), _ip(ids), _dp(q0), _dp(Q), len(ids)))

    def get_state(self):
        xi0, Xs = np.zeros(23), np.zeros(23)
        ids, q0, Q = np.zeros(self.cap, np.int32), np.zeros(3 * self.cap), np.zeros(5 * self.cap)
        N = self.lib.eqf_get_state(self.h, _dp(xi0), _dp(Xs), _ip(ids), _dp(q0), _dp(Q), self.cap)
        if N < 0:
            self._chk0(N)
        return xi0, Xs, ids[:N].copy(), q0[: 3 * N].reshape(N, 3).copy(), Q[: 5 * N].reshape(N, 5).copy()

    def set_sigma(self, S):
        S = np.asfortranarray(S, dtype=np.float64)
        self._chk0(self.lib.eqf_set_sigma(self.h, S.ctypes.data_as(c_double_p), S.shape[0]))

    def set_sigma_diag(self, d):
        d = _f64(d)
        self._chk0(self.lib.eqf_set_sigma_diag(self.h, _dp(d), len(d)))

    def get_sigma(self):
        n = self.n
        out = np.zeros((n, n), order="F")
        self._chk0(self.lib.eqf_get_sigma(self.h, out.ctypes.data_as(c_double_p), n))
        return out

    def get_sigma_block(self, r0, c0, rows, cols):
        out = np.zeros((rows, cols), order="F")
        self._chk0(self.lib.eqf_get_sigma_block(self.h, r0, c0, rows, cols, out.ctypes.data_as(c_double_p)))
        return out

    def state_estimate(self):
        s = np.zeros(23)
        ids, p = np.zeros(self.cap, np.int32), np.zeros(3 * self.cap)
        N = self.lib.eqf_state_estimate(self.h, _dp(s), _ip(ids), _dp(p), self.cap)
        if N < 0:
            self._chk0(N)
        return s, ids[:N].copy(), p[: 3 * N].reshape(N, 3).copy()

    def add_landmarks(self, ids, p, var):
        ids, p = _i32(ids), _f64(p)
        self._chk0(self.lib.eqf_add_landmarks(self.h, _ip(ids), _dp(p), len(ids), var))

    def add_landmarks_held(self, ids, p, var):
        """eqf_add_landmarks_held: the landmarks belong to the time behind the next propagate_fast, which passes them through untouched. Returns False when refused."""
        ids, p = _i32(ids), _f64(p)
        rc = self.lib.eqf_add_landmarks_held(self.h, _ip(ids), _dp(p), len(ids), var)
        if rc == -6:
            return False
        self._chk0(rc)
        return True

    def remove_landmarks(self, indices):
        indices = _i32(indices)
        self._chk0(self.lib.eqf_remove_landmarks(self.h, _ip(indices), len(indices)))

    def remove_invalid_landmarks(self):
        rc = self.lib.eqf_remove_invalid_landmarks(self.h)
        if rc < 0:
            self._chk0(rc)
        return rc

    def integrate_riccati_fast(self, imu13, dt, Qdiag12, Pdiag8):
        imu13, Qd, Pd = _f64(imu13), _f64(Qdiag12), _f64(Pdiag8)
        self._chk0(self.lib.eqf_integrate_riccati_fast(self.h, _dp(imu13), dt, _dp(Qd), _dp(Pd)))

    def integrate_riccati_accurate(self, imu13, dt, Qdiag12, Pdiag8):
        imu13, Qd, Pd = _f64(imu13), _f64(Qdiag12), _f64(Pdiag8)
        self._chk0(self.lib.eqf_integrate_riccati_accurate(self.h, _dp(imu13), dt, _dp(Qd), _dp(Pd)))

    def integrate_riccati_discrete(self, imu13, dt, Qdiag12, Pdiag8):
        imu13, Qdiag12, Pdiag8 = _f64(imu13), _f64(Qdiag12), _f64(Pdiag8)
        self._chk0(self.lib.eqf_integrate_riccati_discrete(self.h, _dp(imu13), dt, _dp(Qdiag12), _dp(Pdiag8)))

    def integrate_observer(self, imu13_k, dt_k, discrete=True):
        imu13_k, dt_k = _f64(imu13_k).reshape(-1, 13), _f64(dt_k).reshape(-1)
        self._chk0(self.lib.eqf_integrate_observer(self.h, _dp(imu13_k), _dp(dt_k), len(dt_k), int(discrete)))

    def outlier_stats(self, cam, ids, y):
        ids, y = _i32(ids), _f64(y)
        N = self.N
        a, p, d = np.zeros(N), np.zeros(N), np.zeros(N)
        self._chk0(self.lib.eqf_outlier_stats(self.h, C.byref(cam), _ip(ids), _dp(y), len(ids), _dp(a), _dp(p), _dp(d)))
        return a, p, d

    def output_cov_all(self, cam):
        """eqf_output_cov_all: getOutputCovById of every landmark, state order, (N, 2, 2)."""
        out = np.zeros(4 * self.N)
        self._chk0(self.lib.eqf_output_cov_all(self.h, C.byref(cam), _dp(out)))
        return out.reshape(self.N, 2, 2)

    def vision_update(self, cam, ids, y, meas_var, use_equivariant=True, discrete=False):
        ids, y = _i32(ids), _f64(y)
        self._chk0(self.lib.eqf_vision_update(self.h, C.byref(cam), _ip(ids), _dp(y), len(ids), meas_var, int(use_equivariant), int(discrete)))

    def propagate_fast(self, imu13_mean, dt_total, Qdiag12, Pdiag8, imu13_k, dt_k, discrete=True):
        """eqf_propagate_fast: fast Riccati at the current X + all observer steps, one call."""
        m, Qd, Pd = _f64(imu13_mean), _f64(Qdiag12), _f64(Pdiag8)
        imus, dts = _f64(np.atleast_2d(imu13_k)).reshape(-1), _f64(dt_k)
        self._chk0(self.lib.eqf_propagate_fast(self.h, _dp(m), dt_total, _dp(Qd), _dp(Pd), _dp(imus), _dp(dts), len(dts), int(discrete)))

    def stage_measurement(self, ids, y):
        """eqf_stage_measurement: hint ahead of the propagation call of the same frame."""
        ids, y = _i32(ids), _f64(y)
        self._chk0(self.lib.eqf_stage_measurement(self.h, _ip(ids), _dp(y), len(ids)))

    def stats_then_update(self, cam, ids, y, thr_abs, thr_prob, meas_var, use_equivariant=True, discrete=False):
        """eqf_stats_then_update: returns (updated, absErr, probErr, depth2)."""
        ids, y = _i32(ids), _f64(y)
        a, p, d = np.zeros(self.N), np.zeros(self.N), np.zeros(self.N)
        upd = C.c_int(0)
        self._chk0(self.lib.eqf_stats_then_update(self.h, C.byref(cam), _ip(ids), _dp(y), len(ids), thr_abs, thr_prob, meas_var, int(use_equivariant), int(discrete), _dp(a),
                                                  _dp(p), _dp(d), C.byref(upd)))
        return upd.value, a, p, d  # 1 updated, 0 cancelled on the device, -1 not applicable

    def stats_select_update(self, cam, ids, y, thr_abs, thr_prob, max_outliers, meas_var, use_equivariant=True, discrete=False):
        """eqf_stats_select_update: returns (updated, absErr, probErr, depth2, former state indices of the discarded landmarks)."""
        ids, y = _i32(ids), _f64(y)
        n0 = self.N
        a, p, d = np.zeros(n0), np.zeros(n0), np.zeros(n0)
        upd, nrm = C.c_int(0), C.c_int(0)
        rm = np.zeros(max(n0, 1), np.int32)
        self._chk0(self.lib.eqf_stats_select_update(self.h, C.byref(cam), _ip(ids), _dp(y), len(ids), thr_abs, thr_prob, int(max_outliers), meas_var, int(use_equivariant),
                                                    int(discrete), _dp(a), _dp(p), _dp(d), C.byref(upd), _ip(rm), C.byref(nrm)))
        return upd.value, a, p, d, rm[: nrm.value].copy()

    def last_gamma(self):
        out = np.zeros(self.n + 64)
        k = self.lib.eqf_last_gamma(self.h, _dp(out), len(out))
        if k < 0:
            self._chk0(k)
        return out[:k].copy()

    def compute_nees(self, sensor, ids, p):
        sensor, ids, p = _f64(sensor), _i32(ids), _f64(p)
        out = C.c_double()
        self._chk0(self.lib.eqf_compute_nees(self.h, _dp(sensor), _ip(ids), _dp(p), len(ids), C.byref(out)))
        return out.value

    def nees_lu_fallbacks(self):
        out = C.c_long()
        self._chk0(self.lib.eqf_nees_lu_fallbacks(self.h, C.byref(out)))
        return out.value

    def debug_get_W(self, rows, cols):
        out = np.zeros((rows, cols), order="F")
        self._chk0(self.lib.eqf_debug_get_W(self.h, out.ctypes.data_as(c_double_p), rows, cols))
        return out

    def debug_matrices_AB(self, imu13):
        imu13 = _f64(imu13)
        n = self.n
        A, B = np.zeros((n, n), order="F"), np.zeros((n, 12), order="F")
        self._chk0(self.lib.eqf_debug_matrices_AB(self.h, _dp(imu13), A.ctypes.data_as(c_double_p), B.ctypes.data_as(c_double_p)))
        return A, B

    def debug_matrix_C(self, cam, ids, y, use_equivariant=True):
        ids, y = _i32(ids), _f64(y)
        M, n = len(ids), self.n
        Cm, yt = np.zeros((2 * M, n), order="F"), np.zeros(2 * M)
        self._chk0(self.lib.eqf_debug_matrix_C(self.h, C.byref(cam), _ip(ids), _dp(y), M, int(use_equivariant), Cm.ctypes.data_as(c_double_p), _dp(yt)))
        return Cm, yt

    def mfma_f64_peak(self):
        t = C.c_double()
        self._chk0(self.lib.eqf_mfma_f64_peak(self.h, C.byref(t)))
        return t.value

    def kernel_times(self, cap=4096):
        which = np.zeros(cap, np.int32)
        us = np.zeros(cap, np.float32)
        k = self.lib.eqf_last_kernel_times(self.h, _ip(which), us.ctypes.data_as(C.POINTER(C.c_float)), cap)
        return [(self.lib.eqf_kernel_name(int(which[i])).decode(), float(us[i])) for i in range(k)]


# ---------------------------------------------------------------------------------------------------------
# include/eqvio_filter.h : the host-side VIOFilter mirror (libeqvio_filter.so)
_flib = None


def load_filter_lib():
    global _flib
    if _flib is not None:
        return _flib
    load_eqf_lib()
    path = os.path.join(LIB_DIR, "libeqvio_filter.so")
    if not os.path.exists(path):
        raise RuntimeError(f"{path} not found: run __graft_entry__.build()")
    lib = C.CDLL(path)
    vp, P = C.c_void_p, C.POINTER
    protos = {
        "eqvio_filter_create": (C.c_int, [P(vp), P(Settings), C.c_int, C.c_int]),
        "eqvio_filter_create_from_state": (C.c_int, [P(vp), P(Settings), C.c_int, C.c_int, c_double_p, c_int_p, c_double_p, C.c_int, C.c_double]),
        "eqvio_filter_destroy": (None, [vp]),
        "eqvio_filter_last_error": (C.c_char_p, [vp]),
        "eqvio_filter_process_imu": (C.c_int, [vp, c_double_p]),
        "eqvio_filter_process_vision": (C.c_int, [vp, C.c_double, P(Camera), c_int_p, c_double_p, C.c_int]),
        "eqvio_filter_state_estimate": (C.c_int, [vp, c_double_p, c_int_p, c_double_p, C.c_int]),
        "eqvio_filter_get_time": (C.c_double, [vp]),
        "eqvio_filter_is_initialised": (C.c_int, [vp]),
        "eqvio_filter_set_state": (C.c_int, [vp, c_double_p, c_int_p, c_double_p, C.c_int]),
        "eqvio_filter_set_landmarks": (C.c_int, [vp, c_int_p, c_double_p, C.c_int]),
        "eqvio_filter_augment_landmark_states": (C.c_int, [vp, c_int_p, C.c_int, c_double_p, c_int_p, c_double_p, C.c_int]),
        "eqvio_filter_get_eqf": (C.c_int, [vp, c_double_p, c_double_p, c_int_p, c_double_p, c_double_p, C.c_int]),
        "eqvio_filter_sigma_dim": (C.c_int, [vp]),
        "eqvio_filter_get_sigma": (C.c_int, [vp, c_double_p, C.c_int]),
        "eqvio_filter_compute_nees": (C.c_int, [vp, c_double_p, c_int_p, c_double_p, C.c_int, c_double_p]),
        "eqvio_filter_get_feature_predictions": (C.c_int, [vp, P(Camera), C.c_double, c_int_p, c_double_p, C.c_int]),
        "eqvio_filter_core": (vp, [vp]),
        "eqvio_filter_last_timing": (C.c_int, [vp, c_double_p, c_double_p, c_double_p]),
        "eqvio_filter_run_frames": (C.c_int, [vp, P(Camera), C.c_int, c_int_p, c_double_p, c_double_p, c_int_p, c_int_p, c_double_p]),
        "eqvio_filter_run_prepared": (C.c_int, [vp, vp, C.c_int, C.c_int]),
    }
    for name, (res, args) in protos.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    lib._declared = sorted(protos)
    _flib = lib
    return lib


class VIOFilter:
    """The reference's class VIOFilter (include/eqvio/VIOFilter.h:36-192) on one MI355X, through include/eqvio_filter.h."""

    def __init__(self, settings, max_landmarks=256, device=0, sensor=None, ids=None, p=None, time=0.0):
        self.lib = load_filter_lib()
        self.h = C.c_void_p()
        self._cap0 = max_landmarks + 64
        if sensor is None:
            rc = self.lib.eqvio_filter_create(C.byref(self.h), C.byref(settings), device, max_landmarks)
        else:
            sensor, ids, p = _f64(sensor), _i32(ids), _f64(p)
            rc = self.lib.eqvio_filter_create_from_state(C.byref(self.h), C.byref(settings), device, max_landmarks, _dp(sensor), _ip(ids), _dp(p), len(ids), time)
        self._chk(rc)

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError("eqvio_filter: " + self.lib.eqvio_filter_last_error(self.h).decode())

    def close(self):
        if self.h:
            self.lib.eqvio_filter_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def process_imu(self, imu13):
        imu13 = _f64(imu13)
        self._chk(self.lib.eqvio_filter_process_imu(self.h, _dp(imu13)))

    def process_vision(self, stamp, cam, ids, y):
        ids, y = _i32(ids), _f64(y)
        self._chk(self.lib.eqvio_filter_process_vision(self.h, stamp, C.byref(cam), _ip(ids), _dp(y), len(ids)))

    def state_estimate(self):
        s, ids, p = np.zeros(23), np.zeros(self.cap, np.int32), np.zeros(3 * self.cap)
        N = self.lib.eqvio_filter_state_estimate(self.h, _dp(s), _ip(ids), _dp(p), self.cap)
        if N < 0:
            self._chk(-1)
        return s, ids[:N].copy(), p[: 3 * N].reshape(N, 3).copy()

    def get_time(self):
        return self.lib.eqvio_filter_get_time(self.h)

    def is_initialised(self):
        return bool(self.lib.eqvio_filter_is_initialised(self.h))

    def set_state(self, sensor, ids, p):
        sensor, ids, p = _f64(sensor), _i32(ids), _f64(p)
        self._chk(self.lib.eqvio_filter_set_state(self.h, _dp(sensor), _ip(ids), _dp(p), len(ids)))

    def set_landmarks(self, ids, p):
        ids, p = _i32(ids), _f64(p)
        self._chk(self.lib.eqvio_filter_set_landmarks(self.h, _ip(ids), _dp(p), len(ids)))

    def augment_landmark_states(self, new_ids, sensor, ids, p):
        new_ids, sensor, ids, p = _i32(new_ids), _f64(sensor), _i32(ids), _f64(p)
        self._chk(self.lib.eqvio_filter_augment_landmark_states(self.h, _ip(new_ids), len(new_ids), _dp(sensor), _ip(ids), _dp(p), len(ids)))

    def get_eqf(self):
        xi0, Xs = np.zeros(23), np.zeros(23)
        ids, q0, Q = np.zeros(self.cap, np.int32), np.zeros(3 * self.cap), np.zeros(5 * self.cap)
        N = self.lib.eqvio_filter_get_eqf(self.h, _dp(xi0), _dp(Xs), _ip(ids), _dp(q0), _dp(Q), self.cap)
        if N < 0:
            raise RuntimeError("eqvio_filter_get_eqf failed")
        return xi0, Xs, ids[:N].copy(), q0[: 3 * N].reshape(N, 3).copy(), Q[: 5 * N].reshape(N, 5).copy()

    def force_eqf(self, xi0_sensor, X_sensor, ids, q0, Q, Sigma):
        """Teacher forcing (SURVEY.md section 8(d) "Parity definition"): overwrite (xi0, X, Sigma) of the device context behind viewEqFState() through
        eqvio_filter_core() + eqf_set_state / eqf_set_sigma, e.g. with the oracle's, so that the next frame starts from exactly the reference's state. The
        landmark ids must be the filter's own (same set, same order): the host mirror's bookkeeping is not touched."""
        lib = load_eqf_lib()
        xi0_sensor, X_sensor, ids, q0, Q = _f64(xi0_sensor), _f64(X_sensor), _i32(ids), _f64(q0), _f64(Q)
        S = np.asfortranarray(Sigma, dtype=np.float64)
        core = self.core_handle()
        if lib.eqf_set_state(core, _dp(xi0_sensor), _dp(X_sensor), _ip(ids), _dp(q0), _dp(Q), len(ids)) != 0 or lib.eqf_set_sigma(core, S.ctypes.data_as(c_double_p), S.shape[0]) != 0:
            raise RuntimeError("force_eqf: eqf_set_state / eqf_set_sigma failed")

    def get_sigma(self):
        n = self.lib.eqvio_filter_sigma_dim(self.h)
        out = np.zeros((n, n), order="F")
        if self.lib.eqvio_filter_get_sigma(self.h, out.ctypes.data_as(c_double_p), n) != 0:
            raise RuntimeError("eqvio_filter_get_sigma failed")
        return out

    def compute_nees(self, sensor, ids, p):
        sensor, ids, p = _f64(sensor), _i32(ids), _f64(p)
        out = C.c_double()
        self._chk(self.lib.eqvio_filter_compute_nees(self.h, _dp(sensor), _ip(ids), _dp(p), len(ids), C.byref(out)))
        return out.value

    def get_feature_predictions(self, cam, stamp):
        ids, y = np.zeros(self.cap, np.int32), np.zeros(2 * self.cap)
        k = self.lib.eqvio_filter_get_feature_predictions(self.h, C.byref(cam), stamp, _ip(ids), _dp(y), self.cap)
        if k < 0:
            self._chk(-1)
        return ids[:k].copy(), y[:2 * k].copy()

    def core_handle(self):
        return self.lib.eqvio_filter_core(self.h)

    @property
    def cap(self):  # output array sizes follow the filter: its landmark capacity grows on demand
        return max(self._cap0, (self.sigma_dim() - 21) // 3 + 64)

    def sigma_dim(self):
        return self.lib.eqvio_filter_sigma_dim(self.h)

    def set_core_option(self, opt, val):
        """eqf_set_option on the device context behind viewEqFState() (e.g. OPT_SIGMA_FP32)."""
        if load_eqf_lib().eqf_set_option(self.core_handle(), opt, val) != 0:
            raise RuntimeError("eqf_set_option failed")

    def synchronize(self):
        load_eqf_lib().eqf_synchronize(self.core_handle())

    def last_timing(self):
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        self.lib.eqvio_filter_last_timing(self.h, C.byref(a), C.byref(b), C.byref(c))
        return {"propagation": a.value, "preprocessing": b.value, "correction": c.value}

    def run_prepared(self, frames, first=0, count=None):
        """eqvio_filter_run_prepared on a PreparedFrames object: no container construction inside the call."""
        count = len(frames) - first if count is None else count
        done = self.lib.eqvio_filter_run_prepared(self.h, frames.h, first, count)
        if done < 0:
            self._chk(-1)
        return done

    def run_frames(self, cam, imu_counts, imu13_all, stamps, meas_counts, ids_all, y_all):
        imu_counts, imu13_all, stamps = _i32(imu_counts), _f64(imu13_all), _f64(stamps)
        meas_counts, ids_all, y_all = _i32(meas_counts), _i32(ids_all), _f64(y_all)
        done = self.lib.eqvio_filter_run_frames(self.h, C.byref(cam), len(stamps), _ip(imu_counts), _dp(imu13_all), _dp(stamps), _ip(meas_counts), _ip(ids_all), _dp(y_all))
        if done < 0:
            self._chk(-1)
        return done


class PreparedFrames:
    """eqvio_frames (include/eqvio_filter.h): IMU samples and VisionMeasurement objects built once from flat arrays."""

    def __init__(self, cam, imu_counts, imu13_all, stamps, meas_counts, ids_all, y_all):
        lib = load_filter_lib()
        lib.eqvio_frames_create.restype = C.c_void_p
        lib.eqvio_frames_create.argtypes = [C.POINTER(Camera), C.c_int, c_int_p, c_double_p, c_double_p, c_int_p, c_int_p, c_double_p]
        lib.eqvio_frames_destroy.restype = None
        lib.eqvio_frames_destroy.argtypes = [C.c_void_p]
        lib.eqvio_frames_count.restype = C.c_int
        lib.eqvio_frames_count.argtypes = [C.c_void_p]
        imu_counts, imu13_all, stamps = _i32(imu_counts), _f64(imu13_all), _f64(stamps)
        meas_counts, ids_all, y_all = _i32(meas_counts), _i32(ids_all), _f64(y_all)
        self.lib = lib
        self.h = lib.eqvio_frames_create(C.byref(cam), len(stamps), _ip(imu_counts), _dp(imu13_all), _dp(stamps), _ip(meas_counts), _ip(ids_all), _dp(y_all))
        if not self.h:
            raise RuntimeError("eqvio_frames_create failed")

    def __len__(self):
        return self.lib.eqvio_frames_count(self.h)

    def edit_id(self, frame, k, new_id):
        """eqvio_frames_edit_id: the k-th feature of the frame's public map gets another id (erase + insert)."""
        self.lib.eqvio_frames_edit_id.restype = C.c_int
        self.lib.eqvio_frames_edit_id.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        if self.lib.eqvio_frames_edit_id(self.h, frame, k, new_id) != 0:
            raise IndexError("eqvio_frames_edit_id")

    def edit_pixel(self, frame, k, u, v):
        """eqvio_frames_edit_pixel: write a pixel through the measurement's public map, as a caller of the reference's type may."""
        self.lib.eqvio_frames_edit_pixel.restype = C.c_int
        self.lib.eqvio_frames_edit_pixel.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double]
        if self.lib.eqvio_frames_edit_pixel(self.h, frame, k, u, v) != 0:
            raise IndexError("eqvio_frames_edit_pixel")

    def close(self):
        if self.h:
            self.lib.eqvio_frames_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------------------------------------------------
# include/eqvio_sim.h : the synthetic-world data server (host only, also in libeqvio_filter.so)
class SimSettings(C.Structure):
    _fields_ = [("numPoints", C.c_int), ("wallDistance", C.c_double), ("randomSeed", C.c_uint), ("numWalls", C.c_int), ("maxFeatures", C.c_int),
                ("initialNoise", C.c_int), ("inputNoise", C.c_int), ("outputNoise", C.c_int), ("duration", C.c_double), ("trajectory", C.c_int),
                ("imuFreq", C.c_double), ("imageFreq", C.c_double)]

    TRAJECTORIES = {"wave": 0, "square": 1, "sine": 2, "line": 3}

    @classmethod
    def defaults(cls, **kw):
        s = cls()
        _load_sim_protos().eqvio_sim_default_settings(C.byref(s))
        for k, v in kw.items():
            setattr(s, k, cls.TRAJECTORIES[v] if k == "trajectory" and isinstance(v, str) else v)
        return s


def _load_sim_protos():
    lib = load_filter_lib()
    if getattr(lib, "_sim_declared", None):
        return lib
    vp, P = C.c_void_p, C.POINTER
    protos = {
        "eqvio_sim_default_settings": (None, [P(SimSettings)]),
        "eqvio_sim_create": (vp, [P(SimSettings), P(Settings)]),
        "eqvio_sim_destroy": (None, [vp]),
        "eqvio_sim_next_measurement_type": (C.c_int, [vp]),
        "eqvio_sim_next_time": (C.c_double, [vp]),
        "eqvio_sim_get_imu": (C.c_int, [vp, c_double_p]),
        "eqvio_sim_get_vision": (C.c_int, [vp, c_double_p, c_int_p, c_double_p, C.c_int]),
        "eqvio_sim_true_state": (C.c_int, [vp, C.c_double, C.c_int, c_double_p, c_int_p, c_double_p, C.c_int]),
        "eqvio_sim_num_points": (C.c_int, [vp]),
        "eqvio_sim_camera": (None, [vp, P(Camera)]),
        "eqvio_sim_camera_offset": (None, [vp, c_double_p]),
        "eqvio_camera_project": (None, [P(Camera), c_double_p, c_double_p]),
        "eqvio_camera_undistort": (None, [P(Camera), c_double_p, c_double_p]),
        "eqvio_camera_jacobian": (None, [P(Camera), c_double_p, c_double_p]),
    }
    for name, (res, args) in protos.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    lib._sim_declared = sorted(protos)
    return lib


class SimulationDataServer:
    """The reference's SimulationDataServer + VIOSimulator (SimulationDataServer.h:25-70, VIOSimulator.h:29-106)."""
    IMAGE, IMU, NONE = 0, 1, 2

    def __init__(self, sim_settings, filter_settings):
        self.lib = _load_sim_protos()
        self.h = self.lib.eqvio_sim_create(C.byref(sim_settings), C.byref(filter_settings))
        if not self.h:
            raise RuntimeError("eqvio_sim_create failed")
        self.num_points = self.lib.eqvio_sim_num_points(self.h)
        self.max_features = sim_settings.maxFeatures
        self.cam = Camera()
        self.lib.eqvio_sim_camera(self.h, C.byref(self.cam))

    def close(self):
        if getattr(self, "h", None):
            self.lib.eqvio_sim_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def next_measurement_type(self):
        return self.lib.eqvio_sim_next_measurement_type(self.h)

    def next_time(self):
        return self.lib.eqvio_sim_next_time(self.h)

    def get_imu(self):
        out = np.zeros(13)
        self.lib.eqvio_sim_get_imu(self.h, _dp(out))
        return out

    def get_vision(self):
        cap = max(self.max_features, 1)
        stamp, ids, y = C.c_double(), np.zeros(cap, np.int32), np.zeros(2 * cap)
        m = self.lib.eqvio_sim_get_vision(self.h, C.byref(stamp), _ip(ids), _dp(y), cap)
        if m < 0:
            raise RuntimeError("eqvio_sim_get_vision: capacity")
        return stamp.value, ids[:m].copy(), y[:2 * m].copy()

    def true_state(self, stamp, with_noise=False):
        cap = self.num_points
        s, ids, p = np.zeros(23), np.zeros(cap, np.int32), np.zeros(3 * cap)
        n = self.lib.eqvio_sim_true_state(self.h, stamp, int(with_noise), _dp(s), _ip(ids), _dp(p), cap)
        if n < 0:
            raise RuntimeError("eqvio_sim_true_state: capacity")
        return s, ids[:n].copy(), p[:3 * n].reshape(n, 3).copy()

    def camera_offset(self):
        out = np.zeros(7)
        self.lib.eqvio_sim_camera_offset(self.h, _dp(out))
        return out
