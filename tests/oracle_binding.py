"""ctypes binding of oracle/liboracle.so — the CPU restatement of the reference (TEST INFRASTRUCTURE).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module."""
import ctypes as C
import os
import subprocess

import numpy as np

from eqvio_amd.capi import Camera, Settings, _dp, _f64, _i32, _ip, c_double_p, c_int_p

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

ARITH_AS_WRITTEN, ARITH_REFERENCE, ARITH_EFFICIENT = 0, 1, 2

_lib = None


def build_oracle():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def load_oracle():
    global _lib
    if _lib is not None:
        return _lib
    path = os.path.join(ORACLE_DIR, "liboracle.so")
    if not os.path.exists(path):
        build_oracle()
    lib = C.CDLL(path)
    vp, P = C.c_void_p, C.POINTER
    protos = {
        "orc_filter_create": (vp, [P(Settings)]),
        "orc_filter_create_from_state": (vp, [P(Settings), c_double_p, c_int_p, c_double_p, C.c_int, C.c_double]),
        "orc_filter_destroy": (None, [vp]),
        "orc_filter_set_arithmetic": (None, [vp, C.c_int]),
        "orc_filter_process_imu": (None, [vp, c_double_p]),
        "orc_filter_process_vision": (None, [vp, C.c_double, P(Camera), c_int_p, c_double_p, C.c_int]),
        "orc_filter_state_estimate": (C.c_int, [vp, c_double_p, c_int_p, c_double_p, C.c_int]),
        "orc_filter_get_time": (C.c_double, [vp]),
        "orc_filter_is_initialised": (C.c_int, [vp]),
        "orc_filter_set_state": (None, [vp, c_double_p, c_int_p, c_double_p, C.c_int]),
        "orc_filter_set_landmarks": (None, [vp, c_int_p, c_double_p, C.c_int]),
        "orc_filter_augment_landmark_states": (None, [vp, c_int_p, C.c_int, c_double_p, c_int_p, c_double_p, C.c_int]),
        "orc_filter_get_eqf": (C.c_int, [vp, c_double_p, c_double_p, c_int_p, c_double_p, c_double_p, C.c_int]),
        "orc_filter_set_eqf": (None, [vp, c_double_p, c_double_p, c_int_p, c_double_p, c_double_p, C.c_int, c_double_p, C.c_double]),
        "orc_filter_get_sigma": (C.c_int, [vp, c_double_p, C.c_int]),
        "orc_filter_sigma_dim": (C.c_int, [vp]),
        "orc_filter_last_gamma": (C.c_int, [vp, c_double_p, C.c_int]),
        "orc_eqf_integrate_riccati_fast": (None, [vp, c_double_p, C.c_double]),
        "orc_eqf_integrate_riccati_accurate": (None, [vp, c_double_p, C.c_double]),
        "orc_filter_get_feature_predictions": (C.c_int, [vp, P(Camera), C.c_double, c_int_p, c_double_p, C.c_int]),
        "orc_cam_project": (None, [P(Camera), c_double_p, c_double_p]),
        "orc_cam_undistort": (None, [P(Camera), c_double_p, c_double_p]),
        "orc_cam_jacobian": (None, [P(Camera), c_double_p, c_double_p]),
        "orc_eqf_integrate_riccati_discrete": (None, [vp, c_double_p, C.c_double]),
        "orc_eqf_integrate_observer": (None, [vp, c_double_p, C.c_double, C.c_int]),
        "orc_eqf_vision_update": (None, [vp, C.c_double, P(Camera), c_int_p, c_double_p, C.c_int]),
        "orc_eqf_remove_landmark_by_index": (None, [vp, C.c_int]),
        "orc_eqf_add_landmarks": (None, [vp, c_int_p, c_double_p, C.c_int, C.c_double]),
        "orc_eqf_compute_nees": (C.c_double, [vp, c_double_p, c_int_p, c_double_p, C.c_int]),
        "orc_filter_outlier_stats": (None, [vp, P(Camera), c_int_p, c_double_p, C.c_int, c_double_p, c_double_p]),
        "orc_filter_output_cov_all": (None, [vp, P(Camera), c_double_p]),
        "orc_state_matrix_A": (C.c_int, [vp, c_double_p, c_double_p, C.c_int]),
        "orc_input_matrix_B": (C.c_int, [vp, c_double_p, C.c_int]),
        "orc_output_matrix_C": (C.c_int, [vp, P(Camera), c_int_p, c_double_p, C.c_int, C.c_int, c_double_p, C.c_int]),
        "orc_state_matrix_A_discrete": (C.c_int, [vp, c_double_p, C.c_double, c_double_p, C.c_int]),
        "orc_state_chart": (C.c_int, [vp, c_double_p, c_int_p, c_double_p, C.c_int, c_double_p, C.c_int]),
        "orc_integrate_system_function": (None, [c_double_p, c_int_p, c_double_p, C.c_int, c_double_p, C.c_double]),
        "orc_se3_log_dist": (C.c_double, [c_double_p, c_double_p]),
        "orc_bench_frame": (C.c_double, [vp, c_double_p, c_double_p, C.c_int, C.c_double, P(Camera), c_int_p, c_double_p, C.c_int, C.c_int, C.c_int]),
    }
    for name, (res, args) in protos.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class OracleFilter:
    """orc::VIOFilter (oracle/vio.hpp) == the reference's VIOFilter + VIO_eqf on the CPU."""

    def __init__(self, settings, sensor=None, ids=None, p=None, time=0.0):
        self.lib = load_oracle()
        self.settings = settings
        if sensor is None:
            self.h = self.lib.orc_filter_create(C.byref(settings))
        else:
            sensor, ids, p = _f64(sensor), _i32(ids), _f64(p)
            self.h = self.lib.orc_filter_create_from_state(C.byref(settings), _dp(sensor), _ip(ids), _dp(p), len(ids), time)
        self.cap = 4096

    def close(self):
        if self.h:
            self.lib.orc_filter_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_arithmetic(self, mode):
        self.lib.orc_filter_set_arithmetic(self.h, mode)

    # ---- VIOFilter API
    def process_imu(self, imu13):
        imu13 = _f64(imu13)
        self.lib.orc_filter_process_imu(self.h, _dp(imu13))

    def process_vision(self, stamp, cam, ids, y):
        ids, y = _i32(ids), _f64(y)
        self.lib.orc_filter_process_vision(self.h, stamp, C.byref(cam), _ip(ids), _dp(y), len(ids))

    def state_estimate(self):
        s, ids, p = np.zeros(23), np.zeros(self.cap, np.int32), np.zeros(3 * self.cap)
        N = self.lib.orc_filter_state_estimate(self.h, _dp(s), _ip(ids), _dp(p), self.cap)
        assert N >= 0
        return s, ids[:N].copy(), p[: 3 * N].reshape(N, 3).copy()

    def get_time(self):
        return self.lib.orc_filter_get_time(self.h)

    def get_feature_predictions(self, cam, stamp):
        ids, y = np.zeros(self.cap, np.int32), np.zeros(2 * self.cap)
        k = self.lib.orc_filter_get_feature_predictions(self.h, C.byref(cam), stamp, _ip(ids), _dp(y), self.cap)
        assert k >= 0
        return ids[:k].copy(), y[:2 * k].copy()

    def set_state(self, sensor, ids, p):
        sensor, ids, p = _f64(sensor), _i32(ids), _f64(p)
        self.lib.orc_filter_set_state(self.h, _dp(sensor), _ip(ids), _dp(p), len(ids))

    def set_landmarks(self, ids, p):
        ids, p = _i32(ids), _f64(p)
        self.lib.orc_filter_set_landmarks(self.h, _ip(ids), _dp(p), len(ids))

    def augment_landmark_states(self, new_ids, sensor, ids, p):
        new_ids, sensor, ids, p = _i32(new_ids), _f64(sensor), _i32(ids), _f64(p)
        self.lib.orc_filter_augment_landmark_states(self.h, _ip(new_ids), len(new_ids), _dp(sensor), _ip(ids), _dp(p), len(ids))

    # ---- VIO_eqf state
    def get_eqf(self):
        xi0, Xs = np.zeros(23), np.zeros(23)
        ids, q0, Q = np.zeros(self.cap, np.int32), np.zeros(3 * self.cap), np.zeros(5 * self.cap)
        N = self.lib.orc_filter_get_eqf(self.h, _dp(xi0), _dp(Xs), _ip(ids), _dp(q0), _dp(Q), self.cap)
        assert N >= 0
        return xi0, Xs, ids[:N].copy(), q0[: 3 * N].reshape(N, 3).copy(), Q[: 5 * N].reshape(N, 5).copy()

    def set_eqf(self, xi0, Xs, ids, q0, Q, Sigma, time=0.0):
        xi0, Xs, ids, q0, Q = _f64(xi0), _f64(Xs), _i32(ids), _f64(q0), _f64(Q)
        S = np.asfortranarray(Sigma, dtype=np.float64)
        self.lib.orc_filter_set_eqf(self.h, _dp(xi0), _dp(Xs), _ip(ids), _dp(q0), _dp(Q), len(ids), S.ctypes.data_as(c_double_p), time)

    def get_sigma(self):
        n = self.lib.orc_filter_sigma_dim(self.h)
        out = np.zeros((n, n), order="F")
        k = self.lib.orc_filter_get_sigma(self.h, out.ctypes.data_as(c_double_p), n * n)
        assert k == n * n
        return out

    def last_gamma(self):
        out = np.zeros(21 + 3 * self.cap)
        k = self.lib.orc_filter_last_gamma(self.h, _dp(out), len(out))
        return out[:k].copy()

    # ---- VIO_eqf methods
    def integrate_riccati_fast(self, imu13, dt):
        imu13 = _f64(imu13)
        self.lib.orc_eqf_integrate_riccati_fast(self.h, _dp(imu13), dt)

    def integrate_riccati_accurate(self, imu13, dt):
        imu13 = _f64(imu13)
        self.lib.orc_eqf_integrate_riccati_accurate(self.h, _dp(imu13), dt)

    def integrate_riccati_discrete(self, imu13, dt):
        imu13 = _f64(imu13)
        self.lib.orc_eqf_integrate_riccati_discrete(self.h, _dp(imu13), dt)

    def integrate_observer(self, imu13, dt, discrete=True):
        imu13 = _f64(imu13)
        self.lib.orc_eqf_integrate_observer(self.h, _dp(imu13), dt, int(discrete))

    def vision_update(self, cam, ids, y, stamp=0.0):
        ids, y = _i32(ids), _f64(y)
        self.lib.orc_eqf_vision_update(self.h, stamp, C.byref(cam), _ip(ids), _dp(y), len(ids))

    def remove_landmark_by_index(self, idx):
        self.lib.orc_eqf_remove_landmark_by_index(self.h, idx)

    def add_landmarks(self, ids, p, var):
        ids, p = _i32(ids), _f64(p)
        self.lib.orc_eqf_add_landmarks(self.h, _ip(ids), _dp(p), len(ids), var)

    def compute_nees(self, sensor, ids, p):
        sensor, ids, p = _f64(sensor), _i32(ids), _f64(p)
        return self.lib.orc_eqf_compute_nees(self.h, _dp(sensor), _ip(ids), _dp(p), len(ids))

    def outlier_stats(self, cam, ids, y):
        ids, y = _i32(ids), _f64(y)
        N = self.lib.orc_filter_sigma_dim(self.h)
        N = (N - 21) // 3
        a, p = np.zeros(N), np.zeros(N)
        self.lib.orc_filter_outlier_stats(self.h, C.byref(cam), _ip(ids), _dp(y), len(ids), _dp(a), _dp(p))
        return a, p

    def output_cov_all(self, cam):
        N = (self.lib.orc_filter_sigma_dim(self.h) - 21) // 3
        out = np.zeros(4 * N)
        self.lib.orc_filter_output_cov_all(self.h, C.byref(cam), _dp(out))
        return out.reshape(N, 2, 2)

    # ---- matrices
    def _n(self):
        return self.lib.orc_filter_sigma_dim(self.h)

    def state_matrix_A(self, imu13):
        imu13 = _f64(imu13)
        n = self._n()
        out = np.zeros((n, n), order="F")
        assert self.lib.orc_state_matrix_A(self.h, _dp(imu13), out.ctypes.data_as(c_double_p), n * n) == n * n
        return out

    def input_matrix_B(self):
        n = self._n()
        out = np.zeros((n, 12), order="F")
        assert self.lib.orc_input_matrix_B(self.h, out.ctypes.data_as(c_double_p), n * 12) == n * 12
        return out

    def output_matrix_C(self, cam, ids, y, use_equivariant=True):
        ids, y = _i32(ids), _f64(y)
        n, M = self._n(), len(ids)
        out = np.zeros((2 * M, n), order="F")
        assert self.lib.orc_output_matrix_C(self.h, C.byref(cam), _ip(ids), _dp(y), M, int(use_equivariant), out.ctypes.data_as(c_double_p), 2 * M * n) == 2 * M * n
        return out

    def state_chart(self, sensor, ids, p):
        sensor, ids, p = _f64(sensor), _i32(ids), _f64(p)
        n = 21 + 3 * len(ids)
        out = np.zeros(n)
        assert self.lib.orc_state_chart(self.h, _dp(sensor), _ip(ids), _dp(p), len(ids), _dp(out), n) == n
        return out

    def bench_frame(self, imu13_k, dts, stamp, cam, ids, y, mode, reps):
        imu13_k, dts, ids, y = _f64(imu13_k), _f64(dts), _i32(ids), _f64(y)
        return self.lib.orc_bench_frame(self.h, _dp(imu13_k), _dp(dts), len(dts), stamp, C.byref(cam), _ip(ids), _dp(y), len(ids), mode, reps)


def oracle_cam_project(cam, p):
    out = np.zeros(2)
    load_oracle().orc_cam_project(C.byref(cam), _dp(_f64(p)), _dp(out))
    return out


def oracle_cam_undistort(cam, y):
    out = np.zeros(3)
    load_oracle().orc_cam_undistort(C.byref(cam), _dp(_f64(y)), _dp(out))
    return out


def oracle_cam_jacobian(cam, p):
    out = np.zeros(6)
    load_oracle().orc_cam_jacobian(C.byref(cam), _dp(_f64(p)), _dp(out))
    return out.reshape(2, 3)


def se3_log_dist(a7, b7):
    lib = load_oracle()
    a7, b7 = _f64(a7), _f64(b7)
    return lib.orc_se3_log_dist(_dp(a7), _dp(b7))
