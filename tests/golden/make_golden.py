"""Generate the golden fixtures under tests/golden/ from the CPU oracle (oracle/, the restated reference).

The reference holds no golden vectors of its own and cannot be built or imported here, so the
fixtures are produced by the oracle after it passed the restated reference property tests
(tests/test_oracle_properties.py). They pin (a) the oracle against regressions and (b) the HIP path on the GPU box
(where /root/reference does not exist).  Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle_binding import OracleFilter  # noqa: E402
from util import CAMERAS, CHARTS, euroc_camera, random_imu, random_spd, reasonable_state, settings_for, synth_measurement  # noqa: E402

CASES = [("euclid", 5, 1), ("invdepth", 5, 2), ("euclid", 20, 3), ("invdepth", 20, 4), ("invdepth", 50, 5)]


def make_case(chart_name, N, seed):
    rng = np.random.default_rng(seed)
    chart = CHARTS[chart_name]
    s = settings_for(chart, fastRiccati=1, useDiscreteInnovationLift=0, measurementNoise=1.5)
    xi0, Xs, ids, q0, Q = reasonable_state(rng, N, shuffle_ids=True)
    S0 = random_spd(rng, 21 + 3 * N)
    cam = euroc_camera()
    imu = random_imu(rng, bias_vel=True)
    k = 6
    imus = np.stack([random_imu(rng, stamp=0.005 * i) for i in range(k)])
    dts = rng.uniform(0.002, 0.006, k)
    orc = OracleFilter(s)
    orc.set_eqf(xi0, Xs, ids, q0, Q, S0)
    out = dict(chart=np.int32(chart), N=np.int32(N), xi0=xi0, Xs=Xs, ids=ids, q0=q0, Q=Q, Sigma0=S0, imu=imu, dt=np.float64(0.05), imus=imus, dts=dts,
               cam=np.array([cam.fx, cam.fy, cam.cx, cam.cy, cam.width, cam.height]), meas_var=np.float64(s.measurementNoise**2),
               Qdiag=s.input_gain_diag12(), Pdiag8=s.state_gain_diag8())
    A, B = orc.state_matrix_A(imu), orc.input_matrix_B()
    orc.integrate_riccati_fast(imu, 0.05)
    S1 = orc.get_sigma()
    for i in range(k):
        orc.integrate_observer(imus[i], dts[i], True)
    _, Xs1, _, _, Q1 = orc.get_eqf()
    mid, y = synth_measurement(rng, cam, ids, q0, Q1, noise_px=1.0)
    C = orc.output_matrix_C(cam, mid, y, True)
    orc.vision_update(cam, mid, y)
    S2 = orc.get_sigma()
    _, Xs2, _, _, Q2 = orc.get_eqf()
    gamma = orc.last_gamma()
    est_sensor, _, est_p = orc.state_estimate()
    out.update(meas_ids=mid, meas_y=y, Xs_after_observer=Xs1, Q_after_observer=Q1, Xs_after_update=Xs2, Q_after_update=Q2, Gamma=gamma,
               est_sensor=est_sensor, est_p=est_p)
    if N <= 20:
        out.update(A=A, B=B, C=C, Sigma_propagated=S1, Sigma_updated=S2)
    else:  # keep the fixture small: diagonal, first 21 rows and Frobenius norms
        out.update(Sigma_propagated_diag=np.diag(S1).copy(), Sigma_propagated_top=S1[:21].copy(), Sigma_propagated_fro=np.float64(np.linalg.norm(S1)),
                   Sigma_updated_diag=np.diag(S2).copy(), Sigma_updated_top=S2[:21].copy(), Sigma_updated_fro=np.float64(np.linalg.norm(S2)),
                   A_fro=np.float64(np.linalg.norm(A)), B_fro=np.float64(np.linalg.norm(B)), C_fro=np.float64(np.linalg.norm(C)))
    return out


# second family: the non-default branches of the same path (accurate Riccati, continuous observer lift, discrete
# innovation lift, distorted cameras, outlier statistics, NEES)
VARIANTS = [("euclid", 8, "radtan", 11), ("invdepth", 8, "equidistant", 12), ("invdepth", 30, "radtan", 13)]


def cam_vector(cam):
    return np.array([cam.model, cam.width, cam.height, cam.fx, cam.fy, cam.cx, cam.cy] + list(cam.dist))


def make_variant(chart_name, N, cam_name, seed):
    rng = np.random.default_rng(seed)
    chart = CHARTS[chart_name]
    s = settings_for(chart, fastRiccati=0, useDiscreteInnovationLift=1, useDiscreteVelocityLift=0, measurementNoise=2.0)
    xi0, Xs, ids, q0, Q = reasonable_state(rng, N, shuffle_ids=True)
    S0 = random_spd(rng, 21 + 3 * N)
    cam = CAMERAS[cam_name]()
    imu = random_imu(rng, bias_vel=True)
    k = 4
    imus = np.stack([random_imu(rng, stamp=0.005 * i) for i in range(k)])
    dts = rng.uniform(0.002, 0.006, k)
    orc = OracleFilter(s)
    orc.set_eqf(xi0, Xs, ids, q0, Q, S0)
    out = dict(chart=np.int32(chart), N=np.int32(N), xi0=xi0, Xs=Xs, ids=ids, q0=q0, Q=Q, Sigma0=S0, imu=imu, dt=np.float64(0.02), imus=imus, dts=dts,
               cam=cam_vector(cam), meas_var=np.float64(s.measurementNoise**2), Qdiag=s.input_gain_diag12(), Pdiag8=s.state_gain_diag8())
    orc.integrate_riccati_accurate(imu, 0.02)
    S1 = orc.get_sigma()
    for i in range(k):
        orc.integrate_observer(imus[i], dts[i], False)
    _, Xs1, _, _, Q1 = orc.get_eqf()
    sub = np.sort(rng.permutation(N)[: N - 2])  # two landmarks unobserved
    mid, y = synth_measurement(rng, cam, ids, q0, Q1, noise_px=1.5, subset=sub)
    absE, probE = orc.outlier_stats(cam, mid, y)
    orc.vision_update(cam, mid, y)
    S2 = orc.get_sigma()
    _, Xs2, _, _, Q2 = orc.get_eqf()
    est_sensor, est_ids, est_p = orc.state_estimate()
    truth_sensor = est_sensor.copy()
    truth_sensor[0:6] += rng.normal(size=6) * 1e-3
    truth_sensor[13:16] += rng.normal(size=3) * 1e-2
    truth_p = est_p + rng.normal(size=est_p.shape) * 1e-2
    nees = orc.compute_nees(truth_sensor, est_ids, truth_p)
    out.update(meas_ids=mid, meas_y=y, Sigma_propagated=S1, Xs_after_observer=Xs1, Q_after_observer=Q1, absErr=absE, probErr=probE, Sigma_updated=S2,
               Xs_after_update=Xs2, Q_after_update=Q2, Gamma=orc.last_gamma(), truth_sensor=truth_sensor, truth_ids=est_ids, truth_p=truth_p,
               nees=np.float64(nees))
    return out


def write(path, maker, *a):
    if os.path.exists(path) and "--force" not in sys.argv:  # the zip container carries timestamps: do not churn the history
        print(path, "exists (use --force to regenerate)")
        return
    np.savez_compressed(path, **maker(*a))
    print(path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    for chart_name, N, cam_name, seed in VARIANTS:
        write(os.path.join(HERE, f"variant_{chart_name}_N{N}_{cam_name}.npz"), make_variant, chart_name, N, cam_name, seed)
    for chart_name, N, seed in CASES:
        write(os.path.join(HERE, f"frame_{chart_name}_N{N}.npz"), make_case, chart_name, N, seed)
