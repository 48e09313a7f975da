"""Second, independent generator of the golden outputs: for the INPUTS held by every tests/golden/frame_*.npz and
variant_*.npz it recomputes the outputs with oracle/indep/eqvio_ref.py (numpy float64; written from /root/reference in
another language and another representation than oracle/*.hpp) and stores them as tests/golden/indep_<name>.npz.

tests/test_golden.py then checks, on CPU and without either generator, that the two fixture families agree to 1e-12
(so the committed golden values are pinned by two independently written restatements of the reference), and on the GPU
that the HIP path reproduces the independent family as well.

Run (build container):  python tests/golden/make_golden_indep.py [--force]
"""
import glob
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle", "indep"))

from eqvio_ref import Camera, EqVIORef, F64  # noqa: E402

KIND = {0: "euclid", 1: "invdepth", 2: "normal"}
GEN = "oracle/indep/eqvio_ref.py (numpy float64), tests/golden/make_golden_indep.py"


def frame_outputs(d):
    r = EqVIORef(F64())
    kind = KIND[int(d["chart"])]
    N = int(d["N"])
    X = r.group_from_flat(d["Xs"], d["ids"], d["Q"])
    xi0 = r.state_from_flat(d["xi0"], d["ids"], d["q0"])
    imu = r.imu_from_flat(d["imu"])
    cam = Camera(0, d["cam"][0], d["cam"][1], d["cam"][2], d["cam"][3])
    Qin, P = r.diag(d["Qdiag"]), r.state_gain(d["Pdiag8"], N)
    A = r.state_matrix_A(kind, X, xi0, imu)
    B = r.input_matrix_B(kind, X, xi0)
    S1 = r.riccati_fast(kind, X, xi0, d["Sigma0"], imu, float(d["dt"]), Qin, P)
    for i in range(len(d["dts"])):
        X = r.integrate_observer(X, xi0, r.imu_from_flat(d["imus"][i]), float(d["dts"][i]), True)
    Xs1, Q1 = r.group_to_flat(X)
    meas = r.meas_from_flat(d["meas_ids"], d["meas_y"])
    C = r.output_matrix_C(kind, xi0, X, cam, meas, True)
    X2, S2, g = r.vision_update(kind, X, xi0, S1, cam, meas, float(d["meas_var"]), True, False)
    Xs2, Q2 = r.group_to_flat(X2)
    est = r.state_action(X2, xi0)
    out = dict(generator=np.array(GEN), Xs_after_observer=Xs1, Q_after_observer=Q1, Xs_after_update=Xs2, Q_after_update=Q2, Gamma=g, est_p=np.asarray(est.p, float),
               est_sensor=r.sensor_to_flat(est.sensor))
    if N <= 20:
        out.update(A=A, B=B, C=C, Sigma_propagated=S1, Sigma_updated=S2)
    else:
        out.update(Sigma_propagated_diag=np.diag(S1).copy(), Sigma_propagated_top=S1[:21].copy(), Sigma_propagated_fro=np.float64(np.linalg.norm(S1)),
                   Sigma_updated_diag=np.diag(S2).copy(), Sigma_updated_top=S2[:21].copy(), Sigma_updated_fro=np.float64(np.linalg.norm(S2)),
                   A_fro=np.float64(np.linalg.norm(A)), B_fro=np.float64(np.linalg.norm(B)), C_fro=np.float64(np.linalg.norm(C)))
    return out


def variant_outputs(d):
    r = EqVIORef(F64())
    kind = KIND[int(d["chart"])]
    N = int(d["N"])
    X = r.group_from_flat(d["Xs"], d["ids"], d["Q"])
    xi0 = r.state_from_flat(d["xi0"], d["ids"], d["q0"])
    c = d["cam"]
    cam = Camera(int(c[0]), c[3], c[4], c[5], c[6], list(c[7:12]))
    Qin, P = r.diag(d["Qdiag"]), r.state_gain(d["Pdiag8"], N)
    S1 = r.riccati_accurate(kind, X, xi0, d["Sigma0"], r.imu_from_flat(d["imu"]), float(d["dt"]), Qin, P)
    for i in range(len(d["dts"])):
        X = r.integrate_observer(X, xi0, r.imu_from_flat(d["imus"][i]), float(d["dts"][i]), False)
    Xs1, Q1 = r.group_to_flat(X)
    meas = r.meas_from_flat(d["meas_ids"], d["meas_y"])
    a, p = r.outlier_stats(kind, X, xi0, S1, cam, meas)  # per measurement, ascending id
    X2, S2, g = r.vision_update(kind, X, xi0, S1, cam, meas, float(d["meas_var"]), True, True)
    Xs2, Q2 = r.group_to_flat(X2)
    nees = r.compute_nees(kind, X2, xi0, S2, r.sensor_from_flat(d["truth_sensor"]), d["truth_ids"], np.asarray(d["truth_p"], float))
    return dict(generator=np.array(GEN), Sigma_propagated=S1, Xs_after_observer=Xs1, Q_after_observer=Q1, absErr_by_meas=np.asarray(a, float), probErr_by_meas=np.asarray(p, float),
                Sigma_updated=S2, Gamma=g, Xs_after_update=Xs2, Q_after_update=Q2, nees=np.float64(nees))


if __name__ == "__main__":
    for pattern, fn in (("frame_*.npz", frame_outputs), ("variant_*.npz", variant_outputs)):
        for path in sorted(glob.glob(os.path.join(HERE, pattern))):
            out_path = os.path.join(HERE, "indep_" + os.path.basename(path))
            if os.path.exists(out_path) and "--force" not in sys.argv:
                print(out_path, "exists (use --force to regenerate)")
                continue
            np.savez_compressed(out_path, **fn(dict(np.load(path))))
            print(out_path, os.path.getsize(out_path) // 1024, "KiB")
