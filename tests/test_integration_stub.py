"""INTEGRATION.md §A, compiled and run. tests/integration/VIO_eqf_mi355x.cpp is the file a maintainer adds to the reference tree
(it replaces the member bodies of src/mathematical/VIO_eqf.cpp with calls into include/eqf_hip.h); tests/integration/standin/ holds
test scaffolding shaped like the Eigen / LiePP / GIFT / eqvio types it touches (none of those libraries is in this image);
tests/integration/run_one_frame.cpp is reference-style caller code: aggregate initialisation (test/test_FilterStatistics.cpp:40),
copies of a filter, a frame of propagation + update, and the writers' direct reads of Sigma (src/VIOWriter.cpp:171-222).

CPU: the binding compiles without warnings and links against libeqf_hip.so. GPU: the driver runs and every number it prints is
compared with the oracle driven through the same sequence of VIO_eqf members."""
import os
import subprocess

import numpy as np
import pytest

from eqvio_amd.capi import COORD_EUCLIDEAN, COORD_INVDEPTH
from oracle_binding import OracleFilter, se3_log_dist
from util import euroc_camera, random_imu, random_spd, reasonable_state, rel_fro, settings_for, synth_measurement

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "integration")
LIBDIR = os.path.join(ROOT, "eqvio_amd", "lib")


def build(tmp_path):
    exe = str(tmp_path / "run_one_frame")
    flags = ["-std=c++20", "-O1", "-Wall", "-Wextra", "-Werror", "-Wno-missing-field-initializers", "-I", os.path.join(SRC, "standin"), "-I", os.path.join(ROOT, "include")]
    objs = []
    for name in ("VIO_eqf_mi355x", "run_one_frame"):
        obj = str(tmp_path / (name + ".o"))
        subprocess.run(["g++", *flags, "-c", os.path.join(SRC, name + ".cpp"), "-o", obj], check=True)
        objs.append(obj)
    subprocess.run(["g++", *objs, "-L", LIBDIR, "-leqf_hip", "-Wl,-rpath," + LIBDIR, "-o", exe], check=True)
    return exe


def test_binding_compiles_and_links(tmp_path):
    exe = build(tmp_path)
    syms = subprocess.run(["nm", "-C", "--undefined-only", exe], check=True, capture_output=True, text=True).stdout
    used = {ln.split()[-1] for ln in syms.splitlines() if " eqf_" in ln}
    # every member the reference declares forwards to the ABI: these are the entry points the binding pulls from libeqf_hip.so (integrateRiccatiStateFast
    # through eqf_propagate_fast since round 5: it is issued together with the observer steps that follow it)
    assert {"eqf_create", "eqf_destroy", "eqf_set_state", "eqf_get_state", "eqf_set_sigma", "eqf_get_sigma", "eqf_get_sigma_block", "eqf_integrate_observer",
            "eqf_propagate_fast", "eqf_integrate_riccati_accurate", "eqf_integrate_riccati_discrete", "eqf_vision_update", "eqf_state_estimate",
            "eqf_compute_nees", "eqf_add_landmarks", "eqf_remove_landmarks", "eqf_remove_invalid_landmarks", "eqf_get_ids", "eqf_output_cov_all"} <= used, used


def write_records(path, rec):
    with open(path, "w") as f:
        for k, v in rec.items():
            v = np.asarray(v, float).reshape(-1, order="F") if np.ndim(v) == 2 and k == "Sigma0" else np.asarray(v, float).reshape(-1)
            f.write(k + " " + str(len(v)) + " " + " ".join(repr(float(x)) for x in v) + "\n")


def read_records(path):
    out = {}
    for ln in open(path):
        t = ln.split()
        out[t[0]] = np.array([float(x) for x in t[2:]])
        assert len(out[t[0]]) == int(t[1])
    return out


def propagate(orc, imus, dts):
    for imu, dt in zip(imus, dts):
        orc.integrate_riccati_fast(imu, dt)
        orc.integrate_observer(imu, dt, True)


@pytest.mark.gpu
@pytest.mark.parametrize("chart", [COORD_EUCLIDEAN, COORD_INVDEPTH])
def test_one_frame_through_the_binding(tmp_path, chart):
    exe = build(tmp_path)
    rng = np.random.default_rng(77 + chart)
    N, k = 12, 4
    settings = settings_for(chart, useEquivariantOutput=1, useDiscreteInnovationLift=0, measurementNoise=1.5)
    xi0, Xs, ids, q0, Q = reasonable_state(rng, N, shuffle_ids=True)
    S0 = random_spd(rng, 21 + 3 * N)
    cam = euroc_camera()
    imus = [random_imu(rng, stamp=0.005 * s, bias_vel=True) for s in range(k)]
    dts = [0.005] * k
    truth_sensor, _, _, truth_p, _ = reasonable_state(np.random.default_rng(5), N)
    new_p, new_var = np.array([0.3, -0.2, 4.0]), 2.5

    def fresh_oracle():
        o = OracleFilter(settings)
        o.set_eqf(xi0, Xs, ids, q0, Q, S0)
        propagate(o, imus, dts)
        return o

    # the measurement is generated at the propagated estimate
    orc = fresh_oracle()
    _, _, _, q0_1, Q_1 = orc.get_eqf()
    mid, y = synth_measurement(rng, cam, ids, q0_1, Q_1, noise_px=1.0)

    write_records(tmp_path / "in.txt", dict(chart=[chart], xi0=xi0, Xs=Xs, ids=ids, p=q0, Q=Q, Sigma0=S0, imus=np.array(imus), dts=dts,
                                            Qdiag12=settings.input_gain_diag12(), Pdiag8=settings.state_gain_diag8(),
                                            cam=[cam.fx, cam.fy, cam.cx, cam.cy, cam.width, cam.height], meas_ids=mid, meas_y=y,
                                            meas_var=[settings.measurementNoise**2], truth_sensor=truth_sensor, truth_p=truth_p, new_p=new_p, new_var=[new_var]))
    subprocess.run([exe, str(tmp_path / "in.txt"), str(tmp_path / "out.txt")], check=True, timeout=300)
    out = read_records(tmp_path / "out.txt")

    def same_state(tag, o):
        s_o, ids_o, p_o = o.state_estimate()
        s_g, p_g = out[tag + "_est_sensor"], out[tag + "_est_p"].reshape(-1, 3)
        assert np.array_equal(out[tag + "_est_ids"].astype(int), ids_o)
        assert se3_log_dist(s_g[6:13], s_o[6:13]) <= 1e-9 * max(1.0, np.linalg.norm(s_o[10:13]))
        assert se3_log_dist(s_g[16:23], s_o[16:23]) <= 1e-9
        assert np.max(np.abs(s_g[0:6] - s_o[0:6])) <= 1e-9 and np.max(np.abs(s_g[13:16] - s_o[13:16])) <= 1e-9 * max(1.0, np.max(np.abs(s_o[13:16])))
        assert np.max(np.linalg.norm(p_g - p_o, axis=1) / np.maximum(1.0, np.linalg.norm(p_o, axis=1))) <= 1e-9

    def same_readers(tag, S):
        n = S.shape[0]
        Sg = out[tag + "_Sigma"].reshape(n, n, order="F")
        assert rel_fro(Sg, S) <= 1e-9
        # the writers' expressions on the host mirror are views of that same matrix
        assert np.array_equal(out[tag + "_poseCov"].reshape(6, 6, order="F"), Sg[6:12, 6:12]) and np.array_equal(out[tag + "_attCov"].reshape(3, 3, order="F"), Sg[6:9, 6:9])
        d = np.diag(Sg)
        assert np.array_equal(out[tag + "_sigmaPose"], d[6:12]) and np.array_equal(out[tag + "_sigmaCamera"], d[15:21]) and np.array_equal(out[tag + "_sigmaBias"], d[0:6])
        return Sg

    # (a) output covariance of one landmark before the update, then the update
    probe = int(mid[0])
    col = 21 + 3 * int(np.where(ids == probe)[0][0])
    C0 = orc.output_matrix_C(cam, [probe], y[:2], use_equivariant=False)[:, col:col + 3]
    S_prop = orc.get_sigma()
    np.testing.assert_allclose(out["a_outputCov"].reshape(2, 2, order="F"), C0 @ S_prop[col:col + 3, col:col + 3] @ C0.T, rtol=1e-9, atol=0)
    orc.vision_update(cam, mid, y)
    same_state("a", orc)
    S_a = orc.get_sigma()
    Sg_a = same_readers("a", S_a)
    assert np.array_equal(out["a_Xid"].astype(int), ids)
    np.testing.assert_allclose(out["a_landmarkCov"].reshape(3, 3, order="F"), S_a[col:col + 3, col:col + 3], rtol=1e-8, atol=1e-12)
    nees_o = orc.compute_nees(truth_sensor, ids, truth_p)
    assert abs(out["a_nees"][0] - nees_o) <= 1e-7 * abs(nees_o)  # conditioning of Sigma enters here (tests/test_gpu_parity.py NEES test)

    # (b) the fork (a copy taken while the device was ahead) went its own way
    orc_b = fresh_oracle()
    orc_b.remove_landmark_by_index(1)
    orc_b.add_landmarks([100000], new_p, new_var)
    keep = mid != ids[1]
    orc_b.vision_update(cam, mid[keep], y.reshape(-1, 2)[keep].reshape(-1))
    same_state("b", orc_b)
    same_readers("b", orc_b.get_sigma())
    assert np.array_equal(out["b_Xid"].astype(int), np.concatenate([np.delete(ids, 1), [100000]]))

    # (c) a copy with current host members, edited on the host (Sigma(0,0) doubled), then one landmark removed
    S_c = Sg_a.copy()
    S_c[0, 0] *= 2.0
    np.testing.assert_array_equal(out["c_landmarkCov"].reshape(3, 3, order="F"), S_c[col:col + 3, col:col + 3])
    S_c = np.delete(np.delete(S_c, slice(21, 24), 0), slice(21, 24), 1)
    assert np.array_equal(out["c_Sigma"].reshape(S_c.shape, order="F"), S_c)  # an upload, a compaction and a download move bits, nothing else
    assert np.array_equal(out["c_Xid"].astype(int), ids[1:])
