"""Which fp64 implementation is closest to the TRUE answer where the 1e-9 bar against the oracle was missed (round 1: config 1
with the reference's template values, 9e-9)?  tests/golden/truth_template_chain.npz holds a 10-frame free-running chain with
the template noise values (point variance 5000, pixel noise 0.003: cond(Sigma) ~ 2e10..2e12) evaluated at 50 digits by the
independent restatement (tests/golden/make_truth_mp.py).  Findings pinned here:

 * the reference's arithmetic AS WRITTEN (LU inverse, Sigma - K C Sigma; VIO_eqf.cpp:116-131) loses the symmetry of Sigma
   (3e-8 relative after three frames) and sits 1e-7 from the truth in fp64 - in the C++ oracle and in the numpy restatement alike;
 * a symmetric evaluation (oracle "efficient": Cholesky, Sigma - K T^T) stays within 2e-9 of the truth;
 * the HIP path (exactly symmetric Sigma, LDL^T chain, Sigma - W W^T) must be at least as close to the truth as the
   as-written fp64 evaluation on every frame of the chain and within 5e-9 of it (-m gpu).
So on this configuration 1e-9 against the as-written oracle is unreachable for ANY fp64 implementation, including a second
evaluation of the as-written formulas themselves; the distance to the truth is the meaningful figure."""
import os
import sys

import numpy as np
import pytest

from eqvio_amd.capi import Camera
from util import rel_fro, settings_for

HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(HERE, "golden", "truth_template_chain.npz")


def _load():
    d = dict(np.load(PATH))
    s = settings_for(0, fastRiccati=1, useDiscreteInnovationLift=1, useDiscreteVelocityLift=1, measurementNoise=float(np.sqrt(d["meas_var"])))
    (s.biasOmegaProcessVariance, s.biasAccelProcessVariance, s.attitudeProcessVariance, s.positionProcessVariance, s.velocityProcessVariance,
     s.cameraAttitudeProcessVariance, s.cameraPositionProcessVariance, s.pointProcessVariance) = [float(x) for x in d["proc8"]]
    q = np.sqrt(d["qin12"])
    s.velGyrNoise, s.velAccNoise, s.velGyrBiasWalk, s.velAccBiasWalk = float(q[0]), float(q[3]), float(q[6]), float(q[9])
    assert np.allclose(s.state_gain_diag8(), d["proc8"], rtol=1e-15) and np.allclose(s.input_gain_diag12(), d["qin12"], rtol=1e-14)
    cam = Camera.pinhole(*[float(x) for x in d["cam"]], 752, 480)
    return d, s, cam


def _oracle_chain(d, s, cam, mode):
    from oracle_binding import OracleFilter

    o = OracleFilter(s)
    o.set_arithmetic(mode)
    o.set_eqf(d["xi0"], d["Xs"], d["ids"], d["q0"], d["Q"], d["Sigma0"])
    err, asym = [], []
    for f in range(len(d["imus"])):
        o.integrate_riccati_fast(d["imus"][f], float(d["dt"]))
        for k in range(d["obs_imus"].shape[1]):
            o.integrate_observer(d["obs_imus"][f, k], float(d["obs_dt"]), True)
        o.vision_update(cam, d["truth_meas_ids"], d["truth_meas_y"][f])
        S = o.get_sigma()
        err.append(rel_fro(S, d["truth_Sigma"][f]))
        asym.append(np.linalg.norm(S - S.T) / np.linalg.norm(S))
    return np.array(err), np.array(asym)


def test_truth_fixture_provenance():
    d = np.load(PATH)
    assert "mpmath 50 digits" in str(d["generator"]) and "oracle/indep/eqvio_ref.py" in str(d["generator"])
    T = d["truth_Sigma"]
    assert np.array_equal(T, np.transpose(T, (0, 2, 1)))  # the exact answer is symmetric
    assert np.linalg.cond(T[-1]) > 1e10


def test_as_written_fp64_is_1e7_from_truth_and_symmetric_evaluation_is_not():
    from oracle_binding import ARITH_AS_WRITTEN, ARITH_EFFICIENT

    d, s, cam = _load()
    e_asw, a_asw = _oracle_chain(d, s, cam, ARITH_AS_WRITTEN)
    e_eff, a_eff = _oracle_chain(d, s, cam, ARITH_EFFICIENT)
    print("as written: err", e_asw, "asym", a_asw)
    print("efficient : err", e_eff, "asym", a_eff)
    assert e_asw[0] < 1e-14 and e_eff[0] < 1e-14  # well-conditioned first frame: everything agrees to rounding
    assert e_asw.max() > 2e-8 and a_asw.max() > 5e-9  # the reference's own arithmetic cannot hold 1e-9 here
    assert e_eff.max() < 5e-9 and a_eff.max() < 1e-11
    # a second, independent fp64 evaluation of the as-written formulas (numpy, LAPACK LU) is just as far from the truth
    sys.path.insert(0, os.path.join(HERE, "golden"))
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle", "indep"))
    from eqvio_ref import F64, EqVIORef
    from make_truth_mp import run_chain

    r = run_chain(EqVIORef(F64()), d, d["truth_meas_y"])
    e_np = np.array([rel_fro(r["Sigma"][f], d["truth_Sigma"][f]) for f in range(len(e_asw))])
    print("numpy f64 as written: err", e_np)
    assert e_np[0] < 1e-14 and e_np.max() > 2e-8


@pytest.mark.gpu
def test_device_is_closer_to_truth_than_the_as_written_fp64_reference():
    from eqvio_amd.capi import EqfCore
    from oracle_binding import ARITH_AS_WRITTEN

    d, s, cam = _load()
    e_asw, _ = _oracle_chain(d, s, cam, ARITH_AS_WRITTEN)
    N = len(d["ids"])
    core = EqfCore(N, 0)
    core.set_state(d["xi0"], d["Xs"], d["ids"], d["q0"], d["Q"])
    core.set_sigma(d["Sigma0"])
    e_dev, e_g = [], []
    for f in range(len(d["imus"])):
        core.integrate_riccati_fast(d["imus"][f], float(d["dt"]), d["qin12"], d["proc8"])
        core.integrate_observer(d["obs_imus"][f], np.full(d["obs_imus"].shape[1], float(d["obs_dt"])), True)
        core.vision_update(cam, d["truth_meas_ids"], d["truth_meas_y"][f], float(d["meas_var"]), True, True)
        S = core.get_sigma()
        assert np.array_equal(S, S.T)
        e_dev.append(rel_fro(S, d["truth_Sigma"][f]))
        e_g.append(rel_fro(core.last_gamma(), d["truth_Gamma"][f]))
    e_dev = np.array(e_dev)
    print("device err vs truth", e_dev, "\nas-written fp64 err vs truth", e_asw, "\ndevice Gamma err", np.array(e_g))
    assert e_dev.max() < 5e-9
    # frame 0 is well conditioned: both at rounding level; frame 1 (cond(S) ~ 1e13 for the first time) leaves every fp64 evaluation ~2.2e-9 from the truth, whichever
    # way the last bits fall (round 6, fused multiply-adds of the S / T entries written out: device 2.24e-9, as written 2.22e-9) - from then on the as-written arithmetic
    # drifts to 1e-7 and the device does not
    assert np.all(e_dev[1:] <= np.maximum(e_asw[1:], 3e-9))
    assert e_asw[2:].min() > 3 * e_dev[2:].max()
    assert e_dev[0] < 1e-13
    _, Xs, _, _, Q = core.get_state()
    assert np.max(np.abs(Xs - d["truth_Xs"][-1])) < 1e-6  # the state follows Gamma (errors 1e-8..1e-7 relative in every fp64 evaluation)
