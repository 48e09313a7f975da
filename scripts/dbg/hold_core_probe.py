"""core-level: the filter's frame (stage -> propagate -> [append] -> stats_then_update) with held / ordinary new landmarks, with and without the float model"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from eqvio_amd.capi import EqfCore, OPT_SIGMA_FP32, OPT_MEASURE_IN_PROPAGATE
from test_gpu_parity import make_pair
from util import CHARTS, default_camera, random_imu, synth_measurement
for mode in (0, 1):
  for mip in (1, 0):
    N, knew = 195, 5
    rng, settings, orc, core, (xi0, Xs, ids, q0, Q, S) = make_pair(CHARTS["invdepth"], N, seed=3, cap=N + knew + 8, useDiscreteInnovationLift=0)
    o = np.argsort(ids); ids, q0, Q = ids[o], q0[o], Q[o]   # ascending, like a tracker's
    # (Sigma stays as it is: a random SPD matrix does not care)
    cam = default_camera()
    imus = [random_imu(rng) for _ in range(17)]; dts = [0.002] * 17; mean = np.mean(imus, axis=0)
    new_ids = (np.arange(knew) + int(ids.max()) + 1).astype(np.int32)
    new_p = rng.uniform(-0.5, 0.5, (knew, 3)) + np.array([0, 0, 5.0])
    Qg, Pg = settings.input_gain_diag12(), settings.state_gain_diag8()
    all_ids = np.concatenate([ids, new_ids]); all_q0 = np.vstack([q0, new_p]); all_Q = np.vstack([Q, np.tile([1, 0, 0, 0, 1.0], (knew, 1))])
    sub = np.array([i for i in range(len(all_ids)) if i not in (18, 23, 35, 52, 63)])
    mid, y = synth_measurement(rng, cam, all_ids, all_q0, all_Q, noise_px=0.3, subset=sub)
    outs = []
    for held in (0, 1):
        c = EqfCore(N + knew + 8, CHARTS["invdepth"])
        c.set_state(xi0, Xs, ids, q0, Q); c.set_sigma(S)
        c.set_option(OPT_SIGMA_FP32, mode); c.set_option(OPT_MEASURE_IN_PROPAGATE, mip)
        if held: assert c.add_landmarks_held(new_ids, new_p, 1.7)
        c.stage_measurement(mid, y)
        c.propagate_fast(mean, sum(dts), Qg, Pg, imus, dts, True)
        Sp = c.get_sigma() if False else None
        if not held: c.add_landmarks(new_ids, new_p, 1.7)
        upd, a, p, d = c.stats_then_update(cam, mid, y, 1e9, 1e9, 1.0, True, False)
        outs.append((c.get_sigma(), upd))
    idx = np.argwhere(np.abs(outs[0][0] - outs[1][0]) > 0)
    print("   differing entries:", [(int(a), int(b)) for a, b in idx[:12]])
    d = np.abs(outs[0][0] - outs[1][0])
    print("mode", mode, "measure-in-propagate", mip, "updated", outs[0][1], outs[1][1], "max diff", d.max(), "entries", int((d > 0).sum()))
