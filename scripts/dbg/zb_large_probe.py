"""Where do Z-inside-the-look-ahead-kernel (17 .. 32 panels) and the k_build_Z route differ? Prints max differences of state, Sigma+ and W by 32 x 32 tile."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from eqvio_amd.capi import COORD_INVDEPTH, OPT_Z_IN_LOOKAHEAD, OPT_LOOKAHEAD, EqfCore
from util import default_camera, random_spd, reasonable_state, settings_for, synth_measurement
N = int(sys.argv[1]) if len(sys.argv) > 1 else 272
rng = np.random.default_rng(1000 + N)
xi0, Xs, ids, q0, Q = reasonable_state(rng, N)
S0 = random_spd(rng, 21 + 3 * N)
cam = default_camera()
mid, y = synth_measurement(rng, cam, ids, q0, Q, noise_px=0.5)
outs = {}
for name, la, zb in (("chain", 0, 2), ("buildz", 1, 2), ("inside", 1, 1)):
    core = EqfCore(N, COORD_INVDEPTH)
    core.set_option(OPT_LOOKAHEAD, la)
    core.set_option(OPT_Z_IN_LOOKAHEAD, zb)
    core.set_state(xi0, Xs, ids, q0, Q)
    core.set_sigma(S0)
    core.vision_update(cam, mid, y, 4.0, True, True)
    m, n = 2 * len(mid), 21 + 3 * N
    outs[name] = (core.get_state(), core.get_sigma(), core.debug_get_W(m + n + 1, m))
    core.close()
for a, b in (("chain", "buildz"), ("buildz", "inside"), ("chain", "inside")):
    (sa, Sa, Wa), (sb, Sb, Wb) = outs[a], outs[b]
    print(a, "vs", b, ": state", [float(np.abs(np.asarray(x, float) - np.asarray(z, float)).max()) for x, z in zip(sa, sb)], "Sigma", np.abs(Sa - Sb).max(), "of", np.abs(Sa).max(), "W", np.abs(Wa - Wb).max(), "of", np.abs(Wa).max())
    D = np.abs(Wa - Wb)
    bad = np.argwhere(D > 0)
    if len(bad):
        print("   W differs in", len(bad), "entries; rows", bad[:, 0].min(), "..", bad[:, 0].max(), "cols", bad[:, 1].min(), "..", bad[:, 1].max(), "first:", bad[:5].tolist())
        rt, ct = bad[:, 0] // 16, bad[:, 1] // 32
        import collections
        print("   (half-row, panel) with differences:", sorted(collections.Counter(zip(rt.tolist(), ct.tolist())).items())[:40])
