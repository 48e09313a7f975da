"""Duration of the propagation kernel against the number of observer steps riding in it (device-side trace, EQF_OPT_TRACE).
usage: python scripts/propagate_vs_steps.py [N]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from eqvio_amd.capi import EqfCore, OPT_TRACE
from util import CHARTS, default_camera, random_imu, random_spd, reasonable_state, settings_for, synth_measurement
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(1)
chart = CHARTS["invdepth"]
s = settings_for(chart, fastRiccati=1)
xi0, Xs, ids, q0, Q = reasonable_state(rng, N, shuffle_ids=False)
S0 = random_spd(rng, 21 + 3 * N)
cam = default_camera()
core = EqfCore(N, chart)
core.set_state(xi0, Xs, ids, q0, Q); core.set_sigma(S0)
core.set_option(OPT_TRACE, 1)
R = 1024
dev = np.zeros((R, 48), np.uint64); host = np.zeros((R, 8), np.int64); last = C.c_uint()
Qd, Pd = s.input_gain_diag12(), s.state_gain_diag8()
for k in (0, 1, 5, 10, 17, 24):
    durs = []
    for rep in range(12):
        imus = np.stack([random_imu(rng, stamp=0.005 * i) for i in range(max(k, 1))])[:k] if k else np.zeros((0, 13))
        dts = np.full(k, 0.005)
        mean = random_imu(rng)
        _, Xs1, _, _, Q1 = core.get_state()
        mid, y = synth_measurement(rng, cam, ids, q0, Q1, noise_px=0.5)
        core.propagate_fast(mean, 0.05, Qd, Pd, imus, dts, True)
        core.stats_then_update(cam, mid, y, 1e9, 1e9, 4.0, True, False)
        core.set_sigma(S0)  # keep the covariance well conditioned
        core.lib.eqf_trace_read(core.h, dev.ctypes.data_as(C.POINTER(C.c_ulonglong)), host.ctypes.data_as(C.POINTER(C.c_longlong)), C.byref(last))
        d = dev[last.value % R].astype(np.float64) * 0.01
        durs.append((d[1] - d[0] if d[0] > 0 else 0.0, d[2] - d[1], d[3] - d[2]))  # no assembly kernel with fused assembly (the default)
    a = np.median(np.array(durs[2:]), axis=0)
    print(f"N={N} k={k:2d} observer steps: assemble {a[0]:.2f} us, propagate {a[1]:.2f} us, build_Z {a[2]:.2f} us (start to next start)")
