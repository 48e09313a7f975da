"""mode-1 float model with / without held landmarks, and against the real float store: which frame, which entries differ"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from eqvio_amd.capi import OPT_SIGMA_FP32, OPT_HOLD_NEW_LANDMARKS, OPT_GATHER_IN_PROPAGATE, SimSettings, SimulationDataServer, VIOFilter
from test_gpu_fp32_sigma import uzh_like_settings
fs = uzh_like_settings()
mf = 200
sim = SimSettings.defaults(duration=1.0, trajectory="sine", numPoints=12000, wallDistance=3.0, numWalls=6, randomSeed=5, maxFeatures=mf, imuFreq=500.0, imageFreq=30.0, outputNoise=1)
srv = SimulationDataServer(sim, fs)
fs.cameraOffset[:] = srv.camera_offset()
s0, ids0, p0 = srv.true_state(0.0, True)
def mk(mode, hold, gather=1):
    f = VIOFilter(fs, max_landmarks=2 * mf + 64, sensor=s0, ids=ids0[:0], p=p0[:0], time=0.0)
    f.set_core_option(OPT_SIGMA_FP32, mode); f.set_core_option(OPT_HOLD_NEW_LANDMARKS, hold); f.set_core_option(OPT_GATHER_IN_PROPAGATE, gather)
    return f
fl = {"m1 hold": mk(1, 1), "m1 nohold": mk(1, 0), "m1 nohold nogather": mk(1, 0, 0), "m2": mk(2, 0), "m0 hold": mk(0, 1), "m0 nohold": mk(0, 0)}
frames = 0
while srv.next_measurement_type() != srv.NONE:
    if srv.next_measurement_type() == srv.IMU:
        imu = srv.get_imu()
        for f in fl.values(): f.process_imu(imu)
        continue
    stamp, ids, y = srv.get_vision()
    for f in fl.values(): f.process_vision(stamp, srv.cam, ids, y)
    frames += 1
    S = {k: f.get_sigma() for k, f in fl.items()}
    def cmp(a, b):
        if S[a].shape != S[b].shape: return "shape %s %s" % (S[a].shape, S[b].shape)
        d = np.abs(S[a] - S[b]); 
        if d.max() == 0: return "equal"
        i, j = np.unravel_index(np.argmax(d), d.shape); return "max %.3e at (%d, %d) n=%d, differing entries %d" % (d.max(), i, j, S[a].shape[0], (d > 0).sum())
    print(frames, "| m1 hold vs m1 nohold:", cmp("m1 hold", "m1 nohold"), "| m1 nohold vs m2:", cmp("m1 nohold", "m2"), "| nogather vs m2:", cmp("m1 nohold nogather", "m2"), "| m0 hold vs nohold:", cmp("m0 hold", "m0 nohold"))
    if frames == 2:
        d = np.abs(S["m1 hold"] - S["m1 nohold"]); idx = np.argwhere(d > 0)
        print("   entries:", [(int(a), int(b), float(d[a, b])) for a, b in idx[:30]], "N =", (S["m1 hold"].shape[0] - 21) // 3)
        ea, eb = fl["m1 hold"].state_estimate(), fl["m1 nohold"].state_estimate()
        print("   ids equal", np.array_equal(ea[1], eb[1]), "sensor diff", np.abs(np.asarray(ea[0]) - np.asarray(eb[0])).max(), "landmark diff", np.abs(ea[2] - eb[2]).max())
        import ctypes as C
        from eqvio_amd.capi import load_eqf_lib
        lib = load_eqf_lib()
        for k in ("m1 hold", "m1 nohold", "m0 hold", "m0 nohold"):
            h = fl[k].core_handle()
            c0, q0_, x0 = C.c_long(), C.c_long(), C.c_long(); lib.eqf_speculation_stats(h, C.byref(c0), C.byref(q0_), C.byref(x0), 0)
            f0, g0 = C.c_long(), C.c_long(); lib.eqf_selection_stats(h, C.byref(f0), C.byref(g0), 0)
            u0 = C.c_long(); lib.eqf_measure_in_propagate_stats(h, C.byref(u0), 0)
            print("   ", k, "spec calls/queued/cancelled", c0.value, q0_.value, x0.value, "selection frames/discarded", f0.value, g0.value, "me used", u0.value, "N", (S[k].shape[0] - 21) // 3)
        for k in ("m1 hold", "m1 nohold"):
            a = C.c_long(); b = C.c_long(); lib.eqf_gather_stats(fl[k].core_handle(), C.byref(a), 0); lib.eqf_hold_stats(fl[k].core_handle(), C.byref(b), 0)
            print("   ", k, "gather launches", a.value, "hold launches", b.value)
    if frames >= 3: break
