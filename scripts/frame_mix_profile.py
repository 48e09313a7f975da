#!/usr/bin/env python
"""Where a frame of bench.py's frame_mix workloads goes: hipEvent time per kernel per frame (EQF_OPT_TIMING spans) next to the wall time.
usage: python scripts/frame_mix_profile.py [shipped|off]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
from eqvio_amd.capi import OPT_TIMING, PreparedFrames, VIOFilter, load_eqf_lib
from eqvio_amd.simworld import SimWorld
mode = sys.argv[1] if len(sys.argv) > 1 else "shipped"
opts = [tuple(int(v) for v in a.split("=")) for a in sys.argv[2:]]  # extra arguments "option=value" are set on the core (A/B of eqf_set_option switches)
lib = load_eqf_lib()
N = 200
s = bench.eurocish_settings()
if mode == "shipped":
    s.outlierThresholdAbs, s.outlierThresholdProb, s.featureRetention, s.initialPointVariance = 4.852186665580312, 0.03229809583062128, 0.18594708334486176, 129.90415638150924
world = SimWorld(seed=321, num_points=2500, max_features=N, trajectory="wave", noise_px=0.5)
frames = list(world.frames(700))
ids0 = frames[0][2]
sensor, ids, p = world.true_state(0.0, ids0)
flt = VIOFilter(s, max_landmarks=N + 64, sensor=sensor, ids=ids, p=p, time=0.0)
pf = PreparedFrames(world.cam, *bench.flatten_frames(frames))
core = flt.core_handle()
for o, v in opts:
    assert lib.eqf_set_option(core, o, v) == 0
flt.run_prepared(pf, 0, 200)
lib.eqf_synchronize(core)
t0 = time.perf_counter(); flt.run_prepared(pf, 200, 300); lib.eqf_synchronize(core); wall = (time.perf_counter() - t0) / 300
lib.eqf_set_option(core, OPT_TIMING, 1)
nf = 100
t0 = time.perf_counter(); flt.run_prepared(pf, 500, nf); lib.eqf_synchronize(core); wall_t = (time.perf_counter() - t0) / nf
which, us = np.zeros(1 << 18, np.int32), np.zeros(1 << 18, np.float32)
cnt = lib.eqf_last_kernel_times(core, which.ctypes.data_as(C.POINTER(C.c_int)), us.ctypes.data_as(C.POINTER(C.c_float)), len(us))
agg = {}
for i in range(cnt):
    agg.setdefault(lib.eqf_kernel_name(int(which[i])).decode(), []).append(float(us[i]))
print(f"{mode}: wall {1e6 * wall:.1f} us/frame ({1 / wall:.0f} updates/s), with event timing on {1e6 * wall_t:.1f} us/frame; landmarks now {(flt.sigma_dim() - 21) // 3}")
tot = 0
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {k:22s} {sum(v) / nf:8.2f} us/frame   {len(v) / nf:5.2f} spans/frame   {np.mean(v):7.2f} us/span")
    tot += sum(v) / nf
print(f"  sum of spans {tot:.1f} us/frame")
calls, secs = (C.c_long * 2)(), (C.c_double * 2)()
lib.eqf_host_wait_stats(core, calls, secs, 0)
gl = C.c_long()
if hasattr(lib, "eqf_gather_stats") and lib.eqf_gather_stats(core, C.byref(gl), 0) == 0:
    print("propagation launches that applied a removal record themselves:", gl.value)
if hasattr(lib, "eqf_hold_stats") and lib.eqf_hold_stats(core, C.byref(gl), 0) == 0:
    print("propagation launches that created the frame's new landmarks themselves:", gl.value)
print("host: doorbell waits", calls[0], f"{1e6 * secs[0] / max(calls[0], 1):.1f} us each; launches", calls[1], f"{1e6 * secs[1] / max(calls[1], 1):.2f} us each")
