#!/usr/bin/env python
"""frame_mix's shipped-threshold world for a profiler: 200 warm-up frames, then `n` frames, nothing else (no event timing).
usage: [rocprofv3 --kernel-trace --stats -d out --] python scripts/dbg/shipped_frames.py [shipped|off|hover] [n] [landmarks]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if os.environ.get("WITH_TORCH"):  # torch's wheel brings its own libamdhip64.so.7: imported first, that runtime serves libeqf_hip.so as well (as in bench.py)
    import torch  # noqa: F401
import bench
from eqvio_amd.capi import PreparedFrames, VIOFilter, load_eqf_lib
from eqvio_amd.simworld import SimWorld
mode = sys.argv[1] if len(sys.argv) > 1 else "shipped"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
lib = load_eqf_lib()
N = int(sys.argv[3]) if len(sys.argv) > 3 else 200
s = bench.eurocish_settings()
if mode == "shipped":
    s.outlierThresholdAbs, s.outlierThresholdProb, s.featureRetention, s.initialPointVariance = 4.852186665580312, 0.03229809583062128, 0.18594708334486176, 129.90415638150924
if mode == "hover":  # the headline's quiet world
    world, frames = bench.build_workload(seed=100, n_frames=200 + n, N=N)
else:
    world = SimWorld(seed=321, num_points=2500, max_features=N, trajectory="wave", noise_px=0.5)
    frames = list(world.frames(200 + n))
sensor, ids, p = world.true_state(0.0, frames[0][2])
flt = VIOFilter(s, max_landmarks=N + 64, sensor=sensor, ids=ids, p=p, time=0.0)
pf = PreparedFrames(world.cam, *bench.flatten_frames(frames))
core = flt.core_handle()
flt.run_prepared(pf, 0, 200)
lib.eqf_synchronize(core)
t0 = time.perf_counter(); flt.run_prepared(pf, 200, n); lib.eqf_synchronize(core); wall = (time.perf_counter() - t0) / n
import ctypes as C
la, fb, live = C.c_long(), C.c_long(), C.c_long()
lib.eqf_lookahead_stats(core, C.byref(la), C.byref(fb), 0)
if hasattr(lib, "eqf_live_columns_stats"):
    lib.eqf_live_columns_stats(core, C.byref(live), 0)
sc, sq, scn, sf, sd = C.c_long(), C.c_long(), C.c_long(), C.c_long(), C.c_long()
lib.eqf_speculation_stats(core, C.byref(sc), C.byref(sq), C.byref(scn), 0)
lib.eqf_selection_stats(core, C.byref(sf), C.byref(sd), 0)
print(f"update calls {sc.value}: tail queued speculatively {sq.value} (cancelled on the device {scn.value}), outlier decision on the device {sf.value} (landmarks discarded {sd.value})")
print(f"{mode}: {1e6 * wall:.1f} us/frame ({1 / wall:.0f} updates/s); landmarks now {(flt.sigma_dim() - 21) // 3}; look-ahead launches {la.value} (redone on the chain {fb.value}), "
      f"of them ending behind the last live column {live.value}")
