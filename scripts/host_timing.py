import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
from eqvio_amd.capi import VIOFilter
N = 200
settings = bench.eurocish_settings()
world, frames = bench.build_workload(seed=100, n_frames=120, N=N)
flt = bench.make_filter(world, settings, N, 0, frames, lambda s, sensor, ids, p, t: VIOFilter(s, max_landmarks=N, sensor=sensor, ids=ids, p=p, time=t))
flt.run_frames(world.cam, *bench.flatten_frames(frames[:20]))
acc = {"propagation": 0, "preprocessing": 0, "correction": 0}; tot = 0
for f in frames[20:100]:
    t0 = time.perf_counter()
    flt.run_frames(world.cam, *bench.flatten_frames([f]))
    tot += time.perf_counter() - t0
    for k, v in flt.last_timing().items(): acc[k] += v
print({k: round(1e6 * v / 80, 1) for k, v in acc.items()}, "sum", round(1e6 * sum(acc.values()) / 80, 1), "python-call wall", round(1e6 * tot / 80, 1))
