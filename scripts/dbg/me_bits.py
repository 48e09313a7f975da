import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from eqvio_amd.capi import VIOFilter, PreparedFrames, load_eqf_lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
lib = load_eqf_lib()
world, frames = bench.build_workload(seed=7, n_frames=8, N=N)
s = bench.eurocish_settings()
mk = lambda s_, sensor, ids, p, t: VIOFilter(s_, max_landmarks=N, sensor=sensor, ids=ids, p=p, time=t)
a = bench.make_filter(world, s, N, None, frames, mk)
b = bench.make_filter(world, s, N, None, frames, mk)
assert lib.eqf_set_option(b.core_handle(), 19, 0) == 0
pf = PreparedFrames(world.cam, *bench.flatten_frames(frames))
for f in range(8):
    a.run_prepared(pf, f, 1); b.run_prepared(pf, f, 1)
    sa, sb = a.state_estimate(), b.state_estimate()
    Sa, Sb = a.get_sigma(), b.get_sigma()
    print(f, "state equal", all(np.array_equal(x, y) for x, y in zip(sa, sb)), "Sigma equal", np.array_equal(Sa, Sb), "max dSigma", float(np.max(np.abs(Sa - Sb))))
