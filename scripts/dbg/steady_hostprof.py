"""host scopes (libraries built with -DEQF_HOST_PROFILE, EQVIO_AMD_LIB_DIR) of the steady headline workload"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from eqvio_amd.capi import VIOFilter, PreparedFrames, load_eqf_lib
lib = load_eqf_lib()
N = 200
world, frames = bench.build_workload(seed=100, n_frames=2300, N=N)
flt = bench.make_filter(world, bench.eurocish_settings(), N, 0, frames, lambda s, se, i, p, t: VIOFilter(s, max_landmarks=N, device=0, sensor=se, ids=i, p=p, time=t))
pf = PreparedFrames(world.cam, *bench.flatten_frames(frames))
flt.run_prepared(pf, 0, 2300)
lib.eqf_synchronize(flt.core_handle())
