"""Per-launch hipEvent spans of k_chol_lookahead in bench.py's span pass (EQF_OPT 100), frame by frame: is a slow pass slow throughout (clock state) or at its start?
usage: python scripts/span_distribution.py [N] [frames]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from eqvio_amd.capi import VIOFilter, load_eqf_lib, OPT_TIMING
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
nfr = int(sys.argv[2]) if len(sys.argv) > 2 else 100
lib = load_eqf_lib()
world, frames = bench.build_workload(seed=100, n_frames=2300 + nfr, N=N)
flt = bench.make_filter(world, bench.eurocish_settings(), N, 0, frames, lambda s, se, i, p, t: VIOFilter(s, max_landmarks=N, device=0, sensor=se, ids=i, p=p, time=t))
core = flt.core_handle()
for rep in range(3):
    flt.run_frames(world.cam, *bench.flatten_frames(frames[rep * 700:(rep + 1) * 700]))  # the GPU has been busy for 60 ms, as behind bench.py's timed region
    lib.eqf_set_option(core, OPT_TIMING, 1)
    flt.run_frames(world.cam, *bench.flatten_frames(frames[2100 + rep * 30:2100 + rep * 30 + 30]))
    which = np.zeros(65536, np.int32); us = np.zeros(65536, np.float32)
    cnt = lib.eqf_last_kernel_times(core, which.ctypes.data_as(C.POINTER(C.c_int)), us.ctypes.data_as(C.POINTER(C.c_float)), len(us))
    lib.eqf_set_option(core, OPT_TIMING, 0)
    la = [float(us[i]) for i in range(cnt) if lib.eqf_kernel_name(int(which[i])).decode() == "k_chol_lookahead"]
    print("pass %d: k_chol_lookahead spans (us), frame by frame: %s | mean %.2f median %.2f min %.2f" % (rep, " ".join("%.1f" % v for v in la), np.mean(la), np.median(la), np.min(la)))
