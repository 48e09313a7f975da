"""Per-step timeline inside the look-ahead factorisation kernel (EQF_OPT_TRACE; include/eqf_hip.h: eqf_debug_lookahead_stamps).
usage: python scripts/lookahead_trace.py [N]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from eqvio_amd.capi import VIOFilter, load_eqf_lib, OPT_TRACE
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
lib = load_eqf_lib()
world, frames = bench.build_workload(seed=100, n_frames=400, N=N)
flt = bench.make_filter(world, bench.eurocish_settings(), N, 0, frames, lambda s, se, i, p, t: VIOFilter(s, max_landmarks=N, device=0, sensor=se, ids=i, p=p, time=t))
core = flt.core_handle()
flt.run_frames(world.cam, *bench.flatten_frames(frames[:300]))
assert lib.eqf_set_option(core, OPT_TRACE, 1) == 0
acc = []
for f in range(300, 400):
    flt.run_frames(world.cam, *bench.flatten_frames(frames[f:f + 1]))
    buf = np.zeros((96, 8), np.uint64)
    assert lib.eqf_debug_lookahead_stamps(core, buf.ctypes.data_as(C.POINTER(C.c_ulonglong))) == 0
    acc.append(buf.astype(np.float64) * 0.01)
a = np.median(np.stack(acc), axis=0)
NJ = (2 * N + 31) // 32
t0 = a[0, 0]
print(f"N={N}, {NJ} panels; microseconds from the owner's first stamp (medians over 100 frames)")
print("owner step k: tail-start  tiles-in  tail-done  post-start  elim-done  D-written  elim-start | step period   (elim-done of row k = L_k ready; tail of step k prepares block row k + 2)")
for k in range(NJ - 1):
    r = a[k] - t0
    per = (a[k + 1, 6] - a[k, 6]) if k + 2 < NJ else float("nan")
    print(f"  k={k:2d}  " + "  ".join(f"{v:8.2f}" if a[k, i] > 0 else "       -" for i, v in enumerate(r[:7])) + f" | {per:6.2f}")
for name, base in (("first T block row", 32), (f"S block row {NJ - 2}", 64)):
    print(name + ", panel p: L+tile in LDS   P done   wave 0: flags seen, updates done | all waves: flags seen, updates done | wave 3, wave 7 done")
    for p in range(NJ):
        if a[base + p, 0] > 0:
            r = a[base + p] - t0
            cols = [0, 1, 7, 2, 3, 4, 6, 5]
            print(f"  p={p:2d}  " + "  ".join(f"{r[i]:8.2f}" if a[base + p, i] > 0 else "       -" for i in cols))
