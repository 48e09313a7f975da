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
print("owner step k: tail-start  tiles-in  tail-done  post-start  elim-done  D-written  elim-start  handoff-seen | step period   (elim-done of row k = L_k ready; tail of step k prepares block row k + 2)")
for k in range(NJ - 1):
    r = a[k] - t0
    per = (a[k + 1, 6] - a[k, 6]) if k + 2 < NJ else float("nan")
    print(f"  k={k:2d}  " + "  ".join(f"{v:8.2f}" if a[k, i] > 0 else "       -" for i, v in enumerate(r[:8])) + f" | {per:6.2f}")
print("tails of step k, microseconds after L_k was ready: b in LDS (wave 3) | done: wave 2, wave 3, wave 5, wave 7 | pivot wave starts step k + 1 | L_(k+1) ready")
for k in range(1, NJ - 2):
    lk = a[k, 4]
    print(f"  k={k:2d}  {a[32 + k, 6] - lk:6.2f} | {a[32 + k, 3] - lk:6.2f} {a[32 + k, 4] - lk:6.2f} {a[k, 2] - lk:6.2f} {a[32 + k, 5] - lk:6.2f} | {a[k + 1, 3] - lk:6.2f} | {a[k + 1, 4] - lk:6.2f}")
print("first T block row, panel p: L+tile in LDS   P done   updates done   (relative to L_p ready in the owner)")
for p in range(NJ):
    if a[32 + p, 0] > 0:
        r = a[32 + p] - t0
        lp = (a[p, 4] - t0) if p >= 1 else float("nan")
        print(f"  p={p:2d}  {r[0]:8.2f} {r[1]:8.2f} {r[2]:8.2f}   ({r[0] - lp:5.2f} {r[1] - lp:5.2f} {r[2] - lp:5.2f})")
if NJ > 16:  # la_row2 stamps every wave: where does a panel of the first T half-row go? (microseconds after its own panel top)
    print("first T block row (17 .. 32 panels), microseconds after the panel's top: W rows stored | flags seen: wave 0, last wave | trailing update done: wave 0, wave 3, wave 7, last wave | next panel's top")
    for p in range(NJ - 1):
        if a[32 + p, 0] > 0 and a[32 + p + 1, 0] > 0:
            r = a[32 + p] - a[32 + p, 0]
            print(f"  p={p:2d}  {r[1]:6.2f} | {r[7]:6.2f} {r[3]:6.2f} | {r[2]:6.2f} {r[6]:6.2f} {r[5]:6.2f} {r[4]:6.2f} | {a[32 + p + 1, 0] - a[32 + p, 0]:6.2f}")
print("S block row I (top half) at its last panel p = I - 3, microseconds after L_p was ready in the owner: waiting for it | L+tile in LDS | P_h flag | updates done | hand-off flag | seen by the owner's wave 3   (the tail of step I - 2 uses it)")
for I in range(3, NJ):
    if a[64 + I, 0] > 0:
        lp = a[I - 3, 4] if I - 3 >= 1 else float("nan")
        r = a[64 + I] - lp
        seen = a[I - 2, 7] - lp
        print(f"  I={I:2d}  {r[5]:8.2f} {r[0]:8.2f} {r[1]:8.2f} {r[2]:8.2f} {r[4]:8.2f} {seen:8.2f}")
