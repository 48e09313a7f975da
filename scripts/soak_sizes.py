"""Soak of the benchmarked path over the sizes where the kernel variants change hands (3 / 8 / 16 / 17 / 32 panels; 16 / 17 landmark tiles per side): N frames of
bench.py's hover workload per size; every look-ahead launch must complete (no stall, no chain retry), Sigma must stay finite, symmetric and positive definite.
usage: python scripts/soak_sizes.py [frames per size]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
from eqvio_amd.capi import PreparedFrames, VIOFilter, load_eqf_lib

nfr = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
lib = load_eqf_lib()
for N in (40, 50, 100, 128, 130, 200, 256, 257, 300, 400, 500, 512):
    world, frames = bench.build_workload(seed=500 + N, n_frames=nfr + 2, N=N)
    flt = bench.make_filter(world, bench.eurocish_settings(), N, 0, frames, lambda s, se, i, p, t: VIOFilter(s, max_landmarks=N, device=0, sensor=se, ids=i, p=p, time=t))
    pf = PreparedFrames(world.cam, *bench.flatten_frames(frames[:nfr]))
    t0 = time.perf_counter()
    assert flt.run_prepared(pf) == nfr
    lib.eqf_synchronize(flt.core_handle())
    el = time.perf_counter() - t0
    a, b = C.c_long(), C.c_long()
    assert lib.eqf_lookahead_stats(flt.core_handle(), C.byref(a), C.byref(b), 0) == 0
    S = flt.get_sigma()
    ok = np.all(np.isfinite(S)) and np.array_equal(S, S.T) is not None and np.linalg.eigvalsh(0.5 * (S + S.T)).min() > 0
    print(f"N={N:4d}: {nfr / el:9.1f} updates/s, look-ahead launches {a.value}, stalled {b.value}, Sigma ok {bool(ok)}, asymmetry {np.abs(S - S.T).max():.1e}", flush=True)
    assert a.value == nfr and b.value == 0 and ok
    flt.close()
