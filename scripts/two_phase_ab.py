"""A/B of the two-phase factorisation steps (EQF_OPT_TWO_PHASE) at a given N: throughput for several thresholds and a bit-identity check.
usage: python scripts/two_phase_ab.py [N] [frames]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
from eqvio_amd.capi import VIOFilter, PreparedFrames, load_eqf_lib, OPT_TWO_PHASE
N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
nfr = int(sys.argv[2]) if len(sys.argv) > 2 else 300
lib = load_eqf_lib()
world, frames = bench.build_workload(seed=100, n_frames=60 + nfr + 2, N=N)
pf = PreparedFrames(world.cam, *bench.flatten_frames(frames[:60 + nfr]))
ref = None
for thr in (0, 2000, 1500, 1000, 700, 500, 300, 1):
    flt = bench.make_filter(world, bench.eurocish_settings(), N, 0, frames, lambda s, se, i, p, t: VIOFilter(s, max_landmarks=N, device=0, sensor=se, ids=i, p=p, time=t))
    core = flt.core_handle()
    assert lib.eqf_set_option(core, OPT_TWO_PHASE, thr) == 0
    flt.run_prepared(pf, 0, 60)
    lib.eqf_synchronize(core)
    t0 = time.perf_counter()
    flt.run_prepared(pf, 60, nfr)
    lib.eqf_synchronize(core)
    el = time.perf_counter() - t0
    out = (flt.state_estimate(), flt.get_sigma())
    if ref is None:
        ref = out
    same = all(np.array_equal(a, b) for a, b in zip(out[0], ref[0])) and np.array_equal(out[1], ref[1])
    print(f"N={N} threshold {thr:5d} tiles: {nfr / el:8.1f} updates/s ({1e6 * el / nfr:7.1f} us/frame)  bit-identical to threshold 0: {same}", flush=True)
    flt.close()
