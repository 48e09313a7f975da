"""Doorbell vs stream-wait stress: long runs must be bit-identical (see tests/test_gpu_robustness.py)."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, bench
from eqvio_amd.capi import VIOFilter, load_eqf_lib, OPT_DOORBELL
lib = load_eqf_lib()
settings = bench.eurocish_settings()
for N, nfr in ((50, 30000), (200, 8000)):
    world, frames = bench.build_workload(seed=7, n_frames=nfr + 2, N=N)
    outs = []
    for door in (1, 0):
        flt = bench.make_filter(world, settings, N, 0, frames, lambda s, se, i, p, t: VIOFilter(s, max_landmarks=N, device=0, sensor=se, ids=i, p=p, time=t))
        lib.eqf_set_option(flt.core_handle(), OPT_DOORBELL, door)
        flt.run_frames(world.cam, *bench.flatten_frames(frames[:nfr]))
        outs.append((flt.state_estimate(), flt.get_sigma()))
        flt.close()
    (a, ia, pa), Sa = outs[0]; (b, ib, pb), Sb = outs[1]
    print(N, nfr, "bit-identical:", np.array_equal(a, b) and np.array_equal(pa, pb) and np.array_equal(Sa, Sb), "finite:", np.isfinite(Sa).all())
