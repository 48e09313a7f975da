"""Soak version of door_stress.py: 360 000 frames at three sizes, doorbell vs stream wait, must be bit-identical (96 s on one MI355X)."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, bench
from eqvio_amd.capi import VIOFilter, PreparedFrames, load_eqf_lib, OPT_DOORBELL
lib = load_eqf_lib()
settings = bench.eurocish_settings()
for N, nfr in ((200, 60000), (50, 150000), (8, 150000)):
    world, frames = bench.build_workload(seed=11, n_frames=nfr + 2, N=N)
    pf = PreparedFrames(world.cam, *bench.flatten_frames(frames[:nfr]))
    outs = []
    for door in (1, 0):
        flt = bench.make_filter(world, settings, N, 0, frames, lambda s, se, i, p, t: VIOFilter(s, max_landmarks=N, device=0, sensor=se, ids=i, p=p, time=t))
        lib.eqf_set_option(flt.core_handle(), OPT_DOORBELL, door)
        assert flt.run_prepared(pf) == nfr
        outs.append((flt.state_estimate(), flt.get_sigma()))
        flt.close()
    (a, ia, pa), Sa = outs[0]; (b, ib, pb), Sb = outs[1]
    print(N, nfr, "bit-identical:", np.array_equal(a, b) and np.array_equal(pa, pb) and np.array_equal(Sa, Sb), "finite:", np.isfinite(Sa).all(), flush=True)
