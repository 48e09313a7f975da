import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from eqvio_amd.capi import EqfCore, EqfError, OPT_LOOKAHEAD
from util import *
from test_gpu_parity import make_pair
LA=int(sys.argv[1]); keep_c1=int(sys.argv[2])
for N,M in [(200,200),(40,40),(60,33),(224,224)]:
    rng, settings, orc, core, (xi0, Xs, ids, q0, Q, S) = make_pair(CHARTS["invdepth"], N, seed=N + M, useDiscreteInnovationLift=0)
    cam = default_camera()
    mid, y = synth_measurement(rng, cam, ids, q0, Q, noise_px=1.0, subset=np.sort(rng.permutation(N)[:M]))
    ref=None; stats={}
    if keep_c1:
        c1 = EqfCore(N, CHARTS["invdepth"]); c1.set_option(OPT_LOOKAHEAD, 1)
    for it in range(60):
        la = 0 if it==0 else LA
        c = EqfCore(N, CHARTS["invdepth"]) if (it<30 or not keep_c1) else c1
        c.set_state(xi0, Xs, ids, q0, Q); c.set_sigma(S); c.set_option(OPT_LOOKAHEAD, la)
        try:
            c.vision_update(cam, mid, y, settings.measurementNoise**2, True, False); err=None
        except EqfError as e: err=e.code
        Sg=c.get_sigma()
        if it==0: ref=Sg; continue
        d=np.abs(Sg-ref); bad=np.unique(np.nonzero(~(d==0))[0]//32) if not np.array_equal(Sg,ref) else []
        key=("fresh" if it<30 else "reused", err, float(np.nanmax(d)) if len(bad) else 0.0)
        stats[key]=stats.get(key,0)+1
    print(N,M, stats)
