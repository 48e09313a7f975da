import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from eqvio_amd.capi import EqfCore, EqfError, OPT_LOOKAHEAD
from util import *
from test_gpu_parity import make_pair
mode=sys.argv[1] if len(sys.argv)>1 else "a"
for N,M in [(200,200),(40,40)]:
    rng, settings, orc, core, (xi0, Xs, ids, q0, Q, S) = make_pair(CHARTS["invdepth"], N, seed=N + M, useDiscreteInnovationLift=0)
    cam = default_camera()
    mid, y = synth_measurement(rng, cam, ids, q0, Q, noise_px=1.0, subset=np.sort(rng.permutation(N)[:M]))
    m=2*M; n=21+3*N; rows=m+n+1
    ref=None; nbad=0
    for it in range(80):
        la = 0 if it==0 else 1
        c = EqfCore(N, CHARTS["invdepth"])
        c.set_state(xi0, Xs, ids, q0, Q); c.set_sigma(S); c.set_option(OPT_LOOKAHEAD, la)
        c.vision_update(cam, mid, y, settings.measurementNoise**2, True, False)
        if mode=="b": c.synchronize()
        Sg=c.get_sigma()
        if it==0: ref=Sg; continue
        if not np.array_equal(Sg,ref):
            nbad+=1; W=c.debug_get_W(rows,m)[m:]
            print(N,'it',it,'Sigma differs', np.abs(Sg-ref).max(), 'Sigma - (S - W W^T):', np.abs(Sg-(S-W@W.T)).max(), np.abs(ref-(S-W@W.T)).max())
    print(N,"done bad",nbad)
