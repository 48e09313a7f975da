import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from eqvio_amd.capi import EqfCore, EqfError, OPT_LOOKAHEAD
from util import *
from test_gpu_parity import make_pair
for N,M in [(200,200),(40,40)]:
    rng, settings, orc, core, (xi0, Xs, ids, q0, Q, S) = make_pair(CHARTS["invdepth"], N, seed=N + M, useDiscreteInnovationLift=0)
    cam = default_camera()
    mid, y = synth_measurement(rng, cam, ids, q0, Q, noise_px=1.0, subset=np.sort(rng.permutation(N)[:M]))
    m=2*M; n=21+3*N; rows=m+n+1
    ref=None
    for it in range(80):
        la = 0 if it==0 else 1
        c = EqfCore(N, CHARTS["invdepth"])
        c.set_state(xi0, Xs, ids, q0, Q); c.set_sigma(S); c.set_option(OPT_LOOKAHEAD, la)
        c.vision_update(cam, mid, y, settings.measurementNoise**2, True, False)
        Sg=c.get_sigma(); W=c.debug_get_W(rows,m)[m:]; g=c.last_gamma()
        if it==0: ref=(W,g,Sg); continue
        if not np.array_equal(Sg,ref[2]): print(N,'it',it,'Sigma differs', np.abs(Sg-ref[2]).max(), 'W equal', np.array_equal(W,ref[0]), 'gamma diff', np.abs(g-ref[1]).max(), 'Sigma - (S - W W^T):', np.abs(Sg-(S-W@W.T)).max(), np.abs(ref[2]-(S-W@W.T)).max())
        if not np.array_equal(W,ref[0]):
            d=~(W==ref[0]); r,cc=np.nonzero(d)
            print(N,"it",it,"W differs: block rows",np.unique(r//32),"panels",np.unique(cc//32),"max",np.abs(W-ref[0]).max(), "first bad col", cc.min(), "rows in first bad panel", np.unique(r[cc//32==cc.min()//32]//32))
        elif np.linalg.norm(g-ref[1])>1e-12*np.linalg.norm(ref[1]):
            print(N,"it",it,"gamma differs only", np.abs(g-ref[1]).max())
    print(N,"done")
