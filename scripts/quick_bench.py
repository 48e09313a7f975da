"""First-look timing of one EqF frame at N landmarks through the C-ABI, with per-kernel HIP-event times."""
import sys, os, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from eqvio_amd.capi import EqfCore, OPT_TIMING, OPT_RICCATI_DENSE, COORD_INVDEPTH
from util import *
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dense = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(1)
chart = COORD_INVDEPTH
settings = settings_for(chart, fastRiccati=1, useDiscreteInnovationLift=0, initialPointVariance=9.0, measurementNoise=1.93)
xi0, Xs, ids, q0, Q = reasonable_state(rng, N)
core = EqfCore(N, chart)
core.set_state(xi0, Xs, ids, q0, Q)
core.set_sigma(np.diag(settings.initial_cov_diag(N)))
print("fp64 MFMA peak (ubench):", core.mfma_f64_peak(), "TFLOP/s")
core.set_option(OPT_RICCATI_DENSE, dense)
cam = euroc_camera()
Qd, Pd = settings.input_gain_diag12(), settings.state_gain_diag8()
k = 10
def frame(timing=False):
    imus = np.stack([random_imu(rng) * np.array([1] + [0.02] * 3 + [0.1] * 3 + [0] * 6) for _ in range(k)])
    imus[:, 4:7] += 0  # body-frame specific force left random-small: the state is synthetic
    dts = np.full(k, 0.005)
    mean = (imus * dts[:, None]).sum(0) / dts.sum()
    core.integrate_riccati_fast(mean, dts.sum(), Qd, Pd)
    core.integrate_observer(imus, dts, True)
    _, Xs_, ids_, q0_, Q_ = core.get_state()
    mid, y = synth_measurement(rng, cam, ids_, q0_, Q_, noise_px=1.0)
    t0 = time.perf_counter()
    core.vision_update(cam, mid, y, settings.measurementNoise**2, True, False)
    return time.perf_counter() - t0
for _ in range(3): frame()
core.set_option(OPT_TIMING, 1)
agg = collections.OrderedDict()
nf = 5
for _ in range(nf):
    frame()
    for name, us in core.kernel_times():
        agg.setdefault(name, []).append(us)
tot = 0
for name, v in agg.items():
    per_frame = sum(v) / nf
    tot += per_frame
    print(f"{name:20s} launches/frame {len(v)/nf:5.1f}  us/frame {per_frame:9.1f}  avg us/launch {np.mean(v):8.2f}")
print("sum of kernel spans per frame: %.1f us" % tot)
core.set_option(OPT_TIMING, 0)
# wall time of update call alone and of the whole frame (python + ctypes overhead included)
ts = [frame() for _ in range(20)]
print("vision_update wall (python call) median %.1f us" % (1e6 * np.median(ts)))
S = core.get_sigma(); print("Sigma finite", np.isfinite(S).all(), "min eig", np.linalg.eigvalsh(0.5*(S+S.T)).min())
