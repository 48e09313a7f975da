"""Where a frame's wall time goes on the host: doorbell spinning (GPU-bound share) vs the host's own work (launch calls, filter logic).
usage: python scripts/host_share.py [N] [frames]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from eqvio_amd.capi import VIOFilter, PreparedFrames, load_eqf_lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
nfr = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
lib = load_eqf_lib()
world, frames = bench.build_workload(seed=100, n_frames=300 + nfr, N=N)
flt = bench.make_filter(world, bench.eurocish_settings(), N, 0, frames, lambda s, se, i, p, t: VIOFilter(s, max_landmarks=N, device=0, sensor=se, ids=i, p=p, time=t))
core = flt.core_handle()
flt.run_frames(world.cam, *bench.flatten_frames(frames[:300]))
calls, secs = (C.c_long * 2)(), (C.c_double * 2)()
for rep in range(3):
    chunk = PreparedFrames(world.cam, *bench.flatten_frames(frames[300 + rep * (nfr // 3):300 + (rep + 1) * (nfr // 3)]))
    lib.eqf_synchronize(core)
    lib.eqf_host_wait_stats(core, calls, secs, 1)
    t0 = time.perf_counter()
    flt.run_prepared(chunk)
    lib.eqf_synchronize(core)
    el = time.perf_counter() - t0
    lib.eqf_host_wait_stats(core, calls, secs, 1)
    print(f"N={N}: frame {1e6 * el / (nfr // 3):7.1f} us = host spinning on the doorbell {1e6 * secs[0] / (nfr // 3):7.1f} us ({calls[0] / (nfr // 3):.2f} waits/frame) + host work {1e6 * (el - secs[0]) / (nfr // 3):6.1f} us, of which {calls[1] / (nfr // 3):.1f} launch calls {1e6 * secs[1] / (nfr // 3):6.1f} us")
