"""A/B of one eqf option on the bench.py workload, same box, alternating runs.
usage: python scripts/ab_option.py <option id>[:valA:valB] [N] [steps]"""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
from eqvio_amd.capi import VIOFilter, PreparedFrames, load_eqf_lib, OPT_TIMING

opt = int(sys.argv[1].split(":")[0])
VALS = tuple(int(v) for v in sys.argv[1].split(":")[1:]) or (1, 0)  # "3:2:0" = option 3, values 2 and 0
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 600
lib = load_eqf_lib()
settings = bench.eurocish_settings()
world, frames = bench.build_workload(seed=100, n_frames=200 + 8 * steps + 60, N=N)
flt = bench.make_filter(world, settings, N, 0, frames, lambda s, se, i, p, t: VIOFilter(s, max_landmarks=N, device=0, sensor=se, ids=i, p=p, time=t))
core = flt.core_handle()
flt.run_frames(world.cam, *bench.flatten_frames(frames[:200]))
pos = 200
for rep in range(4):
    for val in VALS:
        lib.eqf_set_option(core, opt, val)
        chunk = PreparedFrames(world.cam, *bench.flatten_frames(frames[pos:pos + steps])); pos += steps
        lib.eqf_synchronize(core)
        t0 = time.perf_counter()
        flt.run_prepared(chunk)
        lib.eqf_synchronize(core)
        el = time.perf_counter() - t0
        print(f"rep {rep} option {opt}={val}: {steps / el:8.1f} updates/s", flush=True)
# per-launch spans for both settings
from eqvio_amd.capi import EqfCore
for val in VALS:
    lib.eqf_set_option(core, opt, val)
    lib.eqf_set_option(core, OPT_TIMING, 1)
    agg = collections.OrderedDict(); seq = []
    nf = 10
    for k in range(nf):
        flt.run_frames(world.cam, *bench.flatten_frames(frames[pos:pos + 1])); pos += 1
        which = np.zeros(4096, np.int32); us = np.zeros(4096, np.float32)
        import ctypes as C
        cnt = lib.eqf_last_kernel_times(core, which.ctypes.data_as(C.POINTER(C.c_int)), us.ctypes.data_as(C.POINTER(C.c_float)), 4096)
        names = [lib.eqf_kernel_name(int(which[i])).decode() for i in range(cnt)]
        for nme, u in zip(names, us[:cnt]):
            agg.setdefault(nme, []).append(float(u))
        if k == nf - 1:
            seq = [(nme, round(float(u), 1)) for nme, u in zip(names, us[:cnt])]
    lib.eqf_set_option(core, OPT_TIMING, 0)
    print(f"option {opt}={val} spans us/frame:", {k: round(sum(v) / nf, 1) for k, v in agg.items()}, "total", round(sum(sum(v) for v in agg.values()) / nf, 1))
    print("   last frame launch sequence:", seq)
