"""Several independent filters on ONE MI355X ("optionally several filters per GPU to fill CUs"):
R host threads, one VIOFilter + eqf_ctx + stream each, same workload as bench.py. Prints aggregate updates/s."""
import os, sys, threading, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
from eqvio_amd.capi import VIOFilter, load_eqf_lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 600
warm = 100
Rs = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 4, 8]
lib = load_eqf_lib()
settings = bench.eurocish_settings()
res = {}
for R in Rs:
    flts, work = [], []
    for r in range(R):
        world, frames = bench.build_workload(seed=100 + r, n_frames=warm + steps + 2, N=N)
        flt = bench.make_filter(world, settings, N, 0, frames, lambda s, se, i, p, t: VIOFilter(s, max_landmarks=N, device=0, sensor=se, ids=i, p=p, time=t))
        flts.append(flt)
        work.append((world.cam, bench.flatten_frames(frames[:warm]), bench.flatten_frames(frames[warm:warm + steps])))
    for f, (cam, w, _) in zip(flts, work):
        f.run_frames(cam, *w)
        lib.eqf_synchronize(f.core_handle())
    barrier = threading.Barrier(R + 1)
    def run(f, cam, t):
        barrier.wait()
        f.run_frames(cam, *t)
        lib.eqf_synchronize(f.core_handle())
    ths = [threading.Thread(target=run, args=(f, cam, t)) for f, (cam, _, t) in zip(flts, work)]
    for t in ths: t.start()
    barrier.wait(); t0 = time.perf_counter()
    for t in ths: t.join()
    el = time.perf_counter() - t0
    res[R] = R * steps / el
    print(f"N={N} filters={R}: {R * steps / el:9.1f} updates/s aggregate ({steps / el:8.1f} per filter)", flush=True)
    for f in flts: f.close()
print(json.dumps({"N": N, "steps": steps, "aggregate_updates_per_s": res}))
