"""Several independent filters on ONE MI355X as R PROCESSES (one VIOFilter + eqf_ctx + stream + HIP runtime each), against scripts/multi_filter.py's R threads
of one process. Same workload as bench.py. usage: python scripts/multi_process.py [N] [steps] [R,R,...]"""
import json, multiprocessing as mp, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def worker(r, N, steps, warm, barrier, out):
    import bench
    from eqvio_amd.capi import VIOFilter, load_eqf_lib
    lib = load_eqf_lib()
    settings = bench.eurocish_settings()
    world, frames = bench.build_workload(seed=100 + r, n_frames=warm + steps + 2, N=N)
    flt = bench.make_filter(world, settings, N, 0, frames, lambda s, se, i, p, t: VIOFilter(s, max_landmarks=N, device=0, sensor=se, ids=i, p=p, time=t))
    flt.run_frames(world.cam, *bench.flatten_frames(frames[:warm]))
    lib.eqf_synchronize(flt.core_handle())
    timed = bench.flatten_frames(frames[warm:warm + steps])
    barrier.wait()
    t0 = time.perf_counter()
    flt.run_frames(world.cam, *timed)
    lib.eqf_synchronize(flt.core_handle())
    out.put((r, t0, time.perf_counter()))


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 600
    Rs = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 3, 4, 8]
    ctx = mp.get_context("spawn")
    res = {}
    for R in Rs:
        barrier, out = ctx.Barrier(R), ctx.Queue()
        ps = [ctx.Process(target=worker, args=(r, N, steps, 100, barrier, out)) for r in range(R)]
        for p in ps: p.start()
        got = [out.get(timeout=600) for _ in ps]
        for p in ps: p.join()
        el = max(g[2] for g in got) - min(g[1] for g in got)
        res[R] = R * steps / el
        print(f"N={N} processes={R}: {R * steps / el:9.1f} updates/s aggregate ({steps / el:8.1f} per filter)", flush=True)
    print(json.dumps({"N": N, "steps": steps, "aggregate_updates_per_s": res}))
