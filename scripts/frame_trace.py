"""Device-side frame timeline (EQF_OPT_TRACE, include/eqf_hip.h: eqf_trace_read): kernel start times stamped by the kernels
themselves plus the host's own stamps, tied together at the doorbell. No profiler, no events.
usage: python scripts/frame_trace.py [N] [frames] [frames per call]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from eqvio_amd.capi import VIOFilter, load_eqf_lib, OPT_TRACE
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
nfr = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
lib = load_eqf_lib()
world, frames = bench.build_workload(seed=100, n_frames=300 + nfr, N=N)
flt = bench.make_filter(world, bench.eurocish_settings(), N, 0, frames, lambda s, se, i, p, t: VIOFilter(s, max_landmarks=N, device=0, sensor=se, ids=i, p=p, time=t))
core = flt.core_handle()
flt.run_frames(world.cam, *bench.flatten_frames(frames[:300]))
assert lib.eqf_set_option(core, OPT_TRACE, 1) == 0
R = 1024  # frames in the ring
dev = np.zeros((R, 48), np.uint64); host = np.zeros((R, 8), np.int64); last = C.c_uint()
TICK = 0.01  # us per device tick (100 MHz)
rows = []
pos = 300
CH = int(sys.argv[3]) if len(sys.argv) > 3 else min(1000, nfr)  # frames per run_frames call (<= R - 4)
nsteps = 0
while pos + CH <= 300 + nfr:
    flt.run_frames(world.cam, *bench.flatten_frames(frames[pos:pos + CH])); pos += CH
    assert lib.eqf_trace_read(core, dev.ctypes.data_as(C.POINTER(C.c_ulonglong)), host.ctypes.data_as(C.POINTER(C.c_longlong)), C.byref(last)) == 0
    for back in range(CH - 2, 1, -1):  # complete frames only: f and f + 1 both in the ring, not across a read
        f = last.value - back
        d0, d1, h0, h1 = dev[f % R].astype(np.float64) * TICK, dev[(f + 1) % R].astype(np.float64) * TICK, host[f % R] * 1e-3, host[(f + 1) % R] * 1e-3
        if d0[0] == 0:  # fused assembly (the default): no assembly kernel, the frame starts with the propagation kernel
            d0[0], d1[0] = d0[1], d1[1]
        if d0[0] == 0 or d1[0] == 0 or d0[41] == 0:
            continue
        nsteps = int(np.count_nonzero(d0[3:35]))
        t0 = d0[0]
        e = dict(period=d1[0] - d0[0], assemble=d0[1] - d0[0], propagate=d0[2] - d0[1], build_Z=d0[3] - d0[2], chain=d0[40] - d0[3], step=(d0[3 + nsteps - 1] - d0[3]) / max(nsteps - 1, 1),
                 lift=d0[41] - d0[40], lift_to_syrk=d0[42] - d0[41], syrk=d0[43] - d0[42], syrk_end_to_next=d1[0] - d0[43], door_to_next_kernel=d1[0] - d0[41])
        # host stamps relative to the doorbell of frame f (host stamp 0 of frame f == device slot 41 of frame f)
        hd = h0[0]
        e.update(h_next_prop_entry=h1[1] - hd, h_next_assemble_out=h1[2] - hd, h_next_prop_out=h1[3] - hd, h_next_tail_entry=h1[4] - hd, h_next_buildZ_out=h1[5] - hd, h_next_tail_out=h1[6] - hd)
        # the same host events of frame f + 1 on the device time axis of frame f + 1: when did the GPU start what the host had just launched?
        e.update(d_next_assemble_start=d1[0] - d0[41], d_next_propagate_start=d1[1] - d0[41], d_next_buildZ_start=d1[2] - d0[41], d_next_step0_start=d1[3] - d0[41], d_next_lift_start=d1[40] - d0[41])
        rows.append(e)
print(f"N={N}: {len(rows)} frames, factorisation steps per frame: {nsteps}")
per = np.array([r["period"] for r in rows])
print("  period by position inside a call (mean of 100 frames each):", " ".join(f"{per[k:k + 100].mean():.1f}" for k in range(0, min(len(per), CH - 3), 100)))
for k in rows[0]:
    v = np.array([r[k] for r in rows])
    print(f"  {k:26s} median {np.median(v):8.2f} us   mean {v.mean():8.2f}   (min {v.min():8.2f}, p90 {np.percentile(v, 90):8.2f}, max {v.max():8.2f})")
