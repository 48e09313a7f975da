#!/usr/bin/env python
"""A/B for an fp32 ARITHMETIC path (BASELINE config 5 / SURVEY §7.8): the same filter run with Sigma -= W W^T on the f64 MFMA and on
v_mfma_f32_16x16x4_f32 (EQF_OPT_SYRK_F32: operands rounded to float, twice the issue rate), at N = 200 and N = 500, bench.py's workload.
Reports frame rate, per-kernel hipEvent time and the deviation of state / Sigma between the two runs after the same frames.
Output: one JSON document (profiles/r02_fp32_ab.json)."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import bench  # noqa: E402
from eqvio_amd.capi import OPT_SYRK_F32, OPT_TIMING, PreparedFrames, VIOFilter, load_eqf_lib  # noqa: E402


def kernel_times(flt, lib, prepared, start, count):
    core = flt.core_handle()
    lib.eqf_set_option(core, OPT_TIMING, 1)
    flt.run_prepared(prepared, start, count)
    which, us = np.zeros(1 << 18, np.int32), np.zeros(1 << 18, np.float32)
    cnt = lib.eqf_last_kernel_times(core, which.ctypes.data_as(C.POINTER(C.c_int)), us.ctypes.data_as(C.POINTER(C.c_float)), len(us))
    lib.eqf_set_option(core, OPT_TIMING, 0)
    agg = {}
    for i in range(cnt):
        agg.setdefault(lib.eqf_kernel_name(int(which[i])).decode(), []).append(float(us[i]))
    return {k: round(sum(v) / count, 2) for k, v in agg.items()}


def run(N, f32, frames, world, settings, n_warm, n_time, lib):
    flt = bench.make_filter(world, settings, N, 0, frames, lambda s, sensor, ids, p, t: VIOFilter(s, max_landmarks=N, sensor=sensor, ids=ids, p=p, time=t))
    flt.set_core_option(OPT_SYRK_F32, int(f32))
    prepared = PreparedFrames(world.cam, *bench.flatten_frames(frames[: n_warm + n_time + 20]))
    flt.run_prepared(prepared, 0, n_warm)
    lib.eqf_synchronize(flt.core_handle())
    t0 = time.perf_counter()
    flt.run_prepared(prepared, n_warm, n_time)
    lib.eqf_synchronize(flt.core_handle())
    el = time.perf_counter() - t0
    S = flt.get_sigma()
    est = flt.state_estimate()
    per_kernel = kernel_times(flt, lib, prepared, n_warm + n_time, 20)
    flt.close()
    return dict(updates_per_s=n_time / el, us_per_frame=1e6 * el / n_time, per_kernel_us_per_frame=per_kernel), S, est


def main():
    lib = load_eqf_lib()
    out = {"what": __doc__.split("\n\n")[0].replace("\n", " "), "sizes": {}}
    for N, n_warm, n_time in ((200, 300, 3000), (500, 100, 600)):
        settings = bench.eurocish_settings()
        world, frames = bench.build_workload(seed=100, n_frames=n_warm + n_time + 24, N=N)
        a, Sa, ea = run(N, False, frames, world, settings, n_warm, n_time, lib)
        b, Sb, eb = run(N, True, frames, world, settings, n_warm, n_time, lib)
        dpos = float(np.linalg.norm(ea[0][10:13] - eb[0][10:13]))
        datt = float(2 * np.arcsin(min(1.0, np.linalg.norm(ea[0][6:10] - np.sign(np.dot(ea[0][6:10], eb[0][6:10])) * eb[0][6:10]) / 2)))
        out["sizes"]["N%d" % N] = {
            "frames": n_time, "f64_mfma": a, "f32_mfma_syrk": b, "speedup_frame": b["updates_per_s"] / a["updates_per_s"],
            "speedup_k_syrk_sub": a["per_kernel_us_per_frame"]["k_syrk_sub"] / b["per_kernel_us_per_frame"]["k_syrk_sub"],
            "deviation_after_%d_frames" % (n_warm + n_time): {"sigma_rel_fro": float(np.linalg.norm(Sa - Sb) / np.linalg.norm(Sa)), "position_m": dpos, "attitude_rad": datt,
                                                              "landmarks_rel": float(np.max(np.linalg.norm(ea[2] - eb[2], axis=1) / np.linalg.norm(ea[2], axis=1)))},
        }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
