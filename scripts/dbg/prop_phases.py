"""Phase stamps of the propagation kernel's workgroups (debug build with -DEQF_DBG_PROP in scripts/ab_libs_prof). usage: EQVIO_AMD_LIB_DIR=$PWD/scripts/ab_libs_prof python scripts/dbg/prop_phases.py [N]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from eqvio_amd.capi import VIOFilter, load_eqf_lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
lib = load_eqf_lib()
world, frames = bench.build_workload(seed=100, n_frames=400, N=N)
flt = bench.make_filter(world, bench.eurocish_settings(), N, 0, frames, lambda s, se, i, p, t: VIOFilter(s, max_landmarks=N, device=0, sensor=se, ids=i, p=p, time=t))
core = flt.core_handle()
flt.run_frames(world.cam, *bench.flatten_frames(frames[:300]))
out = np.zeros(8192, np.uint64)
lib.eqf_debug_prop_stamps.argtypes = [C.c_void_p, C.c_void_p]
nT = (N + 7) // 8
res = []; obs_chain = []
for rep in range(20):
    flt.run_frames(world.cam, *bench.flatten_frames(frames[300 + rep:301 + rep]))
    assert lib.eqf_debug_prop_stamps(core, out.ctypes.data) == 0
    d = out.reshape(1024, 8).astype(np.float64) * 0.01
    nz = int(np.count_nonzero(d[:, 0]))  # workgroups of the launch
    nt = nz - 3  # + sensor block, observer block, staging block
    t0 = d[:nz, 0].min()
    tiles = d[:nt] - t0
    obs = d[nt + 1:nt + 3] - t0
    obs_chain.append(d[nt + 1, 7] - d[nt + 1, 6])
    res.append([np.median(tiles[:, 6] - tiles[:, 0]), np.median(tiles[:, 7] - tiles[:, 6]), np.median(tiles[:, 1] - tiles[:, 7]), tiles[:, 0].max(), np.median(tiles[:, 1] - tiles[:, 0]), np.median(tiles[:, 2] - tiles[:, 1]), np.median(tiles[:, 3] - tiles[:, 2]), np.median(tiles[:, 4] - tiles[:, 3]),
                tiles[:, 4].max(), np.median(tiles[:, 4] - tiles[:, 0]), obs[0, 0], obs[0, 5], d[nt, 0] - t0, d[nt, 5] - t0, d[nz - 1, 0] - t0, d[nz - 1, 5] - t0])
r = np.median(np.array(res[3:]), axis=0)
print(f"  start -> kernarg terms in LDS {r[0]:.2f}, -> wave 0's Sigma loads arrived {r[1]:.2f}, -> assembled + barrier {r[2]:.2f}")
r = r[3:]
print(f"N={N}: {nt} tile workgroups of {nz}. last tile workgroup starts {r[0]:.2f} us after the first; per workgroup (medians): load+assemble {r[1]:.2f}, G {r[2]:.2f}, strips {r[3]:.2f}, pairs+stores {r[4]:.2f}, total {r[6]:.2f}")
print(f"  observer workgroup: chain only {np.median([x for x in obs_chain]):.2f} us")
print(f"  last tile workgroup done at {r[5]:.2f} us; observer workgroup 0 starts at {r[7]:.2f}, done at {r[8]:.2f}; sensor block starts {r[9]:.2f}, done {r[10]:.2f}; staging block starts {r[11]:.2f}, done {r[12]:.2f}")
