"""host scopes of the member-for-member binding leg (libraries built with -DEQF_HOST_PROFILE in scripts/ab_libs_prof, picked up through LD_LIBRARY_PATH)"""
import os, sys, tempfile, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, bench
from integration_scenario import build_driver, plan_without_decisions, write_scenario, EXE
import integration_scenario as isc
build_driver()
N, nfr = 200, 120
settings = bench.eurocish_settings()
world, frames = bench.build_workload(seed=100, n_frames=nfr + 1, N=N)
sensor, ids, p = world.true_state(0.0, frames[0][2])
with tempfile.TemporaryDirectory() as tmp:
    scen = os.path.join(tmp, "s.bin")
    write_scenario(scen, settings, world.cam, sensor, ids, p, 0.0, frames[:nfr], plan_without_decisions(frames[:nfr], 0.0))
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "scripts", "ab_libs_prof") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([EXE, scen, os.path.join(tmp, "o.bin"), "0", "0", "0", "20"], env=env, capture_output=True, text=True)
    print(r.stdout[-1500:]); print(r.stderr[-6000:])
