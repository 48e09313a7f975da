import sys, os, glob
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from eqvio_amd.capi import EqfCore, EqfError
import test_golden as tg
for path in sorted(glob.glob("tests/golden/frame_*.npz")):
    for la, zb in ((0, 0), (1, 0), (1, 1)):
        d, cam, s = tg._load(path)
        N = int(d["N"])
        core = EqfCore(N, int(d["chart"]))
        core.set_option(12, la); core.set_option(17, zb)
        core.set_state(d["xi0"], d["Xs"], d["ids"], d["q0"], d["Q"]); core.set_sigma(d["Sigma0"])
        core.integrate_riccati_fast(d["imu"], float(d["dt"]), d["Qdiag"], d["Pdiag8"])
        core.integrate_observer(d["imus"], d["dts"], True)
        try:
            core.vision_update(cam, d["meas_ids"], d["meas_y"], float(d["meas_var"]), True, False)
            S = core.get_sigma()
            err = (np.linalg.norm(S - d["Sigma_updated"]) / np.linalg.norm(d["Sigma_updated"])) if "Sigma_updated" in d else abs(np.linalg.norm(S) - d["Sigma_updated_fro"]) / d["Sigma_updated_fro"]
            print(os.path.basename(path), "N", N, "M", len(d["meas_ids"]), "la", la, "zb", zb, "Sigma err", err)
        except EqfError as e:
            print(os.path.basename(path), "N", N, "M", len(d["meas_ids"]), "la", la, "zb", zb, "ERROR", e)
