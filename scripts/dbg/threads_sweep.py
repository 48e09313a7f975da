"""R filters as R threads of one process at N = 200 (bench.several_filters_on_one_gpu's sweep alone). usage: python scripts/dbg/threads_sweep.py [counts, e.g. 3,4,6] [N]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from eqvio_amd.capi import VIOFilter, load_eqf_lib
counts = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "3,4,6").split(","))
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
lib = load_eqf_lib()
import unittest.mock as um
with um.patch("subprocess.run", side_effect=RuntimeError("skipped")):
    r = bench.several_filters_on_one_gpu(bench.eurocish_settings(), N, 0, VIOFilter, lib, counts=counts, sizes=(N,), n_frames=600, n_warm=100)
for row in r["sweep"]["N%d" % N]:
    print(row)
