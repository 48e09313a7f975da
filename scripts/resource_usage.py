#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel instantiation in libeqf_hip.so: compiles eqvio_amd/csrc/eqf_hip.hip with
-Rpass-analysis=kernel-resource-usage (no GPU needed) and prints one line per kernel, demangled. `python scripts/resource_usage.py > profiles/rNN_resource_usage.txt`."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "eqvio_amd", "csrc")
with tempfile.TemporaryDirectory() as td:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I../../include", "-Wno-unused", "-c", "-o", os.path.join(td, "x.o"),
                        "eqf_hip.hip", "-Rpass-analysis=kernel-resource-usage"], cwd=src, capture_output=True, text=True)
    if r.returncode:
        sys.stderr.write(r.stderr[-4000:])
        sys.exit(r.returncode)
rows, cur = [], None
for line in r.stderr.splitlines():
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|SGPRs Spill|LDS Size \[bytes/block\]|TotalSGPRs):\s+(\S+)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
names = subprocess.run(["c++filt"], input="\n".join(x["name"] for x in rows), capture_output=True, text=True).stdout.splitlines()
print("%-6s %-6s %-8s %-6s %-7s %-8s  kernel" % ("VGPR", "AGPR", "scratch", "occ", "spillV", "LDS"))
for x, n in sorted(zip(rows, names), key=lambda t: t[1]):
    n = re.sub(r"\(.*$", "", n)  # the argument list says nothing here
    print("%-6s %-6s %-8s %-6s %-7s %-8s  %s" % (x.get("VGPRs"), x.get("AGPRs"), x.get("ScratchSize [bytes/lane]"), x.get("Occupancy [waves/SIMD]"), x.get("VGPRs Spill"),
                                                 x.get("LDS Size [bytes/block]"), n))
