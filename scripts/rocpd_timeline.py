"""Timeline of one steady-state frame from a rocprofv3 (rocpd sqlite) kernel trace: kernel durations and the gaps
between consecutive dispatches. Usage: python scripts/rocpd_timeline.py results.db [frame_index_from_end]"""
import re
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = con.execute(
    """select d.start, d.end, s.kernel_name from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"""
).fetchall()
names = [re.sub(r"^_ZN3eqf\d+", "", r[2].split("(")[0]).split("E")[0] if r[2].startswith("_ZN3eqf") else r[2][:24] for r in rows]
# a frame starts at k_assemble_AB
starts = [i for i, n in enumerate(names) if n.startswith("k_assemble_AB")]
a, b = starts[-back - 1], starts[-back]
t0 = rows[a][0]
prev_end = None
busy = 0
print(f"frame of {b - a} dispatches, {1e-3 * (rows[b][0] - t0):.1f} us from its first kernel to the next frame's first kernel")
for i in range(a, b):
    s, e, _ = rows[i]
    gap = (s - prev_end) if prev_end is not None else 0
    busy += e - s
    print(f"{1e-3 * (s - t0):8.1f}  {names[i]:20s} dur {1e-3 * (e - s):6.2f}  gap before {1e-3 * gap:6.2f}")
    prev_end = max(prev_end or 0, e)
print(f"kernel time {1e-3 * busy:.1f} us; tail gap to next frame {1e-3 * (rows[b][0] - prev_end):.2f} us")

# averages over the last frames with the same dispatch count (steady state): duration and gap per position
import collections
nf = 0
acc = collections.OrderedDict()
period = 0.0
for k in range(len(starts) - 2, max(len(starts) - 40, 0), -1):
    a2, b2 = starts[k], starts[k + 1]
    if b2 - a2 != b - a:
        continue
    nf += 1
    period += rows[b2][0] - rows[a2][0]
    pe = rows[a2 - 1][1] if a2 > 0 else rows[a2][0]
    for i in range(a2, b2):
        s, e, _ = rows[i]
        key = (i - a2, names[i])
        d = acc.setdefault(key, [0.0, 0.0])
        d[0] += e - s
        d[1] += max(0, s - pe)
        pe = max(pe, e)
if nf:
    print(f"\naverage over {nf} frames: period {1e-3 * period / nf:.1f} us")
    tg = td = 0.0
    for (pos, nme), (d, g) in acc.items():
        print(f"  {pos:3d} {nme:20s} dur {1e-3 * d / nf:6.2f}  gap before {1e-3 * g / nf:6.2f}")
        td += d / nf
        tg += g / nf
    print(f"  kernel time {1e-3 * td:.1f} us, gaps {1e-3 * tg:.1f} us")
