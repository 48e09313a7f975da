"""Steady-state frame from a rocprofv3 (rocpd sqlite) kernel trace: average start offset, duration and gap in front of every dispatch of a frame
(a frame starts at k_propagate_main). usage: python scripts/rocpd_frames.py results.db [frames from the end, default 200]"""
import collections, re, sqlite3, sys
con = sqlite3.connect(sys.argv[1])
nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rows = con.execute("select d.start, d.end, s.kernel_name from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
names = [re.sub(r"^_ZN3eqf\d+", "", r[2].split("(")[0]).split("I")[0].split("E")[0] if r[2].startswith("_ZN3eqf") else r[2][:24] for r in rows]
starts = [i for i, n in enumerate(names) if n.startswith("k_propagate_main")]
starts = starts[-nlast - 1:]
shape = collections.Counter(tuple(names[a:b]) for a, b in zip(starts, starts[1:])).most_common(1)[0][0]
acc = [[0.0, 0.0, 0.0] for _ in shape]
n = 0
period = 0.0
for a, b in zip(starts, starts[1:]):
    if tuple(names[a:b]) != shape:
        continue
    n += 1
    period += rows[b][0] - rows[a][0]
    for k, i in enumerate(range(a, b)):
        acc[k][0] += rows[i][0] - rows[a][0]
        acc[k][1] += rows[i][1] - rows[i][0]
        acc[k][2] += rows[i][0] - rows[i - 1][1]
print(f"{n} frames of {len(shape)} dispatches, period {1e-3 * period / n:.2f} us")
for k, nm in enumerate(shape):
    print(f"  {1e-3 * acc[k][0] / n:8.2f}  {nm:22s} dur {1e-3 * acc[k][1] / n:7.2f}  gap in front {1e-3 * acc[k][2] / n:6.2f}")
