"""How the kernels of several filters on ONE GPU overlap, from a rocprofv3 kernel trace of scripts/multi_filter.py (rocpd sqlite database):
per kernel family the average duration and - for the persistent look-ahead factorisation - how many of them were in flight at once (time-weighted),
and how much of the wall time had k look-ahead kernels running. usage: python scripts/multi_filter_overlap.py results.db"""
import re, sqlite3, sys
con = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in con.execute("pragma table_info(rocpd_kernel_dispatch)")]
qcol = "queue_id" if "queue_id" in cols else None
rows = con.execute(f"select d.start, d.end, s.kernel_name{', d.' + qcol if qcol else ''} from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
def fam(n):
    m = re.search(r"\d+(k_[a-z_]+?)(?:I[A-Za-z]|E[a-z])", n) or re.search(r"(k_[a-z_]+)", n)
    return m.group(1) if m else n[:20]
# steady state: the last 60 % of the trace
t_lo = rows[0][0] + 0.4 * (rows[-1][1] - rows[0][0])
rows = [r for r in rows if r[0] >= t_lo]
span = rows[-1][1] - rows[0][0]
by = {}
for r in rows:
    by.setdefault(fam(r[2]), []).append(r[1] - r[0])
print(f"steady-state window {span * 1e-6:.2f} ms, {len(rows)} dispatches" + (f", {len(set(r[3] for r in rows))} hardware queues in use" if qcol else ""))
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {k:22s} {len(v):6d} launches  avg {sum(v) / len(v) * 1e-3:8.2f} us   busy {sum(v) / span:5.2f} x wall")
def in_flight(iv, label):
    ev = sorted([(s, 1) for s, e in iv] + [(e, -1) for s, e in iv])
    hist, cur, prev = {}, 0, ev[0][0]
    for t, d in ev:
        hist[cur] = hist.get(cur, 0) + (t - prev)
        cur += d
        prev = t
    tot = sum(hist.values())
    print(label + " in flight (share of the window): " + ", ".join(f"{k}: {v / tot:.2f}" for k, v in sorted(hist.items())) + f"   mean {sum(k * v for k, v in hist.items()) / tot:.2f}")
in_flight([(r[0], r[1]) for r in rows if fam(r[2]) == "k_chol_lookahead"], "look-ahead kernels")
in_flight([(r[0], r[1]) for r in rows], "kernels of any kind")
if qcol:
    for q in sorted(set(r[3] for r in rows)):
        mine = [r for r in rows if r[3] == q]
        print(f"  queue {q}: {len(mine)} dispatches, busy {sum(r[1] - r[0] for r in mine) / span:.2f} x wall")
