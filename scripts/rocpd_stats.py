"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / avg / min / max durations.
Usage: python scripts/rocpd_stats.py results.db > profiles/<name>_kernel_stats.csv"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
rows = cur.execute(
    """select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start)
       from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
       group by s.kernel_name order by 3 desc"""
).fetchall()
tot = sum(r[2] for r in rows)
print("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs")
for name, calls, total, avg, mn, mx in rows:
    short = name.split("(")[0]
    print(f'"{short}",{calls},{total},{avg:.1f},{100.0 * total / tot:.2f},{mn},{mx}')
