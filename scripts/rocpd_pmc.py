"""Per-kernel average of a PMC counter from a rocprofv3 rocpd database. Usage: rocpd_pmc.py results.db"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(rocpd_pmc_event)")]
pcols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_pmc)")]
rows = cur.execute(
    """select s.kernel_name, p.name, count(*), avg(e.value), min(e.value), max(e.value)
       from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id
       join rocpd_kernel_dispatch d on e.event_id = d.event_id
       join rocpd_info_kernel_symbol s on d.kernel_id = s.id
       group by s.kernel_name, p.name order by 4 desc"""
).fetchall()
print("Kernel,Counter,Dispatches,Avg,Min,Max")
for name, pn, n, avg, mn, mx in rows:
    print(f'"{name.split("(")[0]}",{pn},{n},{avg:.1f},{mn},{mx}')
