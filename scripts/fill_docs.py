"""Fill the @PLACEHOLDERS@ of DESIGN.md / README.md from a collected profile set: python scripts/fill_docs.py r06_v2 (works on copies kept as *.tmpl so that it can be re-run)."""
import csv, json, os, sys
tag = sys.argv[1]
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.loads(open(f"{R}/profiles/{tag}_full_bench.json").read().strip().splitlines()[-1])
la = None
for row in csv.DictReader(open(f"{R}/profiles/{tag}_kernel_stats.csv")):
    if "k_chol_lookaheadILi4ELi3ELb1" in row["Name"]:
        la = float(row["AverageNs"]) / 1e3
r = d["roofline"]
flops = 2.2005e8
vals = {"HEAD": f"{d['value']:,.0f}".replace(",", " "), "HEADUS": f"{1e3 * d['ms_per_step']:.1f}", "LAROC": f"{la:.1f}", "LATF": f"{flops / la / 1e6:.2f}", "LAFRAC": f"{flops / la / 1e6 / 78.6:.4f}",
        "LASPAN": f"{r['avg_launch_us']:.1f}", "BENCHFRAC": f"{r['frac']:.4f}", "MFMABUSY": f"{100 * r.get('mfma_busy_frac', 0):.1f}", "DENSE": f"{2.004e9 * d['value'] / 1e12:.1f}",
        "HBMGBPS": f"{r.get('hbm_gbps', 0):.0f}"}
for f in ("DESIGN.md", "README.md"):
    tm = f"{R}/{f}.tmpl"
    if not os.path.exists(tm):
        open(tm, "w").write(open(f"{R}/{f}").read())
    s = open(tm).read().replace("r06_v2", tag)
    for k, v in vals.items():
        s = s.replace("@" + k + "@", v)
    open(f"{R}/{f}", "w").write(s)
print(vals)
