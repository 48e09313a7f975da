"""profiles/<round>_pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of scripts/collect_profiles.sh (one tag per size).
usage: python scripts/pmc_traffic_json.py r03 N200=r03_v1 N500=r03_v1_N500 N50=r03_v1_N50   (tags under profiles/)"""
import csv, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
ROUND = sys.argv[1]
for arg in sys.argv[2:]:
    key, tag = arg.split("=")
    kern = {}
    for which, fn in (("fetch", f"{tag}_pmc_fetch_size.csv"), ("write", f"{tag}_pmc_write_size.csv")):
        agg = {}
        for row in csv.DictReader(open(os.path.join(ROOT, "profiles", fn))):
            m = re.match(r"_ZN3eqf\d+(k_[a-z_A-Z0-9]+?)(I[A-Za-z0-9_]*)?E", row["Kernel"])
            name = m.group(1) if m else row["Kernel"]
            name = re.sub(r"I[a-zA-Z]?L?[bi]?\d.*$", "", name)
            a = agg.setdefault(name, [0.0, 0])
            a[0] += float(row["Avg"]) * int(row["Dispatches"])
            a[1] += int(row["Dispatches"])
        for name, (tot, cnt) in agg.items():
            kern.setdefault(name, {})[f"{which}_kib_per_launch"] = round(tot / max(cnt, 1), 1)
    kern = {k: v for k, v in kern.items() if k.startswith("k_") and "fetch_kib_per_launch" in v and "write_kib_per_launch" in v}
    out[key] = {"source": f"profiles/{tag}_pmc_fetch_size.csv, _pmc_write_size.csv (rocprofv3 --kernel-trace --pmc, separate passes; KiB per launch as reported; template "
                          "instantiations of one kernel averaged by dispatch count)", "kernels": kern}
json.dump(out, open(os.path.join(ROOT, "profiles", ROUND + "_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
