"""Condense rocprofv3 output directories into the small text/CSV summaries kept under profiles/."""
import collections
import csv
import glob
import json
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)


def short(name):
    return name.replace("void nqa::", "").replace("nqa::", "").split("(")[0][:70]


for tag in ("trace256", "trace1080", "traceadists"):
    fs = glob.glob(f"{src}/{tag}/*/*_kernel_stats.csv")
    if not fs:
        continue
    rows = list(csv.DictReader(open(fs[0])))
    with open(f"{dst}/{tag}_kernel_stats.csv", "w") as f:
        f.write("kernel,calls,total_ms,avg_us,min_us,max_us,pct\n")
        for r in rows:
            if "nqa::" not in r["Name"]:
                continue
            f.write(f"\"{short(r['Name'])}\",{r['Calls']},{float(r['TotalDurationNs'])/1e6:.3f},"
                    f"{float(r['AverageNs'])/1e3:.2f},{float(r['MinNs'])/1e3:.2f},{float(r['MaxNs'])/1e3:.2f},{r['Percentage']}\n")
    b = f"{src}/bench{tag[5:]}.json"
    if os.path.exists(b):
        lines = [l for l in open(b).read().splitlines() if l.startswith("{")]
        if lines:
            open(f"{dst}/{tag}_bench_line.json", "w").write(lines[-1] + "\n")

# HBM traffic per kernel (1080p run): FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE reports half
# the bytes of wide coalesced reads on gfx950 (MI355X_MICROARCH.md, HBM section) -> doubled here.
tr = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.Counter())
dur = collections.defaultdict(float)
for tag, cname in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    for f in glob.glob(f"{src}/{tag}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "nqa::" in r["Kernel_Name"] and r["Counter_Name"] == cname:
                k = short(r["Kernel_Name"])
                tr[k][cname] += float(r["Counter_Value"])
                cnt[k][cname] += 1
                if cname == "FETCH_SIZE":
                    dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
with open(f"{dst}/hbm_traffic_1080p.csv", "w") as f:
    f.write("kernel,launches,avg_fetch_MB_corrected(x2),avg_write_MB,avg_us_under_pmc,GBps_under_pmc\n")
    for k in sorted(tr):
        n = max(cnt[k]["FETCH_SIZE"], 1)
        fe = 2 * tr[k]["FETCH_SIZE"] * 1024 / n / 1e6
        wr = tr[k]["WRITE_SIZE"] * 1024 / max(cnt[k]["WRITE_SIZE"], 1) / 1e6
        us = dur[k] / n
        f.write(f"\"{k}\",{n},{fe:.1f},{wr:.1f},{us:.1f},{(fe + wr) / max(us, 1e-9) * 1e3:.0f}\n")

agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for f in glob.glob(f"{src}/pmc_mfma/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "nqa::" in r["Kernel_Name"]:
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                n[k] += 1
with open(f"{dst}/mfma_util_256.csv", "w") as f:
    f.write("kernel,launches,MfmaUtil(= MFMA_BUSY / (GUI_ACTIVE/8 * 1024 SIMDs))\n")
    for k in sorted(agg):
        gui = agg[k]["GRBM_GUI_ACTIVE"] / 8
        if gui > 0:
            f.write(f"\"{k}\",{n[k]},{agg[k]['SQ_VALU_MFMA_BUSY_CYCLES'] / (gui * 1024):.3f}\n")
print("summaries in", dst)
for p in sorted(os.listdir(dst)):
    print("==", p)
    print(open(os.path.join(dst, p)).read()[:3000])
