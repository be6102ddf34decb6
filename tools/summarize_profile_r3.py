"""Condense the rocprofv3 output of tools/profile_r3.sh / profile_r4.sh into the small summaries kept under profiles/
(rNN_*), and into traffic.json (with the sha256 of the profiled library's sources), from which bench.py fills
`roofline.traffic`.  usage: python tools/summarize_profile_r3.py <rocprof output dir> <summary dir>"""
import collections
import csv
import glob
import json
import os
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONV = ("conv3x3_igemm_kernel", "conv3x3_regw_kernel", "conv3x3_regw128_kernel", "conv1_regw_kernel", "conv1_fused_kernel",
        "conv1_tile_kernel", "conv1_regw_split_kernel", "conv3x3_regw128_pool_kernel", "conv1_pool_kernel", "conv3x3_regw_split_kernel")  # (substring match)


def demangle(name):
    """rocprofv3 leaves some template instances mangled (_ZN3nqa<len><name>I<args>E...: binutils' c++filt does not know
    the DF16_ in their signatures); kernel name and integer / bool template arguments are all that is needed here."""
    m = re.match(r"_ZN3nqa(\d+)", name)
    if not m:
        return name
    n = int(m.group(1))
    base, rest = name[m.end():m.end() + n], name[m.end() + n:]
    args = []
    if rest.startswith("I"):
        for kind, val in re.findall(r"L([ib])(\d+)E", rest[1:rest.index("EE") + 1] if "EE" in rest else ""):
            args.append(("true" if val == "1" else "false") if kind == "b" else val)
    return "nqa::" + base + ("<" + ", ".join(args) + ">" if args else "") + "("


def ours(name):
    return "nqa::" in name or name.startswith("_ZN3nqa")


def short(name):
    return demangle(name).replace("void nqa::", "").replace("nqa::", "").split("(")[0][:100]


def klass(name):
    if any(c in name for c in CONV):
        return "conv"
    if "pool_stats_kernel" in name:
        return "pool"
    if "pool_seam_kernel" in name:
        return "seam"
    return None


for d in sorted(glob.glob(f"{src}/trace_*")):
    if not os.path.isdir(d):
        continue
    tag = os.path.basename(d)[len("trace_"):]
    fs = glob.glob(f"{d}/*/*_kernel_stats.csv")
    if not fs:
        continue
    rows = list(csv.DictReader(open(fs[0])))
    with open(f"{dst}/trace_{tag}_kernel_stats.csv", "w") as f:
        f.write("kernel,calls,total_ms,avg_us,min_us,max_us,pct\n")
        tot = collections.defaultdict(lambda: [0, 0.0])
        for r in rows:
            if not ours(r["Name"]):
                continue
            f.write(f"\"{short(r['Name'])}\",{r['Calls']},{float(r['TotalDurationNs'])/1e6:.3f},"
                    f"{float(r['AverageNs'])/1e3:.2f},{float(r['MinNs'])/1e3:.2f},{float(r['MaxNs'])/1e3:.2f},{r['Percentage']}\n")
            k = klass(demangle(r["Name"]))
            if k:
                tot[k][0] += int(r["Calls"])
                tot[k][1] += float(r["TotalDurationNs"]) / 1e6
        for k, (n, ms) in tot.items():
            f.write(f"\"== class {k}: all launches\",{n},{ms:.3f},{ms / max(n, 1) * 1e3:.2f},,,\n")
    b = f"{src}/bench_{tag}.json"
    if os.path.exists(b):
        lines = [ln for ln in open(b).read().splitlines() if ln.startswith("{")]
        if lines:
            open(f"{dst}/trace_{tag}_bench_line.json", "w").write(lines[-1] + "\n")

# HBM traffic per kernel: FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE reports half the bytes of wide coalesced
# reads on gfx950 (MI355X_MICROARCH.md, HBM section) -> doubled here.  Per launch, averaged over the run's launches.
traffic = {}
for d in sorted(glob.glob(f"{src}/pmc_FETCH_SIZE_*")):
    if not os.path.isdir(d):
        continue
    wl, pr = os.path.basename(d)[len("pmc_FETCH_SIZE_"):].rsplit("_", 1)
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(collections.Counter)
    for cname in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(f"{src}/pmc_{cname}_{wl}_{pr}/*/*_counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if ours(r["Kernel_Name"]) and r["Counter_Name"] == cname:
                    k = short(r["Kernel_Name"])
                    per[k][cname] += float(r["Counter_Value"])
                    cnt[k][cname] += 1
    if not per:
        continue
    cls = collections.defaultdict(lambda: [0.0, 0])
    with open(f"{dst}/hbm_traffic_{wl}_{pr}.csv", "w") as f:
        f.write("kernel,launches,avg_fetch_MB_corrected(x2),avg_write_MB\n")
        for k in sorted(per):
            n = max(cnt[k]["FETCH_SIZE"], 1)
            fe = 2 * per[k]["FETCH_SIZE"] * 1024 / n
            wr = per[k]["WRITE_SIZE"] * 1024 / max(cnt[k]["WRITE_SIZE"], 1)
            f.write(f"\"{k}\",{n},{fe / 1e6:.1f},{wr / 1e6:.1f}\n")
            c = klass(k)
            if c:
                cls[c][0] += (fe + wr) * n
                cls[c][1] += n
    traffic[f"{wl}/{pr}"] = {c: round(v[0] / max(v[1], 1)) for c, v in cls.items()}
    traffic[f"{wl}/{pr}"]["unit"] = "HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, averaged over the class's launches)"
sys.path.insert(0, ROOT)
from nerf_qa_amd import build as nqa_build  # noqa: E402
traffic["lib_sha16"] = nqa_build.source_hash()  # the HIP sources + flags the profiled library was built from
json.dump(traffic, open(f"{dst}/traffic.json", "w"), indent=1)

for d in sorted(glob.glob(f"{src}/pmc_mfma_*")):
    if not os.path.isdir(d):
        continue
    tag = os.path.basename(d)[len("pmc_mfma_"):]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for f in glob.glob(f"{d}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if ours(r["Kernel_Name"]):
                k = short(r["Kernel_Name"])
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                    n[k] += 1
    with open(f"{dst}/mfma_util_{tag}.csv", "w") as f:
        f.write("kernel,launches,MfmaUtil(= MFMA_BUSY / (GUI_ACTIVE/8 * 1024 SIMDs))\n")
        for k in sorted(agg):
            gui = agg[k]["GRBM_GUI_ACTIVE"] / 8
            if gui > 0 and agg[k]["SQ_VALU_MFMA_BUSY_CYCLES"] > 0:
                f.write(f"\"{k}\",{n[k]},{agg[k]['SQ_VALU_MFMA_BUSY_CYCLES'] / (gui * 1024):.3f}\n")
print("summaries in", dst)
for p in sorted(os.listdir(dst)):
    print("==", p)
    print(open(os.path.join(dst, p)).read()[:1500])
