"""Effective shader clock and matrix-pipe utilisation per conv kernel from the pmc_mfma_* passes of tools/profile_r3.sh
(rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE): per dispatch, clock = GRBM_GUI_ACTIVE / 8
XCDs / (End - Start), MfmaUtil = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024 SIMDs); medians over the launches of >= 0.3 ms.
usage: python tools/effective_clock_r3.py gpurun_out/prof_r3 > profiles/r03_effective_clock.txt"""
import collections
import csv
import glob
import os
import statistics
import sys

src = sys.argv[1]
import re

CONV = ("conv3x3_igemm_kernel", "conv3x3_regw_kernel", "conv3x3_regw128_kernel", "conv1_regw_kernel",
        "conv1_regw_split_kernel", "conv3x3_regw128_pool_kernel", "conv1_pool_kernel", "conv3x3_regw_split_kernel")  # (substring match; most specific last)


def demangle(name):  # (see summarize_profile_r3.py: rocprofv3 leaves some template instances mangled)
    m = re.match(r"_ZN3nqa(\d+)", name)
    if not m:
        return name
    n = int(m.group(1))
    base, rest = name[m.end():m.end() + n], name[m.end() + n:]
    args = [("true" if v == "1" else "false") if k == "b" else v
            for k, v in re.findall(r"L([ib])(\d+)E", rest[1:rest.index("EE") + 1] if rest.startswith("I") and "EE" in rest else "")]
    return "nqa::" + base + ("<" + ", ".join(args) + ">" if args else "") + "("


def short(name):
    return demangle(name).replace("void nqa::", "").replace("nqa::", "").split("(")[0][:100]


print("Effective shader clock and matrix-pipe utilisation per conv kernel, B=8 1080p step (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES\n"
      "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE of `bench.py --workload <w> --precision <p> --only`; launches of >= 0.3 ms only; medians over\n"
      "the launches).  clock = GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration (MI355X_MICROARCH.md, DVFS give-back);\n"
      "MfmaUtil = MFMA_BUSY / (GUI_ACTIVE/8 * 1024 SIMDs).  Nominal peak 2.5 PFLOP/s assumes 2.4 GHz.\n")
for d in sorted(glob.glob(f"{src}/pmc_mfma_*")):
    if not os.path.isdir(d):
        continue
    disp = collections.defaultdict(dict)
    for f in glob.glob(f"{d}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if any(c in r["Kernel_Name"] for c in CONV):
                key = (f, r["Dispatch_Id"])
                disp[key]["k"] = short(r["Kernel_Name"])
                disp[key][r["Counter_Name"]] = float(r["Counter_Value"])
                disp[key]["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    per = collections.defaultdict(list)
    for v in disp.values():
        if v.get("ns", 0) >= 300_000 and v.get("GRBM_GUI_ACTIVE", 0) > 0:
            gui = v["GRBM_GUI_ACTIVE"] / 8
            per[v["k"]].append((gui / v["ns"], v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * 1024)))
    print(os.path.basename(d)[len("pmc_mfma_"):])
    for k in sorted(per, key=lambda k: (max(i for i, c in enumerate(CONV) if c in k), k)):
        clk = statistics.median(c for c, _ in per[k])
        util = statistics.median(u for _, u in per[k])
        print(f"  {k:<62} launches {len(per[k]):3d}  clock {clk:.2f} GHz  MfmaUtil {util:.3f}  "
              f"util x clock / 2.4 GHz = {util * clk / 2.4:.3f} of nominal")
