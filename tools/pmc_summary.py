#!/usr/bin/env python3
"""Per-kernel summary of tools/pmc_shape.sh's passes (means per launch of the dmf:: kernels).
   python tools/pmc_summary.py <dir>"""
import collections, csv, glob, sys
from pathlib import Path

root = Path(sys.argv[1])


def short(name):
    return name.split("(")[0].replace("void ", "").split("::")[-1]


def counters(sub):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(str(root / sub / "**" / "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "dmf::" in row["Kernel_Name"]:
                out[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in out.items()}


def durations(sub):
    out = collections.defaultdict(list)
    for f in glob.glob(str(root / sub / "**" / "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "dmf::" in row["Kernel_Name"]:
                out[short(row["Kernel_Name"])].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    return {k: sum(v) / len(v) for k, v in out.items()}


insts, wait = counters("sq_insts"), counters("sq_wait")
fetch, write = counters("fetch"), counters("write")
dur = durations("fetch")  # (the byte passes perturb the kernels least)
print(f"{'kernel':34s} {'us':>7s} {'HBM GB':>7s} {'TB/s':>5s} {'VALU':>9s} {'MFMA':>8s} {'LDS':>8s} {'VMEM':>8s} {'issue %':>8s} {'wait %':>7s} {'mfma busy %':>11s}")
for k in sorted(dur, key=lambda k: -dur[k]):
    if dur[k] < 5:
        continue
    i, w = insts.get(k, {}), wait.get(k, {})
    hbm = (2 * fetch.get(k, {}).get("FETCH_SIZE", 0.0) + write.get(k, {}).get("WRITE_SIZE", 0.0)) * 1024
    gui = i.get("GRBM_GUI_ACTIVE", 0.0) / 8  # GPU clocks the launch was active (the counter is summed over the 8 XCDs)
    n_issue = sum(i.get(c, 0.0) for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"))
    # a SIMD issues one vector / LDS / memory instruction of a wave per 4 cycles at best: slots = 1024 SIMDs x clocks / 4
    issue = 100 * n_issue / (1024 * gui / 4) if gui else float("nan")
    wait_pct = 100 * w.get("SQ_WAIT_ANY", 0.0) / w["SQ_WAVE_CYCLES"] if w.get("SQ_WAVE_CYCLES") else float("nan")
    gui_w = w.get("GRBM_GUI_ACTIVE", 0.0) / 8
    mfma = 100 * w.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024 * gui_w) if gui_w else float("nan")
    print(f"{k[:34]:34s} {dur[k]:7.1f} {hbm / 1e9:7.3f} {hbm / dur[k] / 1e6:5.2f} {i.get('SQ_INSTS_VALU', 0):9.3g} {i.get('SQ_INSTS_MFMA', 0):8.3g} "
          f"{i.get('SQ_INSTS_LDS', 0):8.3g} {i.get('SQ_INSTS_VMEM_RD', 0) + i.get('SQ_INSTS_VMEM_WR', 0):8.3g} {issue:8.1f} {wait_pct:7.1f} {mfma:11.1f}")
