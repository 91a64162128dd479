#!/usr/bin/env python3
"""Turn the four rocprofv3 --pmc passes of tools/pmc_shape.sh (sq_insts, sq_wait, fetch, write: one counter group per
pass, --kernel-trace the only trace domain) into profiles/counters.json, the per-kernel figures bench.py quotes next to
its roofline: HBM bytes per launch, the share of the SIMDs' issue slots taken, the share of wave time spent waiting, the
matrix pipes' busy share -- stamped with the commit they were taken at.

    python tools/collect_counters.py <workload> <pmc dir> [out.json]

Bytes per MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports half of the bytes
of a wide coalesced streaming read, so it is doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.
issue_frac = (vector + LDS + memory instructions) / (1024 SIMDs x clocks / 4): a SIMD issues one such instruction of a wave
per 4 cycles at best.  wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES.  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 x clocks)."""
import collections
import csv
import glob
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def short(name):
    return name.split("(")[0].replace("void ", "").split("::")[-1].split("<")[0]  # (instantiations of one kernel together)


def counters(root, sub):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(str(root / sub / "**" / "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "dmf::" in row["Kernel_Name"]:
                out[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in out.items()}


def durations(root, sub):
    out = collections.defaultdict(list)
    for f in glob.glob(str(root / sub / "**" / "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "dmf::" in row["Kernel_Name"]:
                out[short(row["Kernel_Name"])].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    return {k: (sum(v) / len(v), len(v)) for k, v in out.items()}


def main():
    workload, root = sys.argv[1], Path(sys.argv[2])
    out_path = Path(sys.argv[3]) if len(sys.argv) > 3 else ROOT / "profiles" / "counters.json"
    insts, wait = counters(root, "sq_insts"), counters(root, "sq_wait")
    fetch, write = counters(root, "fetch"), counters(root, "write")
    dur = durations(root, "fetch")  # (the byte passes perturb the kernels least)
    table = json.loads(out_path.read_text()) if out_path.exists() else {}
    entry = table[workload] = {}  # kernels of older commits do not linger
    for k, (us, n) in sorted(dur.items(), key=lambda kv: -kv[1][0]):
        i, w = insts.get(k, {}), wait.get(k, {})
        gui_i, gui_w = i.get("GRBM_GUI_ACTIVE", 0.0) / 8, w.get("GRBM_GUI_ACTIVE", 0.0) / 8  # (summed over the 8 XCDs)
        n_issue = sum(i.get(c, 0.0) for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"))
        entry[k] = {
            "avg_us_under_profiler": us, "launches": n,
            "hbm_bytes_per_launch": (2.0 * fetch.get(k, {}).get("FETCH_SIZE", 0.0) + write.get(k, {}).get("WRITE_SIZE", 0.0)) * 1024.0,
            "FETCH_SIZE_KiB_raw": fetch.get(k, {}).get("FETCH_SIZE"), "WRITE_SIZE_KiB_raw": write.get(k, {}).get("WRITE_SIZE"),
            "issue_frac": n_issue / (1024 * gui_i / 4) if gui_i else None,
            "wait_frac": w.get("SQ_WAIT_ANY", 0.0) / w["SQ_WAVE_CYCLES"] if w.get("SQ_WAVE_CYCLES") else None,
            "mfma_busy_frac": w.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024 * gui_w) if gui_w else None,
            "instructions": {c: i.get(c) for c in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU") if c in i},
        }
    head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=ROOT).stdout.strip()
    table["_commit"] = head or "unknown"
    table["_how"] = "tools/pmc_shape.sh <dir> N S n_c n_u T1 (four separate rocprofv3 --kernel-trace --pmc passes of tools/one_shape.py), then tools/collect_counters.py"
    out_path.write_text(json.dumps(table, indent=1, sort_keys=True) + "\n")
    for k, v in entry.items():
        if v["avg_us_under_profiler"] >= 5:
            print(f"{k[:34]:34s} {v['avg_us_under_profiler']:8.1f} us  {v['hbm_bytes_per_launch'] / 1e9:7.3f} GB  issue {v['issue_frac']}  wait {v['wait_frac']}  mfma {v['mfma_busy_frac']}")


if __name__ == "__main__":
    main()
