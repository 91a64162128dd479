#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace of tools/two_stream_restarts.py: how much of the time do kernels of the two streams
actually run AT THE SAME TIME?   python tools/stream_overlap.py <dir with *kernel_trace.csv>"""
import csv
import glob
import sys
from collections import defaultdict

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "dmf::" in r["Kernel_Name"]:
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").split("::")[-1].split("<")[0]
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "?")))
rows.sort()
busy = union = 0
cur_end = None
pair = defaultdict(int)
active = []  # (end, name)
for s, e, name, q in rows:
    busy += e - s
    active = [(ae, an) for ae, an in active if ae > s]
    for ae, an in active:
        pair[tuple(sorted((name, an)))] += min(ae, e) - s
    active.append((e, name))
    if cur_end is None or s >= cur_end:
        union += e - s
        cur_end = e
    elif e > cur_end:
        union += e - cur_end
        cur_end = e
span = rows[-1][1] - rows[0][0]
print(f"{len(rows)} kernel launches over {span / 1e6:.1f} ms: sum of kernel durations {busy / 1e6:.1f} ms, time with at least one kernel "
      f"running {union / 1e6:.1f} ms ({100 * union / span:.1f} % of the span), time in which two ran at once {(busy - union) / 1e6:.1f} ms "
      f"({100 * (busy - union) / busy:.1f} % of the kernel time)")
for (a, b), ns in sorted(pair.items(), key=lambda kv: -kv[1])[:8]:
    print(f"   {a} with {b}: {ns / 1e6:.2f} ms")
