#!/usr/bin/env python3
"""Reads a rocprofv3 kernel-trace CSV and prints, for the last complete frame, every kernel's start/end relative to
the frame start plus how much of it overlapped another kernel (evidence for/against lane concurrency)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# frame boundary = the stem kernel
stems = [i for i, n in enumerate(names) if "k_stem" in n]
a, b = stems[-2], stems[-1]
t0 = int(rows[a]["Start_Timestamp"])
frame = rows[a:b]
busy = 0
for i, r in enumerate(frame):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    ov = 0
    for j, q in enumerate(frame):
        if i == j:
            continue
        s2, e2 = int(q["Start_Timestamp"]), int(q["End_Timestamp"])
        ov = max(ov, min(e, e2) - max(s, s2))
    short = r["Kernel_Name"].split("(")[0][-60:]
    print("%8.1f %8.1f  dur %7.1f us  overlap %6.1f us  q%s  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, max(ov, 0) / 1e3,
                                                                  r.get("Queue_Id", "?"), short))
print("frame span %.1f us" % ((int(frame[-1]["End_Timestamp"]) - t0) / 1e3))
