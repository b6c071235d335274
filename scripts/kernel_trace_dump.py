#!/usr/bin/env python3
"""Print start/duration of the last N kernel launches in a rocprofv3 kernel_trace.csv (gaps included)."""
import csv, glob, sys
d, n = sys.argv[1], int(sys.argv[2])
rows = list(csv.DictReader(open(glob.glob(d + "/*/*kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev_end = None
for r in rows[-n:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print("%-56s dur %8.1f us  gap %7.1f us" % (r["Kernel_Name"][:56], (e - s) / 1e3, gap))
    prev_end = e
