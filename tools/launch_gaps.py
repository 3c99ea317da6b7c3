"""Gaps between consecutive dispatches of the advance kernel in a rocprofv3 --kernel-trace csv.  Usage: launch_gaps.py DIR"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "advance" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort()
durs = [(e - s) / 1e3 for s, e in rows]
gaps = [(rows[i + 1][0] - rows[i][1]) / 1e3 for i in range(len(rows) - 1)]
print("n", len(rows), "dur us", [round(d, 1) for d in durs[:12]])
print("gaps us", [round(g, 1) for g in gaps[:12]])
