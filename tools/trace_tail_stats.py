"""Mean / min / max duration of the LAST `count` dispatches of a kernel in a rocprofv3 --kernel-trace CSV directory: the timed region
of a bench.py run (its settle and warm-up launches come first).    python tools/trace_tail_stats.py <dir> <kernel substring> <count>"""
import csv, glob, os, sys

d, sub, count = sys.argv[1], sys.argv[2], int(sys.argv[3])
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if sub in r.get("Kernel_Name", ""):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
tail = rows[-count:]
dur = [(e - s) / 1e3 for s, e, _ in tail]
alld = [(e - s) / 1e3 for s, e, _ in rows]
if not dur:
    sys.exit(f"no dispatch of a kernel matching {sub!r} under {d}")
print(f"kernel: {tail[-1][2][:140]}")
print(f"all {len(alld)} dispatches (settle + warm-up + timed): mean {sum(alld) / len(alld):.1f} us")
print(f"last {len(dur)} dispatches (the timed region): mean {sum(dur) / len(dur):.1f} us, min {min(dur):.1f}, max {max(dur):.1f}")
