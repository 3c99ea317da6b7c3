#!/usr/bin/env python3
"""tools/ab_rate_limiter.sh output -> markdown table (mean of the interleaved repeats, of the 8 TB/s, at 32768 / 65536 / 131072 envs).
    python tools/ab_rate_limiter_table.py profiles/<round>_pace_ab.txt"""
import collections
import sys

d = collections.OrderedDict()
mode = None
for line in open(sys.argv[1]):
    if line.startswith("=="):
        mode = "off" if line.split()[-1] == "0" else "on"
        continue
    p = [x.strip() for x in line.split("|")]
    if mode and len(p) >= 3 and p[1].isdigit():
        d.setdefault(p[0], {}).setdefault(int(p[1]), {}).setdefault(mode, []).append(float(p[2]))
sizes = sorted({n for r in d.values() for n in r})
print("| row | limiter off (" + " / ".join(map(str, sizes)) + " envs) | limiter on |")
print("|---|---|---|")
for lab, r in d.items():
    m = lambda mode: " / ".join(f"{sum(r[n][mode]) / len(r[n][mode]):.2f}" for n in sizes)  # noqa: E731
    print(f"| {lab} | {m('off')} | {m('on')} |")
