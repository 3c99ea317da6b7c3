#!/usr/bin/env python3
"""tools/ab_rate_limiter.sh output -> markdown table (mean of the interleaved repeats, of the 8 TB/s, at 32768 / 65536 / 131072 envs).
    python tools/ab_rate_limiter_table.py profiles/<round>_pace_ab.txt"""
import collections
import sys

d = collections.OrderedDict()
mode = None
for line in open(sys.argv[1]):
    if line.startswith("=="):
        mode = {"0": "off", "open": "open", "closed": "on", "default": "on"}[line.split()[-1]]
        continue
    p = [x.strip() for x in line.split("|")]
    if mode and len(p) >= 3 and p[1].isdigit():
        d.setdefault(p[0], {}).setdefault(int(p[1]), {}).setdefault(mode, []).append(float(p[2]))
sizes = sorted({n for r in d.values() for n in r})
has_open = any("open" in v for r in d.values() for v in r.values())
print("| row | limiter off (" + " / ".join(map(str, sizes)) + " envs) |" + (" open loop (built-in targets) |" if has_open else "") + " closed loop (default) | closed - off |")
print("|---|---|---|---|" + ("---|" if has_open else ""))
worst = 0.0
for lab, r in d.items():
    avg = lambda n, mode: sum(r[n][mode]) / len(r[n][mode]) if mode in r[n] else float("nan")  # noqa: E731
    m = lambda mode: " / ".join(f"{avg(n, mode):.2f}" for n in sizes)  # noqa: E731
    delta = [avg(n, "on") - avg(n, "off") for n in sizes]
    worst = min(worst, min(delta))
    print(f"| {lab} | {m('off')} |" + (f" {m('open')} |" if has_open else "") + f" {m('on')} | " + " / ".join(f"{x:+.2f}" for x in delta) + " |")
print(f"\nworst closed-loop row against its unpaced figure: {worst:+.3f} of the roofline")
