#!/usr/bin/env python3
"""profiles/hbm_traffic.json from a collection's PMC passes (tools/collect_profiles.sh -> gpurun_out/<tag>_pmc_raw.txt).

    python tools/update_hbm_traffic.py gpurun_out/r02d_pmc_raw.txt r02d

Each line of the raw file: `<workload> <FETCH_SIZE|WRITE_SIZE> <kernel name ...> <counter> <value per dispatch> (<n> dispatches)` (pmc_sum.py).
Per the guide's HBM section (/opt/skills/guides/MI355X_MICROARCH.md): FETCH_SIZE and WRITE_SIZE are collected in SEPARATE passes
(3 + 2 TCC slots), both in KiB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced streaming reads -> doubled;
WRITE_SIZE is taken as reported (it matches the algorithmic byte count of this kernel's 16-byte stores to < 0.1 %).
traffic per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes, keyed `<workload>:<envs>:<steps per launch>` as bench.py looks it up."""
import json
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENVS = {"pmsm": 16384, "permexdc": 4096, "scim": 65536, "scim_constspeed": 65536}


def main():
    raw, tag = sys.argv[1], sys.argv[2]
    spl = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
    vals = {}
    for line in open(raw):
        m = re.match(r"(\w+) (FETCH_SIZE|WRITE_SIZE) .*?(FETCH_SIZE|WRITE_SIZE)\s+([0-9.]+)\s+\((\d+) dispatches\)", line.strip())
        if m and ("advance" in line or "dc_stream" in line):
            vals.setdefault(m.group(1), {})[m.group(2)] = float(m.group(4))
    path = os.path.join(REPO, "profiles", "hbm_traffic.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    for wl, v in vals.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            out[f"{wl}:{ENVS[wl]}:{spl}"] = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
            out.setdefault("_raw", {})[f"{tag}:{wl}"] = v
    out["_note"] = ("bytes per launch of the dominant advance kernel; (2 x FETCH_SIZE + WRITE_SIZE) x 1024, counters in KiB from separate rocprofv3 --pmc "
                    "passes, FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md (HBM section); latest collection: " + tag)
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if not k.startswith("_")}, indent=1))


if __name__ == "__main__":
    main()
