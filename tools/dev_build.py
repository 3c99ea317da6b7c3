#!/usr/bin/env python3
"""Variant build of SOME kernel units for same-box A/B runs and timing probes (round 6: one shared object per unit).

    python tools/dev_build.py --units 2_2_0,1_2_0 --defs GEMX_TIMING --out variants/timing [--only 0,1,0]

recompiles the named units (sys_conv_f64) with the extra macro definitions into <out>/libgemx_u*.so and links every OTHER unit of the
product build into <out> (symlinks), so that `GEMX_UNIT_DIR=$PWD/<out>` serves any handle: the named units from the variant, the rest from
the product.  Struct layouts (KArgs, gemx_handle) must be those of the product's libgemx.so (gemx_unit_init checks the handle size).
--only LOAD,SOLVER,IL compiles ONE (load kind, solver kind, dead time) combination of each named unit (GEMX_DEV_ONLY: seconds instead of
minutes; any other configuration of that unit then fails with "unsupported load/solver combination")."""
import argparse
import concurrent.futures as cf
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from gym_electric_motor_amd import build as b  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--units", required=True, help="comma-separated sys_conv_f64 units to recompile, e.g. 2_2_0,1_1_0")
    ap.add_argument("--defs", default="", help="comma-separated extra macro definitions, e.g. GEMX_TIMING,GEMX_PREP_Q=8")
    ap.add_argument("--out", required=True, help="variant directory (use it as GEMX_UNIT_DIR)")
    ap.add_argument("--only", default=None, help="LOAD,SOLVER,IL: compile this one combination only (0|1, 0|1|2, 0|1)")
    ap.add_argument("--no-compress", action="store_true")
    args = ap.parse_args()
    out = os.path.abspath(args.out)
    os.makedirs(out, exist_ok=True)
    hipcc = b.hipcc_path()
    defs = [d for d in args.defs.split(",") if d]
    if args.only:
        ld, sv, il = (int(x) for x in args.only.split(","))
        defs += ["GEMX_DEV_ONLY", f"GEMX_DEV_LOAD={ld}", f"GEMX_DEV_SOLVER={sv}", f"GEMX_DEV_IL={il}"]
    flags = [f for f in b.FLAGS if not (args.no_compress and f == "--offload-compress")]
    inc = ["-I" + os.path.join(REPO, "include"), "-I" + b.CSRC]
    cmds = []
    named = set()
    for u in [u for u in args.units.split(",") if u]:
        s, c, f = u.split("_")
        named.add(f"libgemx_u{s}_{c}_{f}.so")
        cmds.append([hipcc] + flags + inc + ["-D" + d for d in defs] + [f"-DGEMX_INST_SYS={s}", f"-DGEMX_INST_CONV={c}", f"-DGEMX_INST_F64={f}",
                                                                         "-fvisibility=hidden", "-shared", os.path.join(b.CSRC, "gemx_inst.hip"), "-o",
                                                                         os.path.join(out, f"libgemx_u{s}_{c}_{f}.so")])
    with cf.ThreadPoolExecutor(max_workers=min(len(cmds), os.cpu_count() or 4)) as ex:
        list(ex.map(subprocess.check_call, cmds))
    for s, c in b.UNITS:
        for f64 in (0, 1):
            name = os.path.basename(b.unit_lib(s, c, f64))
            dst = os.path.join(out, name)
            if name in named:
                continue
            if os.path.lexists(dst):
                os.remove(dst)
            os.symlink(os.path.relpath(b.unit_lib(s, c, f64), out), dst)  # (relative: the tree travels to the GPU box)
    print(out, "units:", sorted(named), "defs:", defs)


if __name__ == "__main__":
    main()
