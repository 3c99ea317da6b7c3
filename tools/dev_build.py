#!/usr/bin/env python3
"""Variant build of libgemx.so for same-box A/B runs: recompile SOME instantiation units (with extra -D flags) and link them with
the other objects of the main build (gym_electric_motor_amd/build, which must be up to date: `build.build_library()`).

    python tools/dev_build.py --units 0_0_0,1_1_0 --defs GEMX_DCS_D1=32 --out gpurun_out/lib/libgemx_d32.so

The result is used with GEMX_LIBRARY=<path> (gym_electric_motor_amd/_lib.py); struct layouts (KArgs, gemx_handle) must be the
same as the main build's, i.e. only code inside the kernels / launchers of the named units may differ."""
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
    ap.add_argument("--units", required=True, help="comma-separated sys_conv_f64 units to recompile, e.g. 0_0_0,1_1_0; 'capi' for gemx_capi.hip")
    ap.add_argument("--defs", default="", help="comma-separated extra macro definitions, e.g. GEMX_DCS_D1=32,GEMX_TIMING")
    ap.add_argument("--out", default=None)
    ap.add_argument("--inplace", action="store_true", help="recompile the units INTO the main build directory and relink the main library "
                                                           "(a partial rebuild after an edit that only touches those units' code)")
    ap.add_argument("--tag", default=None, help="object directory suffix (default: derived from --defs)")
    ap.add_argument("--slim", action="store_true", help="link ONLY the named units (+ C ABI, refgen); every other unit becomes a stub that "
                                                        "fails with a message: a few MB instead of ~100 (gpurun snapshots are capped at 512 MiB)")
    args = ap.parse_args()
    units = [u for u in args.units.split(",") if u]
    defs = [d for d in args.defs.split(",") if d]
    tag = args.tag or ("_".join(d.replace("=", "") for d in defs) or "plain")
    objdir = b.OBJ_DIR if args.inplace else os.path.join(b.OBJ_DIR, "variant_" + tag)
    if args.inplace:
        assert not defs, "--inplace builds the product library: no extra definitions"
        args.out = b.LIB
    assert args.out, "--out or --inplace"
    os.makedirs(objdir, exist_ok=True)
    hipcc = b.hipcc_path()
    inc = ["-I" + os.path.join(REPO, "include"), "-I" + b.CSRC]
    dflags = ["-D" + d for d in defs]
    cmds, replaced = [], {}
    for u in units:
        if u == "capi":
            obj = os.path.join(objdir, "gemx_capi.o")
            cmds.append([hipcc] + b.FLAGS + inc + dflags + ["-c", os.path.join(b.CSRC, "gemx_capi.hip"), "-o", obj])
            replaced["gemx_capi.o"] = obj
            continue
        s, c, f = u.split("_")
        obj = os.path.join(objdir, f"gemx_inst_{u}.o")
        cmds.append([hipcc] + b.FLAGS + inc + dflags + [f"-DGEMX_INST_SYS={s}", f"-DGEMX_INST_CONV={c}", f"-DGEMX_INST_F64={f}", "-c",
                                                          os.path.join(b.CSRC, "gemx_inst.hip"), "-o", obj])
        replaced[f"gemx_inst_{u}.o"] = obj
    with cf.ThreadPoolExecutor(max_workers=min(len(cmds), os.cpu_count() or 4)) as ex:
        list(ex.map(subprocess.check_call, cmds))
    objs = []
    stubs = []
    for s, c in b.UNITS:
        for f64 in (0, 1):
            name = f"gemx_inst_{s}_{c}_{f64}.o"
            if args.slim and name not in replaced:
                stubs.append(f"int launch_unit_{s}_{c}_{f64}(gemx_handle *, const void *, int, void *, uint8_t *, int, hipStream_t) "
                             f'{{ return fail(GEMX_ERR_ARG, "unit {s}_{c}_{f64} is not part of this slim variant build"); }}')
                continue
            objs.append(replaced.get(name, os.path.join(b.OBJ_DIR, name)))
    if stubs:
        src = os.path.join(objdir, "stubs.hip")
        with open(src, "w") as fh:
            fh.write('#include "gemx_common.hpp"\nnamespace gemx {\n' + "\n".join(stubs) + "\n}\n")
        sobj = os.path.join(objdir, "stubs.o")
        subprocess.check_call([hipcc] + b.FLAGS + inc + ["-c", src, "-o", sobj])
        objs.append(sobj)
    objs.append(replaced.get("gemx_capi.o", os.path.join(b.OBJ_DIR, "gemx_capi.o")))
    objs.append(os.path.join(b.OBJ_DIR, "gemx_refgen.o"))
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", args.out] + objs)
    if args.inplace:
        # (advisor finding, round 3) the library-wide stamp may only say "current" if EVERY object linked above was built from the current
        # sources: after an edit to a shared header (gemx_kernels.hpp, gemx_common.hpp) the objects that were not named here are stale, and a
        # mixed-layout library (KArgs / DevParams) would otherwise be loaded in silence
        stale = []
        for s_, c_ in b.UNITS:
            for f64 in (0, 1):
                name = f"gemx_inst_{s_}_{c_}_{f64}.o"
                if name in replaced:
                    continue
                st = os.path.join(b.OBJ_DIR, name + ".sha256")
                want = b._digest([os.path.join(b.CSRC, f) for f in b._DEPS["inst"]])
                if not os.path.exists(st) or open(st).read().strip() != want:
                    stale.append(name)
        for name, kind in (("gemx_capi.o", "capi"), ("gemx_refgen.o", "refgen")):
            if name in replaced:
                continue
            st = os.path.join(b.OBJ_DIR, name + ".sha256")
            if not os.path.exists(st) or open(st).read().strip() != b._digest([os.path.join(b.CSRC, f) for f in b._DEPS[kind]]):
                stale.append(name)
        for name, obj in replaced.items():  # the per-object stamps build_library() goes by
            kind = "capi" if name == "gemx_capi.o" else "inst"
            with open(obj + ".sha256", "w") as fh:
                fh.write(b._digest([os.path.join(b.CSRC, f) for f in b._DEPS[kind]]))
        if stale:
            print(f"dev_build --inplace: {len(stale)} other object(s) were built from OLDER sources ({', '.join(stale[:4])}{' ...' if len(stale) > 4 else ''}): "
                  "the library-wide stamp is NOT updated -- build.is_stale() stays True until build_library() has recompiled them", file=sys.stderr)
            if os.path.exists(b.STAMP):
                os.remove(b.STAMP)
        else:
            with open(b.STAMP, "w") as fh:
                fh.write(b._digest())
    print(args.out)


if __name__ == "__main__":
    main()
