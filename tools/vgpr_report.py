#!/usr/bin/env python3
"""Register report of the built kernels: VGPRs, spilled VGPRs and scratch bytes of every kernel in gym_electric_motor_amd/libgemx_u*.so
(llvm-readelf --notes of the gfx950 code objects), and the kernels that sit just ABOVE an occupancy line.

    python tools/vgpr_report.py [--units 2_2_0,1_1_0] [--all] > profiles/<round>_vgpr_report.md

gfx950: 512 VGPRs per SIMD lane, allocated in blocks of 8 -> <= 128 VGPRs: four waves per SIMD, <= 168: three, <= 256: two.  The shallow
pipelined shapes (<4, 2>, <2, 2>: four workgroups of four waves per CU = four waves per SIMD) need <= 128 to keep four workgroups
resident: a kernel at 129-136 loses a quarter of its residency (round 4: four pinned VGPRs took BASELINE config 4 under
ScipyOdeSolver() from 0.15 to 0.075 of the roofline that way).  `cliff` rows = within 8 registers above a line."""
import argparse
import glob
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(REPO, "gym_electric_motor_amd", "build")
LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects_of(lib):
    """the gfx950 code objects of a built library (one per translation unit), extracted into gym_electric_motor_amd/build (never beside a
    shipped .so): .hip_fatbin section -> its bundles (a library linked from several objects carries several, back to back) ->
    clang-offload-bundler --unbundle, which also inflates --offload-compress bundles (llvm-objdump --offloading writes those out still
    compressed)"""
    os.makedirs(BUILD, exist_ok=True)
    base = os.path.join(BUILD, os.path.basename(lib))
    done = sorted(glob.glob(base + ".*.gfx950.co"))
    if done and all(os.path.getmtime(c) >= os.path.getmtime(lib) for c in done):
        return done
    for c in done:
        os.remove(c)
    fb = base + ".fatbin"
    subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fb, lib, os.devnull], check=True, capture_output=True)
    blob = open(fb, "rb").read()
    os.remove(fb)
    starts = sorted(m.start() for magic in (b"CCOB", b"__CLANG_OFFLOAD_BUNDLE__") for m in re.finditer(re.escape(magic), blob))
    # (a compressed bundle's payload may contain the magic by chance: keep the starts that unbundle)
    out = []
    for i, a in enumerate(starts):
        part, co = base + f".{i}.bundle", base + f".{len(out)}.gfx950.co"
        seg = blob[a:starts[i + 1] if i + 1 < len(starts) else len(blob)]
        if seg[:4] == b"CCOB":  # compressed bundle: its header carries the exact size (the section pads every bundle to 4 KB, which the inflater refuses)
            ver = int.from_bytes(seg[4:6], "little")
            size = int.from_bytes(seg[8:16], "little") if ver >= 3 else (int.from_bytes(seg[8:12], "little") if ver == 2 else len(seg))
            seg = seg[:size] if 0 < size <= len(seg) else seg
        open(part, "wb").write(seg)
        r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + part,
                            "--output=" + co, "--unbundle"], capture_output=True)
        os.remove(part)
        if r.returncode == 0 and os.path.exists(co) and os.path.getsize(co) > 0:
            out.append(co)
        elif os.path.exists(co):
            os.remove(co)
    return out


def kernels_of(obj):
    out = []
    for co in code_objects_of(obj):
        txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
        for b in txt.split("- .agpr_count")[1:]:
            name = re.search(r"\.name:\s+(\S+)", b).group(1)
            g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", b).group(1))  # noqa: E731
            out.append((name, g("vgpr_count"), g("vgpr_spill_count"), g("private_segment_fixed_size"), g("sgpr_count")))
    names = subprocess.run(["c++filt"], input="\n".join(k[0] for k in out), capture_output=True, text=True).stdout.splitlines()
    return [(re.sub(r"\((gemx::)?KArgs<(float|double)>\)$", "", n.replace("gemx::", "").replace("void ", "")),) + k[1:] for n, k in zip(names, out)]


def unit_path(u):
    """u = '<sys>_<conv>_<f64>' -> the unit's shared object (round 6: one per unit, beside libgemx.so)"""
    return os.path.join(REPO, "gym_electric_motor_amd", f"libgemx_u{u}.so")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--units", default=None, help="comma-separated sys_conv_f64 units (default: every fp32 unit)")
    ap.add_argument("--all", action="store_true", help="list every kernel, not only cliffs / spills")
    args = ap.parse_args()
    objs = sorted(glob.glob(unit_path("*_0"))) if args.units is None else [unit_path(u) for u in args.units.split(",")]
    print("| unit | kernel | VGPRs | spilled | scratch B | note |")
    print("|---|---|---|---|---|---|")
    n_all = n_cliff = n_spill = 0
    for obj in objs:
        unit = os.path.basename(obj)[len("libgemx_u"):-3]
        for name, v, sp, sc, sg in kernels_of(obj):
            n_all += 1
            shallow = bool(re.search(r"advance_pipe_kernel<.*, (4|2), 2, (false|true)(, (false|true))?>", name))
            cliff = 128 < v <= 136 or 168 < v <= 176
            note = []
            if cliff:
                note.append("cliff: just above the %d-VGPR line" % (128 if v <= 136 else 168) + (" (shallow shape: 3 resident workgroups per CU instead of 4)" if shallow and v <= 136 else ""))
                n_cliff += 1
            if sp or sc:
                note.append("spills" if sp else "scratch")
                n_spill += 1
            if args.all or note:
                print(f"| {unit} | `{name}` | {v} | {sp} | {sc} | {'; '.join(note)} |")
    print(f"\n{n_all} kernels in {len(objs)} units: {n_cliff} within 8 VGPRs above an occupancy line, {n_spill} with spills or scratch")


if __name__ == "__main__":
    sys.exit(main())
