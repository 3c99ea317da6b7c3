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


def code_object_of(lib):
    """the gfx950 code object of a built library / object file, extracted into gym_electric_motor_amd/build (never beside a shipped .so)"""
    os.makedirs(BUILD, exist_ok=True)
    co = os.path.join(BUILD, os.path.basename(lib) + ".0.hipv4-amdgcn-amd-amdhsa--gfx950")
    if not os.path.exists(co) or os.path.getmtime(co) < os.path.getmtime(lib):
        link = os.path.join(BUILD, os.path.basename(lib))
        if os.path.abspath(link) != os.path.abspath(lib):  # llvm-objdump --offloading writes next to its input: give it one inside build/
            if os.path.lexists(link):
                os.remove(link)
            os.symlink(os.path.abspath(lib), link)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", os.path.basename(lib)], cwd=BUILD, capture_output=True)
    return co


def kernels_of(obj):
    co = code_object_of(obj)
    txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
    out = []
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
