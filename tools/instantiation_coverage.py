#!/usr/bin/env python3
"""Which of the compiled kernel instantiations does anything ever launch?  (round 5 verdict, item 3)

A process run with GEMX_COVERAGE_FILE=<path> appends every DISTINCT kernel instantiation it launches to that file, once, as the kernel
symbol's template arguments (gemx_capi.hip: gemx_cov_note -- the stepping kernels of the units through gemx_last_launch()'s record, the
small kernels of libgemx.so by name).  This tool diffs the union of such files against the kernel symbols of the built libraries
(gfx950 code objects of gym_electric_motor_amd/libgemx*.so) and prints the markdown report:

    GEMX_COVERAGE_FILE=$PWD/gpurun_out/cov_tests.txt python -m pytest tests -m gpu -q
    GEMX_COVERAGE_FILE=$PWD/gpurun_out/cov_bench.txt python bench.py
    GEMX_COVERAGE_FILE=$PWD/gpurun_out/cov_matrix.txt python tools/bench_matrix.py ...
    python tools/instantiation_coverage.py gpurun_out/cov_*.txt > profiles/r06_instantiation_coverage.md     (exit status 1 if a kernel is unreached)
"""
import glob
import importlib.util
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "gym_electric_motor_amd")


def _vr():
    spec = importlib.util.spec_from_file_location("vgpr_report", os.path.join(REPO, "tools", "vgpr_report.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def norm(name):
    """one spelling for a kernel: no namespace, no argument list, no `void`"""
    n = name.strip().replace("gemx::", "").replace("(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"\(.*\)$", "", n)
    return re.sub(r"\s+", " ", n)


def compiled():
    """{library basename: [kernel, ...]}"""
    vr = _vr()
    out = {}
    for lib in sorted(glob.glob(os.path.join(PKG, "libgemx*.so"))):
        out[os.path.basename(lib)] = sorted(norm(k[0]) for k in vr.kernels_of(lib))
    return out


def main():
    files = sys.argv[1:]
    if not files:
        print(__doc__)
        return 2
    launched = {}
    for f in files:
        for ln in open(f):
            ln = norm(ln)
            if ln:
                launched.setdefault(ln, set()).add(os.path.basename(f))
    libs = compiled()
    n_all = sum(len(v) for v in libs.values())
    every = {k for v in libs.values() for k in v}
    unknown = sorted(k for k in launched if k not in every)
    print("# Instantiation coverage: launched / compiled kernels per library\n")
    print(f"Coverage files: {', '.join(os.path.basename(f) for f in files)}.  Compiled: {n_all} kernels in {len(libs)} libraries; "
          f"launched (distinct): {len([k for k in launched if k in every])}.\n")
    print("| library | MB | compiled | launched | unreached |")
    print("|---|---|---|---|---|")
    missing_all = {}
    for lib, ks in libs.items():
        miss = [k for k in ks if k not in launched]
        missing_all[lib] = miss
        mb = os.path.getsize(os.path.join(PKG, lib)) / 1e6
        print(f"| {lib} | {mb:.1f} | {len(ks)} | {len(ks) - len(miss)} | {len(miss)} |")
    tot_miss = sum(len(v) for v in missing_all.values())
    print(f"| **total** | {sum(os.path.getsize(os.path.join(PKG, l)) for l in libs) / 1e6:.1f} | {n_all} | {n_all - tot_miss} | {tot_miss} |")
    if tot_miss:
        print("\n## Unreached kernels\n")
        for lib, miss in missing_all.items():
            if miss:
                print(f"### {lib}\n")
                for k in miss:
                    print(f"- `{k}`")
                print()
    if unknown:
        print("\n## Launch records without a compiled kernel of that name (tool / logging mismatch)\n")
        for k in unknown:
            print(f"- `{k}`")
    return 1 if tot_miss or unknown else 0


if __name__ == "__main__":
    sys.exit(main())
