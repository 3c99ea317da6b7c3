#!/usr/bin/env python3
"""Static instruction mix of the LOOPS of one kernel (which wave role issues what): disassembles a unit's gfx950 code object, finds the
backward branches of the named kernel and prints, per loop body, the count of VALU / SALU / LDS / VMEM / SMEM instructions and a few
marker instructions that identify the role (v_pk_fma: integrator, global_store: output waves, global_load_lds: loader).

    python tools/isa_loops.py 2_2_0 "advance_pipe_kernel<2, 2, 1, 1, false, float, 2, 2, false, false, false>"
"""
import collections
import importlib.util
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("vr", os.path.join(REPO, "tools", "vgpr_report.py"))
vr = importlib.util.module_from_spec(spec)
spec.loader.exec_module(vr)


def cls(op):
    if op.startswith("v_"):
        return "VALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "flat_", "scratch_", "buffer_")):
        return "VMEM"
    if op in ("s_waitcnt", "s_nop", "s_barrier", "s_sleep", "s_setprio", "s_endpgm", "s_waitcnt_depctr"):
        return "wait/other"
    if op.startswith(("s_load", "s_memtime", "s_memrealtime", "s_buffer_load")):
        return "SMEM"
    if op.startswith("s_"):
        return "SALU"
    return "wait/other"


def main():
    unit, want = sys.argv[1], sys.argv[2]
    co = vr.code_objects_of(vr.unit_path(unit))[0]
    out = subprocess.run([os.path.join(vr.LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
    blocks = re.split(r"\n(?=[0-9a-f]+ <)", out)
    for blk in blocks:
        m = re.match(r"([0-9a-f]+) <(\S+)>:", blk)
        if not m:
            continue
        name = subprocess.run(["c++filt", m.group(2)], capture_output=True, text=True).stdout.strip()
        if want not in name:
            continue
        ins = []  # (address, op, text)
        for ln in blk.splitlines()[1:]:
            mm = re.match(r"\s*(\S+)(.*?)//\s*([0-9A-Fa-f]+):", ln)
            if mm:
                ins.append((int(mm.group(3), 16), mm.group(1), ln))
        addr_index = {a: i for i, (a, _, _) in enumerate(ins)}
        base = int(m.group(1), 16)
        loops = set()
        for i, (a, op, ln) in enumerate(ins):
            if op.startswith(("s_cbranch", "s_branch")):
                t = re.search(r"\+0x([0-9a-f]+)>", ln)
                if t:
                    ta = base + int(t.group(1), 16)
                    if ta in addr_index and addr_index[ta] <= i:
                        loops.add((addr_index[ta], i))
        print(f"# {name}: {len(ins)} instructions, {len(loops)} loops (backward branches)\n")
        print("| loop (instruction range) | length | VALU | SALU | LDS | VMEM | SMEM | markers |")
        print("|---|---|---|---|---|---|---|---|")
        for a, b in sorted(loops, key=lambda x: -(x[1] - x[0])):
            seg = ins[a:b + 1]
            if len(seg) < 24:
                continue
            c = collections.Counter(cls(op) for _, op, _ in seg)
            txt = "\n".join(op for _, op, _ in seg)
            marks = [f"{t}: {txt.count(t)}" for t in ("v_pk_fma", "v_pk_mul", "global_store", "global_load_lds", "s_setprio", "s_memrealtime", "s_barrier", "s_cbranch",
                                                        "v_readfirstlane", "v_readlane", "s_and_saveexec", "s_mov_b32", "s_mov_b64", "s_add", "s_cmp", "s_and_b64", "s_or_b64", "s_cselect", "s_lshl", "s_mul")
                     if txt.count(t)]
            print(f"| {a}..{b} | {len(seg)} | {c['VALU']} | {c['SALU']} | {c['LDS']} | {c['VMEM']} | {c['SMEM']} | {', '.join(marks)} |")
        return
    print("kernel not found")


if __name__ == "__main__":
    main()
