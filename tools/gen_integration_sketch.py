#!/usr/bin/env python3
"""Derives the ctypes mirror of `gemx_config` from include/gemx.h, so that INTEGRATION.md's binding sketch cannot drift from the ABI.

    python tools/gen_integration_sketch.py          -> prints the `_fields_` block INTEGRATION.md section 1 carries
tests/test_host_cpu.py::test_config_struct_header_binding_and_docs_agree compares header, gym_electric_motor_amd/_lib.py and the
markdown with `parse_struct()` / `render_fields()`."""
import os
import re

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CTYPES = {"int32_t": "C.c_int32", "uint32_t": "C.c_uint32", "uint64_t": "C.c_uint64", "int64_t": "C.c_int64", "double": "C.c_double", "float": "C.c_float"}


def header_constants(text):
    """#define NAME <int expr> -> {NAME: int}"""
    out = {}
    for m in re.finditer(r"^#define\s+(GEMX_\w+)\s+\(?([0-9]+)\)?\s*(?:/\*.*)?$", text, re.M):
        out[m.group(1)] = int(m.group(2))
    return out


def parse_struct(text, name="gemx_config"):
    """[(c_type, field, array_len or None)] of `typedef struct <name> { ... } <name>;` in declaration order."""
    body = re.search(r"typedef struct " + name + r"\s*\{(.*?)\}\s*" + name + r"\s*;", text, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    consts = header_constants(text)
    fields = []
    for decl in body.split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        ctype, rest = decl.split(" ", 1)
        for item in rest.split(","):
            item = item.strip()
            m = re.match(r"(\w+)\s*(?:\[(.+)\])?$", item)
            n = None
            if m.group(2):
                expr = m.group(2)
                for k, v in consts.items():
                    expr = re.sub(r"\b" + k + r"\b", str(v), expr)
                n = int(eval(expr, {"__builtins__": {}}))  # noqa: S307 -- integer products of the header's own constants
            fields.append((ctype, m.group(1), n))
    return fields


def render_fields(fields, indent="    "):
    lines, cur = [], indent + "_fields_ = ["
    for ctype, name, n in fields:
        t = CTYPES[ctype] + (f" * {n}" if n else "")
        item = f'("{name}", {t}), '
        if len(cur) + len(item) > 118:
            lines.append(cur.rstrip())
            cur = indent + "            "
        cur += item
    lines.append(cur.rstrip().rstrip(",") + "]")
    return "\n".join(lines)


if __name__ == "__main__":
    print(render_fields(parse_struct(open(os.path.join(REPO, "include", "gemx.h")).read())))
