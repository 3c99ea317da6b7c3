"""Build the in-tree HIP library `libgemx.so` for gfx950 (explicit hipcc, no JIT cache)."""
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG_DIR)
SRC = os.path.join(PKG_DIR, "csrc", "gemx.hip")
HEADER = os.path.join(REPO, "include", "gemx.h")
LIB = os.path.join(PKG_DIR, "libgemx.so")


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build gym_electric_motor_amd/libgemx.so")


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in (SRC, HEADER))


def build_library(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> gym_electric_motor_amd/libgemx.so (cross-compiles without a GPU)."""
    if not force and not is_stale():
        return LIB
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on", "-I" + os.path.join(REPO, "include"),
           "-shared", "-fPIC", "-o", LIB, SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB
