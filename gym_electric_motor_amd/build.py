"""Build the in-tree HIP libraries for gfx950 (explicit hipcc, no JIT cache).

Sources (gym_electric_motor_amd/csrc):
    gemx_common.hpp, gemx_kernels.hpp   device templates
    gemx_inst.hip                       ONE kernel unit, compiled once per (system, converter unit, dtype) into its OWN shared object
                                        libgemx_u<S>_<C>_<F>.so, which gemx_create loads with dlopen (round 6: a handle maps the C ABI and
                                        the one unit it runs, not all 38)
    gemx_capi.hip                       C ABI (include/gemx.h), validation, small kernels, unit loader  } libgemx.so
    gemx_refgen.hip                     device-side Wiener-process reference generation (gemx_refgen_*) }
The 38 units (19 system/converter pairs x fp32, fp64) and the two objects of libgemx.so are compiled in parallel.
"""
import concurrent.futures as cf
import glob
import hashlib
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
SOURCES = [os.path.join(CSRC, f) for f in ("gemx_common.hpp", "gemx_kernels.hpp", "gemx_inst.hip", "gemx_capi.hip", "gemx_refgen.hip")]
HEADER = os.path.join(REPO, "include", "gemx.h")
LIB = os.path.join(PKG_DIR, "libgemx.so")


def unit_lib(s, c, f64):
    """the shared object of one kernel unit (beside libgemx.so: gemx_capi.hip's load_unit looks there)"""
    return os.path.join(PKG_DIR, f"libgemx_u{s}_{c}_{int(f64)}.so")


OBJ_DIR = os.path.join(PKG_DIR, "build")
STAMP = LIB + ".sha256"  # next to the library (the object directory does not travel to the GPU box: .gpurunignore)

# (system_kind, converter_kind) pairs on the accelerated path; each for fp32 (0) and fp64 (1)
UNITS = [(0, 0), (1, 1), (1, 2), (2, 1), (2, 2), (0, 3), (3, 0), (3, 3), (4, 0), (4, 3), (5, 4), (5, 5), (6, 6), (6, 7), (7, 8), (7, 9), (1, 10), (2, 10), (6, 11)]
# --offload-compress: the gfx950 code objects are stored zstd-compressed in the shared objects (the HIP runtime inflates them when a library
# is loaded): 144 MB of libraries -> ~20 MB pushed to every GPU lease, nothing else changes
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on", "-fPIC", "-Wno-unused-variable", "--offload-compress"]


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build gym_electric_motor_amd/libgemx.so")


def _digest(sources=None, header=None):
    h = hashlib.sha256()
    for p in (SOURCES if sources is None else sources) + [header or HEADER]:
        h.update(open(p, "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


# what each kind of object is compiled from (an edit to the C ABI alone does not recompile the 38 kernel units)
_DEPS = {"inst": ["gemx_common.hpp", "gemx_kernels.hpp", "gemx_inst.hip"], "capi": ["gemx_common.hpp", "gemx_capi.hip"],
         "refgen": ["gemx_common.hpp", "gemx_refgen.hip"]}
# compile cost of an fp32 unit by system kind (object size, MB: induction > synchronous > DC); fp64 units take the single-wave kernel only
_COST = {7: 6.0, 2: 5.3, 6: 4.9, 1: 4.5, 5: 4.0, 4: 3.9, 3: 3.5, 0: 3.5}


def all_libs():
    return [LIB] + [unit_lib(s, c, f) for s, c in UNITS for f in (0, 1)]


def is_stale():
    if not os.path.exists(STAMP) or not all(os.path.exists(p) for p in all_libs()):
        return True
    return open(STAMP).read().strip() != _digest()


def build_library(force=False, verbose=False, jobs=None):
    """hipcc --offload-arch=gfx950 -> gym_electric_motor_amd/libgemx.so (cross-compiles without a GPU)."""
    if not force and not is_stale():
        return LIB
    hipcc = hipcc_path()
    os.makedirs(OBJ_DIR, exist_ok=True)
    # ONE build at a time (a CPU test calls build() too: two builds racing for the snapshot directory and the objects produce garbage);
    # a second caller waits here and then finds the objects the first one made
    import fcntl

    lock = open(os.path.join(OBJ_DIR, ".build.lock"), "w")
    fcntl.flock(lock, fcntl.LOCK_EX)
    try:
        if not force and not is_stale():
            return LIB
        return _build_locked(hipcc, force, verbose, jobs)
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()


def _build_locked(hipcc, force, verbose, jobs):
    # compile from a SNAPSHOT of the sources taken now: a full build runs for minutes with the units starting at different times, and an
    # edit of a header in between would otherwise give objects of two different versions of it (and a stamp that matches neither)
    snap = os.path.join(OBJ_DIR, "src_snapshot")
    shutil.rmtree(snap, ignore_errors=True)
    os.makedirs(snap)
    for f in SOURCES + [HEADER]:
        shutil.copy2(f, snap)
    snap_sources = [os.path.join(snap, os.path.basename(f)) for f in SOURCES]
    snap_header = os.path.join(snap, os.path.basename(HEADER))
    csrc = snap
    inc = ["-I" + snap]
    cmds = []  # (output, command, kind, cost)
    for s, c in UNITS:
        for f64 in (0, 1):
            out = unit_lib(s, c, f64)
            cmds.append((out, [hipcc] + FLAGS + inc + [f"-DGEMX_INST_SYS={s}", f"-DGEMX_INST_CONV={c}", f"-DGEMX_INST_F64={f64}", "-fvisibility=hidden",
                                                      "-shared", os.path.join(csrc, "gemx_inst.hip"), "-o", out], "inst", _COST[s] * (0.15 if f64 else 1.0)))
    capi_obj = os.path.join(OBJ_DIR, "gemx_capi.o")
    cmds.append((capi_obj, [hipcc] + FLAGS + inc + ["-c", os.path.join(csrc, "gemx_capi.hip"), "-o", capi_obj], "capi", 0.3))
    refgen_obj = os.path.join(OBJ_DIR, "gemx_refgen.o")
    cmds.append((refgen_obj, [hipcc] + FLAGS + inc + ["-c", os.path.join(csrc, "gemx_refgen.hip"), "-o", refgen_obj], "refgen", 0.2))
    digests = {k: _digest([os.path.join(csrc, f) for f in v], snap_header) for k, v in _DEPS.items()}

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    def compile_one(job):  # skipped when the object was built from the same sources with the same flags
        obj, cmd, kind, _ = job
        stamp = os.path.join(OBJ_DIR, os.path.basename(obj) + ".sha256")
        if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read().strip() == digests[kind]:
            return
        run(cmd)
        with open(stamp, "w") as fh:
            fh.write(digests[kind])

    jobs = jobs or min(len(cmds), os.cpu_count() or 4)
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:  # longest first: the big induction-machine units do not end up as the tail
        list(ex.map(compile_one, sorted(cmds, key=lambda j: -j[3])))
    run([hipcc, "--offload-arch=gfx950", "--offload-compress", "-shared", "-fPIC", "-o", LIB, capi_obj, refgen_obj, "-ldl"])
    for stale in glob.glob(os.path.join(PKG_DIR, "libgemx_u*.so")):  # a unit that is no longer in UNITS must not be found by dlopen
        if stale not in [j[0] for j in cmds]:
            os.remove(stale)
    with open(STAMP, "w") as fh:
        fh.write(_digest(snap_sources, snap_header))  # (of what was compiled: an edit made meanwhile leaves the library stale, as it should)
    return LIB
