#!/usr/bin/env python3
"""Record what the REFERENCE's env shell consumes from its physical system -- TEST INFRASTRUCTURE ONLY (build container: needs
/root/reference; the GPU box replays the committed transcripts against the HIP stepper, tests/test_gpu_parity.py).

    MPLBACKEND=Agg python oracle/make_shell_transcript.py        ->  tests/golden/shell_<env id>.json

The unmodified `gem.make(env_id)` is run with ONE change: `ElectricMotorEnvironment.__init__` (core.py:197-289) receives a recording
proxy in place of the `PhysicalSystem` instance the env class built.  The proxy forwards everything to the real SCML system and logs,
in order, every attribute the shell READS (by the env itself, the reference generator, the reward function, the constraint monitor and
the dashboards: `set_modules(physical_system)` and friends) and every method it CALLS with arguments and return value
(`simulate(action)` core.py:328-371, `reset()` core.py:300-319, `seed()`, `close()`), over a seeded 200-step random-action run with
reset on termination.  A replacement physical system that answers that transcript identically is, by construction, a drop-in for the
shell on that run.  One case also carries a wrapper stack (DeadTimeProcessor around the system): there the proxy sits OUTSIDE the
wrapper, where the folded kernel's interface is."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = os.environ.get("GEM_REFERENCE", "/root/reference")
os.environ.setdefault("MPLBACKEND", "Agg")
sys.path.insert(0, os.path.join(HERE, "gymnasium_standin"))
sys.path.insert(0, os.path.join(REF, "src"))

import numpy as np  # noqa: E402

import gym_electric_motor as gem  # noqa: E402
from gym_electric_motor import core  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")


def encode(v, depth=0):
    """JSON-able, lossless for what the shell consumes (float64 via repr round trip)."""
    if isinstance(v, (bool, int, str)) or v is None:
        return v
    if isinstance(v, float):
        return {"f": repr(v)}
    if isinstance(v, np.generic):
        return encode(v.item())
    if isinstance(v, np.ndarray):
        return {"nd": [repr(float(x)) for x in v.ravel()], "shape": list(v.shape), "dtype": str(v.dtype)}
    if isinstance(v, (list, tuple)):
        return {"seq": [encode(x, depth + 1) for x in v], "tuple": isinstance(v, tuple)}
    if isinstance(v, dict):
        return {"dict": {str(k): encode(x, depth + 1) for k, x in v.items()}}
    name = type(v).__name__
    if name == "Box":
        return {"Box": {"low": encode(np.asarray(v.low, dtype=float)), "high": encode(np.asarray(v.high, dtype=float)), "shape": list(v.shape)}}
    if name == "Discrete":
        return {"Discrete": int(v.n)}
    if name == "MultiDiscrete":
        return {"MultiDiscrete": [int(x) for x in v.nvec]}
    return {"object": name}  # components (converter, motor, ...), the unwrapped system itself: identity only


class RecordingProxy(core.PhysicalSystem):
    """Stands where the shell expects a PhysicalSystem (isinstance holds); every access goes to the real one and into the log."""

    def __init__(self, real, log):
        object.__setattr__(self, "_rp_real", real)
        object.__setattr__(self, "_rp_log", log)

    def __getattribute__(self, name):
        if name in ("_rp_real", "_rp_log", "__class__", "__dict__"):
            return object.__getattribute__(self, name)
        real, log = object.__getattribute__(self, "_rp_real"), object.__getattribute__(self, "_rp_log")
        val = getattr(real, name)
        if callable(val) and not isinstance(val, type):
            def call(*args, **kwargs):
                ret = val(*args, **kwargs)
                log.append({"op": "call", "name": name, "args": encode(list(args)), "kwargs": encode(kwargs), "ret": encode(ret)})
                return ret

            return call
        log.append({"op": "get", "name": name, "value": encode(val)})
        return val

    def __setattr__(self, name, value):
        object.__getattribute__(self, "_rp_log").append({"op": "set", "name": name, "value": encode(value)})
        setattr(object.__getattribute__(self, "_rp_real"), name, value)


def record(env_id, steps=200, seed=4321, wrappers=None, tag=None):
    log = []
    orig_init = core.ElectricMotorEnvironment.__init__

    def patched(self, physical_system, *a, **kw):
        ws = tuple(kw.pop("physical_system_wrappers", ()))
        for w in ws:  # wrappers INSIDE the proxy: the transcript is taken at the interface a folded (wrapper-aware) kernel offers
            physical_system = w.set_physical_system(physical_system)
        return orig_init(self, RecordingProxy(physical_system, log), *a, physical_system_wrappers=(), **kw)

    core.ElectricMotorEnvironment.__init__ = patched
    try:
        kw = {}
        if wrappers:
            from gym_electric_motor.physical_system_wrappers import DeadTimeProcessor

            kw["physical_system_wrappers"] = tuple(DeadTimeProcessor(steps=int(w[4:])) for w in wrappers)
        env = gem.make(env_id, **kw)
    finally:
        core.ElectricMotorEnvironment.__init__ = orig_init
    n_init = len(log)
    env.reset(seed=0)
    rng = np.random.default_rng(seed)
    space = env.action_space
    for k in range(steps):
        if type(space).__name__ == "Discrete":
            a = int(rng.integers(0, space.n))
        else:
            a = rng.uniform(-1, 1, space.shape) * rng.uniform(0, 1)
        _, _, terminated, _, _ = env.step(a)
        if terminated:
            env.reset()
    env.close()
    names = sorted({e["name"] for e in log})
    doc = {"env_id": env_id, "wrappers": list(wrappers or ()), "steps": steps, "seed": seed, "entries_during_construction": n_init,
           "attribute_names": names, "log": log,
           "source": "reference ElectricMotorEnvironment (core.py:197-371) around a recording proxy of its own physical system"}
    path = os.path.join(OUT, f"shell_{tag or env_id}.json")
    with open(path, "w") as fh:
        json.dump(doc, fh, separators=(",", ":"))
    ncall = sum(e["op"] == "call" for e in log)
    print(f"{os.path.basename(path):48s} {len(log):6d} entries ({n_init} during construction, {ncall} calls), names: {names}")


def main():
    for env_id in ("Cont-CC-PermExDc-v0", "Finite-CC-PMSM-v0", "Cont-SC-SCIM-v0"):
        record(env_id)
    record("Finite-CC-PMSM-v0", wrappers=("dead2",), tag="Finite-CC-PMSM-v0_DeadTime2")


if __name__ == "__main__":
    main()
