import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import test_gpu_parity as T
for lm in ("1", "0"):
    os.environ["GEMX_LINMAP"] = lm
    d, meta, obs, done = T._run_golden("permexdc_free_held_til_euler", "float32")
    rel, col = T._rel_err(obs[d["state_index"]], d["states"], meta["state_names"])
    print("LINMAP", lm, rel, col, meta["tau"], meta.get("interlocking_time"), meta["solver"], meta["converter"])
    print(obs[:4], d["states"][:4])
