#!/usr/bin/env python3
"""GPU box: where do the DFIM's field-oriented columns leave the 1e-4 contract on random-action lanes, and is it conditioning?
For every lane of LANE_SAMPLE: device fp32 row vs the fp64 oracle (same integrator) step by step, with the oracle's rotor flux
magnitude at the START of the step (the flux the field angle of that row is the arctan2 of).  TEST INFRASTRUCTURE (imports oracle/)."""
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
import test_gpu_parity as T  # noqa: E402


def main(name="default_finite_tc_dfim_dopri5", solver="default", n_envs=70):
    import torch

    from oracle import oracle as orc

    d, meta = T._load(name)
    env = T._make_from_meta(meta, n_envs, solver=solver, dtype="float32", auto_reset=True)
    ps = env.physical_system
    acts = d["actions"]
    K = acts.shape[0]
    a_np = np.repeat(acts.reshape(K, 1, -1), n_envs, axis=1).copy()
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    recorded = sorted({0, 64 % n_envs, n_envs - 1})
    others = [j for j in range(n_envs) if j not in recorded]
    if ps._discrete:
        nvec = [int(v) for v in ps.action_space.nvec] if hasattr(ps.action_space, "nvec") else [int(ps.action_space.n)]
        for c, nv in enumerate(nvec):
            a_np[:, others, c] = rng.integers(0, nv, (K, len(others))).astype(a_np.dtype)
    else:
        a_np[:, others, :] = rng.uniform(-1.0, 1.0, (K, len(others), a_np.shape[2]))
    a = torch.as_tensor(a_np)
    if ps._discrete and acts.ndim == 1:
        a = a.reshape(K, n_envs)
    sol_obj = ps._ode_solver
    obs, done = env.rollout(a.cuda())
    torch.cuda.synchronize()
    obs = obs.double().cpu().numpy()
    done = done.cpu().numpy().astype(bool)
    env.close()
    osol = T._oracle_solver_for(meta, sol_obj)
    print(f"# {name} solver={solver} -> oracle {osol}; K={K}")
    names = meta["state_names"]
    lim = np.asarray(meta["limits"])
    p = orc.params_from_meta(meta, solver=osol[0])
    p.nsteps = osol[1]
    for j in range(1, n_envs):
        if j in recorded:
            continue
        e = orc.OracleEnv(p)
        e.reset()
        aj = a_np[:, j, :] if acts.ndim > 1 else a_np[:, j, 0]
        aj = aj.astype(np.float64).reshape(K, -1)
        ref = np.zeros((K, len(names)))
        psi0 = np.zeros(K)
        psi1 = np.zeros(K)
        for k in range(K):
            psi0[k] = np.hypot(e.y[3], e.y[4])
            o = e.step(aj[k])
            psi1[k] = np.hypot(e.y[3], e.y[4])
            ref[k] = o
            if e.done(o):
                e.reset()
        diff = np.abs(obs[:, j] - ref)
        i = names.index("epsilon")
        diff[:, i] = np.minimum(diff[:, i], 2.0 - diff[:, i])
        scale = np.maximum(np.abs(ref).max(axis=0), 1e-3)
        per_col = diff.max(axis=0) / scale
        c = int(np.argmax(per_col))
        k = int(np.argmax(diff[:, c]))
        nd = {n: f"{per_col[names.index(n)]:.1e}" for n in ("i_sa", "i_sd", "i_sq", "u_sd", "u_sq", "i_rd", "u_rd", "torque", "epsilon")}
        print(f"lane {j:2d} worst {names[c]} {per_col[c]:.2e} at step {k}: |psi_r| start {psi0[k]:.3e} Wb (max {psi0.max():.3e}), "
              f"err*|psi|/max|psi| = {per_col[c] * psi0[k] / psi0.max():.2e}; cols {nd}")
        if per_col[c] > 5e-5:
            # the conditioning bound: |d u_sd| <= |u_s| d(angle), d(angle) ~ d(psi) / |psi|
            for kk in np.argsort(-diff[:, c])[:5]:
                print(f"    step {kk}: err {diff[kk, c] / scale[c]:.2e} |psi| {psi0[kk]:.3e} scaled {diff[kk, c] / scale[c] * psi0[kk] / psi0.max():.2e}")


if __name__ == "__main__":
    main(*sys.argv[1:3])
