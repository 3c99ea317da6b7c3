#!/usr/bin/env python3
"""Error-controlled solver: what does a WAVE pay per control step, and would a wave-shared first try change it?  (round 5 verdict, item 5)

CPU statistics through oracle/gemx_oracle.c's diagnostic restatement of the device's controller (ORC_SOLVER_DEV_ADAPTIVE: the carried
proposal, first try = tau / ceil(0.9 tau / proposal), rejections cut by clamp(0.9 err^-1/5, 0.2, 1)), 64 lock-stepped lanes per wave,
BASELINE config 4's env (Cont-SC-SCIM-v0, PolynomialStaticLoad, default constraint + auto-reset, tau = 1e-4) under i.i.d. uniform random
duty cycles -- the workload of bench.py's `scim_error_controlled` leg -- and under held actions (piecewise constant over 50 steps).

  per lane : mean attempts per control step of one env (what an env-by-env solver pays)
  per wave : mean over control steps of the SLOWEST of the 64 lanes (what the lock-stepped wave pays)
  shared   : every lane's first try = the minimum of the wave's own first tries (one DPP min-reduction on the device); rejected lanes
             still cut their own steps (the verdict's proposal)
  kinks    : every attempt on the smooth model system of the fixed-step kink correction, the kink's defect added in closed form
             (ScipyOdeSolver(split_kinks=True), the product since round 6; oracle: ORC_SOLVER_DEV_ADAPTIVE_KINK)

    python tools/wave_step_statistics.py [--waves 8] [--steps 2000] > profiles/r06_wave_step_statistics.md
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc  # noqa: E402


def stats(h):
    n = h.sum()
    a = np.arange(len(h))
    mean = float((h * a).sum() / n)
    tail = {k: float(h[k:].sum() / n) for k in (2, 3, 4, 5, 6)}
    return mean, tail


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--waves", type=int, default=8)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--golden", default="scim_epi_uniform_euler")
    args = ap.parse_args()
    orc.build()
    _, meta = orc.load_golden(args.golden)
    meta = dict(meta, tau=1e-4)
    p_plain = orc.params_from_meta(meta, solver="dev_adaptive", episodic=True)
    p_kink = orc.params_from_meta(meta, solver="dev_adaptive_kink", episodic=True)
    p_smooth = orc.params_from_meta(meta, solver="dev_adaptive", episodic=True)
    p_smooth.load_a = 0.0  # the same env without the load's kink (a = 0: no saturation term)
    rng = np.random.default_rng(1234)
    print("# Error-controlled solver: attempts per control step, per lane and per 64-lane wave (CPU statistics, device controller in fp64)\n")
    print(f"Env: {meta.get('env_id', args.golden)}, tau 1e-4, rtol 1e-6 / atol 1e-9, {args.waves} waves x 64 lanes x {args.steps} control steps, episodic.\n")
    print("| actions | first try | per lane: mean | P(>=2) | P(>=3) | P(>=4) | per wave (slowest of 64): mean | P(>=3) | P(>=4) | P(>=5) | P(>=6) |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    rows = {}
    # (name, params, wave-shared first try, omega's absolute tolerance in normalised units)
    variants = (("per lane (rounds 4-5)", p_plain, False, False), ("wave-shared minimum", p_plain, True, False),
                ("per lane, kinks in closed form", p_kink, False, False),
                ("per lane, omega's atol normalised (atol x speed limit)", p_plain, False, True),
                ("per lane, kinks in closed form + omega's atol normalised (round 6 product)", p_kink, False, True),
                ("per lane, the env WITHOUT the load's kink (a = 0)", p_smooth, False, False))
    for label, held in (("i.i.d. uniform per step", 1), ("held for 50 steps", 50)):
        for vname, p, shared, at_w in variants:
            orc.lib().orc_dev_set_atol_omega_scaled(int(at_w))
            hl, hw = np.zeros(32, dtype=np.int64), np.zeros(32, dtype=np.int64)
            r2 = np.random.default_rng(1234)
            for _ in range(args.waves):
                a = r2.uniform(-1, 1, ((args.steps + held - 1) // held, 64, 3))
                a = np.repeat(a, held, axis=0)[: args.steps]
                l, w = orc.wave_attempts(p, a, shared=shared)
                hl += l
                hw += w
            ml, tl = stats(hl)
            mw, tw = stats(hw)
            rows[(label, vname)] = (ml, mw)
            print(f"| {label} | {vname} | {ml:.2f} | {tl[2]:.3f} | {tl[3]:.3f} | {tl[4]:.4f} | "
                  f"{mw:.2f} | {tw[3]:.3f} | {tw[4]:.3f} | {tw[5]:.3f} | {tw[6]:.4f} |")
    orc.lib().orc_dev_set_atol_omega_scaled(0)
    print()
    for label in ("i.i.d. uniform per step", "held for 50 steps"):
        v = [rows[(label, x[0])][1] for x in variants]
        print(f"- {label}: the wave pays {v[0]:.2f} attempts per control step with per-lane first tries, {v[1]:.2f} with the wave-shared minimum "
              f"({(v[0] / v[1] - 1) * 100:+.0f} % rate at an attempt-bound launch), {v[2]:.2f} with the load's kinks corrected in closed form, {v[3]:.2f} with "
              f"omega's absolute tolerance in normalised units, {v[4]:.2f} with both (the product: {(v[0] / v[4] - 1) * 100:+.0f} %); the same env without the kink: {v[5]:.2f}.")
    print("\nReading: a lane needs ~1.1 attempts per control step; what the wave pays is the slowest of its 64 lanes.  Two things make some lane slow in most "
          "control steps, both at omega ~ 0, where every speed-control episode starts (and under random duty cycles an episode lasts ~40 steps): "
          "(1) the PolynomialStaticLoad's kink at |omega| = a tau_decay / J = 0.009 rad/s, whose crossing costs the error estimate five to eight cuts and leaves a "
          "small carried proposal behind; (2) an ABSOLUTE tolerance of 1e-9 rad/s on a state whose relative term vanishes there -- 2e-12 of the speed range, below "
          "anything an observation can show -- so that steps with |omega| ~ 1e-4 are rejected at error norms of 1-5.  A wave-shared first try cannot help (the cuts "
          "are forced, not the result of a poor proposal) and makes held actions slower (every lane takes the most careful lane's step).  The product integrates "
          "the smooth model system with the kink's defect in closed form AND prices omega's absolute error in normalised units (atol x speed limit = 4e-7 rad/s): "
          "one attempt per control step for every lane; against the scipy-dopri5 restatement 4e-6 either way (fp64, 3000 steps).")


if __name__ == "__main__":
    main()
