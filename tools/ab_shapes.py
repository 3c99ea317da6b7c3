#!/usr/bin/env python3
"""A/B of the pipelined kernel's shapes on one box (GEMX_PIPE_SHAPE: <12,3> / <4,2> / <2,2>; GEMX_PIPE=0: single-wave kernel) for the
bench workloads.  python tools/ab_shapes.py [--workloads pmsm permexdc scim] [--envs N ...]  -> markdown table (kernel-only, HIP events)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", nargs="+", default=["pmsm", "permexdc", "scim"])
    ap.add_argument("--envs", type=int, nargs="*", default=[])
    ap.add_argument("--spl", type=int, default=1000)
    args = ap.parse_args()
    import torch
    import torch.distributed as dist

    import bench
    import gym_electric_motor_amd as ga

    dev = torch.device("cuda", 0)
    print("| workload | envs | GEMX_PIPE_SHAPE | GEMX_PIPE | launch ms | G env-steps/s | frac of 8 TB/s | kernel |")
    print("|---|---|---|---|---|---|---|---|")
    for key in args.workloads:
        w = dict(bench.WORKLOADS[key], key=key)
        for n in (args.envs or [w["envs"]]):
            for shape, pipe in (("", ""), ("0", ""), ("1", ""), ("2", ""), ("", "0")):
                for k, v in (("GEMX_PIPE_SHAPE", shape), ("GEMX_PIPE", pipe)):
                    if v:
                        os.environ[k] = v
                    else:
                        os.environ.pop(k, None)
                env = bench.make_env(ga, w, n, 0)
                t = bench.measure(torch, dist, env, n, 10, 3, args.spl, dev, 1, seed=3)
                r = bench.roofline_of(w, n, args.spl, t.launch_ms, env.physical_system.last_launch(), key)
                env.close()
                kern = r["kernel"].split("gemx::")[1].split(">")[0] + "> " + r["kernel"].split("grid=")[1].split(",")[0]
                print(f"| {key} | {n} | {shape or 'auto'} | {pipe or 'auto'} | {t.launch_ms:.4f} | {n * args.spl / t.launch_ms / 1e6:.1f} | {r['frac']:.3f} | {kern} |", flush=True)


if __name__ == "__main__":
    main()
