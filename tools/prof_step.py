#!/usr/bin/env python3
"""Single-step (closed-loop) path: one gemx_step launch per control step; host vs device time."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="pmsm")
    ap.add_argument("--envs", type=int, default=16384)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured HIP graph")
    args = ap.parse_args()
    import torch
    import bench
    import gym_electric_motor_amd as ga

    w = dict(bench.WORKLOADS[args.workload], key=args.workload)
    env = bench.make_env(ga, w, args.envs, 0)
    ps = env.physical_system
    acts = bench.make_actions(torch, ps, 64, args.envs, torch.device("cuda", 0), 1)
    for k in range(50):
        ps.simulate(acts[k % 64])
    torch.cuda.synchronize()
    # (a) python simulate()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ps.simulate(acts[k % 64])
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    # (b) raw ctypes call with precomputed pointers
    L, h = ps._L, ps._handle
    st = ps._stream()
    ptrs = [C.c_void_p(acts[k].data_ptr()) for k in range(64)]
    po, pd = C.c_void_p(ps._obs.data_ptr()), C.c_void_p(ps._done.data_ptr())
    t0 = time.perf_counter()
    for k in range(args.steps):
        L.gemx_step(h, ptrs[k % 64], po, pd, st)
    t_issue2 = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all2 = time.perf_counter() - t0
    print(f"{args.workload} N={args.envs}: simulate() issue {t_issue/args.steps*1e6:.2f} us/step, total {t_all/args.steps*1e6:.2f} us/step; "
          f"raw gemx_step issue {t_issue2/args.steps*1e6:.2f}, total {t_all2/args.steps*1e6:.2f} us/step  [{ps.last_launch()}]", flush=True)
    if args.graph:
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for k in range(3):
                ps.simulate(acts[0])
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                for k in range(64):
                    ps.simulate(acts[k])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps // 64):
            g.replay()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"   HIP graph of 64 captured gemx_step launches: {dt/(args.steps//64*64)*1e6:.2f} us/step", flush=True)


if __name__ == "__main__":
    main()
