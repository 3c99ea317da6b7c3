"""Multi-GPU: env instances are independent, so the path shards with NO data-path collective.

Rank r of W (one process per GPU, torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for
the tests) owns the contiguous env range `shard_range(n_total, r, W)`, with its own SoA state on its own GPU; the
(few hundred bytes of) parameters are replicated.  The only collective is on the batched-return path, when a
caller wants every rank's observations in one place: `gather_observations` = ONE all-gather of the
`[n_local, S_out]` shards (+ the `[n_local]` done bytes).  On the 8-GPU xGMI mesh every shard travels once over
each direct link (no ring all-reduce anywhere), so gather per K-step chunk, not per step, when the policy is not
co-located (DESIGN.md section "Multi-GPU").
"""
import os


def shard_range(n_total, rank, world):
    """Contiguous, balanced partition of env indices: the first n_total % world ranks get one extra env."""
    base, rem = divmod(int(n_total), int(world))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_sizes(n_total, world):
    return [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]


def free_port():
    """A TCP port that is free on 127.0.0.1 right now -- for a PARENT that spawns its own ranks and hands every one of them the same
    MASTER_PORT (bench.py --gpus N without torchrun).  Ranks cannot pick one independently: they must agree on it."""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def init_from_env(backend=None, timeout_s=180.0, force=False):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun, the driver, or a spawning parent that
    used free_port()).  Returns (rank, world, local_rank).

    timeout_s bounds the rendezvous AND every later collective / barrier: a rank that died leaves the others with an exception after
    that long instead of a hang.  MASTER_PORT is NOT defaulted: two jobs on one node would silently meet on a fixed port (round-2
    finding) -- a multi-rank launch without it is a launcher bug and raises.  force: also initialise a world of ONE (a 1-GPU box can
    then run the collectives of the batched-return path through RCCL: tests/test_gpu_parity.py)."""
    import datetime

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            if world > 1:
                raise RuntimeError("MASTER_PORT is not set: launch the ranks with torchrun (--master-addr 127.0.0.1 --master-port P), or let "
                                   "the spawning parent pick distributed.free_port() and export it to every rank")
            os.environ["MASTER_PORT"] = str(free_port())  # a world of one talks to itself
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=float(timeout_s)))
    return rank, world, local_rank


def _all_gather_rows(out, x, group):
    """out[world * n, ...] <- every rank's x[n, ...] in rank order.  RCCL (and gloo on host tensors) gather straight into `out`; gloo with
    DEVICE tensors (bench.py --oversubscribe: several ranks on one GPU, where RCCL refuses duplicate devices) goes through the list form,
    which ProcessGroupGloo stages through host memory."""
    import torch.distributed as dist

    if dist.get_backend(group) == "gloo" and x.is_cuda:
        world = dist.get_world_size(group)
        n = x.shape[0]
        dist.all_gather([out[r * n : (r + 1) * n] for r in range(world)], x, group=group)
    else:
        dist.all_gather_into_tensor(out, x, group=group)


def gather_observations(obs_local, done_local=None, n_total=None, group=None, force=False):
    """All-gather the per-rank observation shards into `[n_total, S_out]` (and done into `[n_total]`) on every rank.

    Shards may differ by one env (unbalanced tail); they are padded to the largest shard for the collective
    and trimmed afterwards.  Works for CPU tensors with gloo and device tensors with RCCL.  A world of one returns its inputs
    untouched unless `force` sends them through the collective anyway (single-GPU test of the RCCL path)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return obs_local, done_local
    world = dist.get_world_size(group)
    n_local = obs_local.shape[0]
    if n_total is None:
        sizes_t = torch.tensor([n_local], device=obs_local.device, dtype=torch.int64)
        all_sizes = [torch.zeros_like(sizes_t) for _ in range(world)]
        dist.all_gather(all_sizes, sizes_t, group=group)
        sizes = [int(s.item()) for s in all_sizes]
    else:
        sizes = shard_sizes(n_total, world)
    n_max = max(sizes)

    def _gather(x):
        if x.shape[0] < n_max:
            pad = torch.zeros((n_max - x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
            x = torch.cat([x, pad], dim=0)
        x = x.contiguous()
        out = torch.empty((world * n_max,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        _all_gather_rows(out, x, group)
        parts = [out[r * n_max : r * n_max + sizes[r]] for r in range(world)]
        return torch.cat(parts, dim=0) if any(s != n_max for s in sizes) else out

    obs_all = _gather(obs_local)
    done_all = _gather(done_local) if done_local is not None else None
    return obs_all, done_all


def gather_rollout(obs_chunk, done_chunk=None, group=None, force=False, out=None):
    """All-gather one fused launch's outputs: `[K, n_local, S_out]` observation chunks (+ `[K, n_local]` done bytes) of every rank
    -> `[W, K, n_local, S_out]` (+ `[W, K, n_local]`) on every rank, RANK-MAJOR and zero-copy: row `[r, k, i]` is env
    `shard_range(n_total, r, W)[0] + i` at control step k.  (Interleaving the shards into `[K, n_total, S_out]` would cost a second
    pass over W times the chunk; a consumer that needs that view indexes `[:, k]` instead.)  All ranks must hold equally sized shards
    (bench.py / make_sharded with n_total % W == 0); unequal shards go through gather_observations per step.
    out: optional pair of preallocated result tensors `([W, K, n_local, S_out], [W, K, n_local])` (a training loop gathers every
    launch's chunk into the same buffers: W x 0.9 GB per launch of the headline config is not something to allocate per call)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return obs_chunk.unsqueeze(0), (done_chunk.unsqueeze(0) if done_chunk is not None else None)
    world = dist.get_world_size(group)

    def _gather(x, res):
        x = x.contiguous()
        if res is None:
            res = torch.empty((world,) + tuple(x.shape), dtype=x.dtype, device=x.device)
        elif tuple(res.shape) != (world,) + tuple(x.shape) or res.dtype != x.dtype or res.device != x.device or not res.is_contiguous():
            raise ValueError(f"gather_rollout: out buffer {tuple(res.shape)} {res.dtype} does not match world {world} x chunk {tuple(x.shape)} {x.dtype}")
        _all_gather_rows(res.view((world * x.shape[0],) + tuple(x.shape[1:])), x, group)
        return res

    o_obs, o_done = out if out is not None else (None, None)
    return _gather(obs_chunk, o_obs), (_gather(done_chunk, o_done) if done_chunk is not None else None)


def make_sharded(env_id, n_envs_total, rank, world, device, **kwargs):
    """This rank's shard of a `n_envs_total`-env batched environment: envs [lo, hi) of the job, with `env_base = lo`, so that every
    device-side random stream (random initial states, `rollout_synthetic` actions, the Wiener reference generators that are given this
    system) is keyed by the env's GLOBAL index -- W shards draw exactly what one unsharded system of n_envs_total envs draws (round 5
    keyed them by the local index: every rank replayed rank 0's draws)."""
    from .envs import make

    lo, hi = shard_range(n_envs_total, rank, world)
    kwargs.setdefault("env_base", lo)
    env = make(env_id, n_envs=hi - lo, device=device, **kwargs)
    env.shard = (lo, hi)
    env.n_envs_total = n_envs_total
    return env
