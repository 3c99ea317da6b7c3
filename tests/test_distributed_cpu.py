"""world_size-2 gloo test of the N>1 path (sharding + the one gather collective) on CPU."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from gym_electric_motor_amd import distributed as gd

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_exactly():
    for n in (1, 7, 8, 262144, 262147):
        for w in (1, 2, 4, 8):
            edges = [gd.shard_range(n, r, w) for r in range(w)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(w - 1))
            sizes = gd.shard_sizes(n, w)
            assert sum(sizes) == n and max(sizes) - min(sizes) <= 1
    assert gd.shard_range(262144, 3, 8) == (98304, 131072)  # BASELINE config 5: 8 x 32768


def _worker(rank, world, port, n_total, out_dir):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist

    from gym_electric_motor_amd import distributed as gdd

    r, w, _ = gdd.init_from_env(backend="gloo")
    lo, hi = gdd.shard_range(n_total, r, w)
    # stand-in for this rank's observation shard: row i of the global batch is filled with i
    obs = torch.arange(lo, hi, dtype=torch.float32).reshape(-1, 1).repeat(1, 14)
    done = (torch.arange(lo, hi) % 3 == 0).to(torch.uint8)
    all_obs, all_done = gdd.gather_observations(obs, done, n_total=n_total)
    all_obs2, _ = gdd.gather_observations(obs, None)  # sizes discovered by a collective
    # chunk form: what one fused launch of K control steps writes on every rank -- obs [K, n_local, S_out] f32, done [K, n_local] u8
    # (equal shards only: the bench / make_sharded shapes); entry [k, i, j] encodes (step, global env, column)
    chunk = None
    if n_total % w == 0:
        K, S = 5, 14
        g = torch.arange(lo, hi, dtype=torch.float32)
        oc = (torch.arange(K, dtype=torch.float32).reshape(K, 1, 1) * 1000 + g.reshape(1, -1, 1) + torch.arange(S, dtype=torch.float32).reshape(1, 1, S) / 100).contiguous()
        dc = ((torch.arange(K).reshape(K, 1) + torch.arange(lo, hi).reshape(1, -1)) % 2).to(torch.uint8).contiguous()
        chunk = gdd.gather_rollout(oc, dc)
        assert chunk[0].shape == (w, K, hi - lo, S) and chunk[0].dtype == torch.float32 and chunk[1].shape == (w, K, hi - lo) and chunk[1].dtype == torch.uint8
    # this rank's shard of a sharded env (host-side configuration only: no GPU here).  Every device random stream is keyed by the
    # GLOBAL env index: the shard's first env goes into the physical system's config AND into a reference generator given that system
    import gym_electric_motor_amd as ga

    env = gdd.make_sharded("Finite-CC-PMSM-v0", n_total, r, w, device=0, seed=5, _defer_create=True)
    gen = ga.BatchedWienerProcessReferenceGenerator(reference_states=("i_sd",), seed=5).set_modules(env.physical_system, _defer_create=True)
    bases = (env.shard, int(env.physical_system._cfg.env_base), env.physical_system.env_base, int(gen._cfg.env_base), env.physical_system.n_envs)
    torch.save((all_obs, all_done, all_obs2, chunk, bases), os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [64, 101])
def test_gather_observations_gloo_world2(tmp_path, n_total):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, n_total, str(tmp_path)), nprocs=2, join=True)
    expect = torch.arange(n_total, dtype=torch.float32).reshape(-1, 1).repeat(1, 14)
    for r in range(2):
        all_obs, all_done, all_obs2, chunk, bases = torch.load(os.path.join(tmp_path, f"r{r}.pt"))
        lo, hi = gd.shard_range(n_total, r, 2)
        assert bases == ((lo, hi), lo, lo, lo, hi - lo)  # make_sharded: env_base = the shard's first env, on every rank
        assert torch.equal(all_obs, expect) and torch.equal(all_obs2, expect)
        assert torch.equal(all_done, (torch.arange(n_total) % 3 == 0).to(torch.uint8))
        if n_total % 2 == 0:  # rank-major chunk gather: [W, K, n_local, S_out] -> [K, n_total, S_out] must be the global batch
            oc, dc = chunk
            W, K, nl, S = oc.shape
            full = oc.permute(1, 0, 2, 3).reshape(K, W * nl, S)
            want = (torch.arange(K, dtype=torch.float32).reshape(K, 1, 1) * 1000 + torch.arange(n_total, dtype=torch.float32).reshape(1, -1, 1) +
                    torch.arange(S, dtype=torch.float32).reshape(1, 1, S) / 100)
            assert torch.equal(full, want)
            dfull = dc.permute(1, 0, 2).reshape(K, W * nl)
            assert torch.equal(dfull, ((torch.arange(K).reshape(K, 1) + torch.arange(n_total).reshape(1, -1)) % 2).to(torch.uint8))
        else:
            assert chunk is None
