"""CPU, world_size 2, gloo: the N>1 path — contiguous sharding of the batch and
the one all-gather that re-assembles per-instance outputs — gives exactly the
single-process result.  The per-rank producer here is the CPU oracle (the HIP
kernels cannot run without a GPU); on the GPU box the same code runs with the
`nccl` (= RCCL) backend from bench.py."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from opfgym_amd.dist import all_gather_rows, shard_bounds


def _worker(rank, world, port, total, q):
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), here, os.path.join(here, 'golden')]
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from env_cases import oracle_env
    orc = oracle_env('maxren_lv')
    rng = np.random.default_rng(5)
    steps = rng.integers(2000, 30000, total)
    actions = rng.random((total, sum(len(i) for _, _, i in orc.act_keys)))
    lo, hi = shard_bounds(total, rank, world)
    sizes = [shard_bounds(total, r, world)[1] - shard_bounds(total, r, world)[0] for r in range(world)]
    rew, obs = [], []
    for k in range(lo, hi):
        orc.reset(int(steps[k]))
        out = orc.step(actions[k])
        rew.append(out['reward'])
        obs.append(out['obs'])
    full_r = all_gather_rows(torch.tensor(rew, dtype=torch.float64), world, sizes)
    full_o = all_gather_rows(torch.tensor(np.array(obs), dtype=torch.float64), world, sizes)
    if rank == 0:
        q.put((full_r.numpy(), full_o.numpy(), steps, actions))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_cover_batch_exactly():
    for total, world in ((8192, 8), (10, 3), (7, 8), (65536, 8)):
        b = [shard_bounds(total, r, world) for r in range(world)]
        assert b[0][0] == 0 and b[-1][1] == total
        assert all(b[r][1] == b[r + 1][0] for r in range(world - 1))
        sizes = [hi - lo for lo, hi in b]
        assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize('total', [6, 7])          # equal and ragged shards
def test_two_rank_gather_equals_single_process(total):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29600 + total
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    full_r, full_o, steps, actions = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from env_cases import oracle_env
    orc = oracle_env('maxren_lv')
    for k in range(total):
        orc.reset(int(steps[k]))
        out = orc.step(actions[k])
        assert full_r[k] == out['reward']
        assert np.array_equal(full_o[k], out['obs'])
