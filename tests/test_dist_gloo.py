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


class _OracleRows:
    """Per-rank environment for the ShardedBatch test: the CPU oracle behind the batched interface
    (reset(options={'step': [b]}) / step(actions [b, na]) -> torch tensors)."""

    def __init__(self, b):
        from env_cases import oracle_env
        self.orc = oracle_env('maxren_lv')
        self.b = b

    def reset(self, options):
        self.steps = options['step']

    def step(self, actions):
        outs = []
        for k in range(self.b):
            self.orc.reset(int(self.steps[k]))
            outs.append(self.orc.step(actions[k].numpy()))
        f64 = lambda key: torch.tensor(np.array([o[key] for o in outs]), dtype=torch.float64)
        flags = torch.tensor([bool(o['terminated']) for o in outs])
        return f64('obs'), f64('reward'), flags, torch.zeros_like(flags), {'cost': f64('cost')}


def _sharded_worker(rank, world, port, total, q):
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), here, os.path.join(here, 'golden')]
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from opfgym_amd.dist import ShardedBatch, init_from_env
    r_, w_, _ = init_from_env('gloo')
    assert (r_, w_) == (rank, world)
    sb = ShardedBatch(_OracleRows, total, rank, world, gather=('reward', 'obs', 'terminated', 'cost'))
    rng = np.random.default_rng(11)
    steps = rng.integers(2000, 30000, total)
    actions = torch.tensor(rng.random((total, 3)))
    sb.reset(options={'step': steps[sb.lo:sb.hi]})
    local, full = sb.step(actions[sb.lo:sb.hi])
    assert local[1].shape[0] == sb.hi - sb.lo and sum(sb.sizes) == total
    if rank == 0:
        q.put(({k: v.numpy() for k, v in full.items()}, steps, actions.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('total', [5, 8])
def test_sharded_batch_reassembles_the_full_batch(total):
    """`opfgym_amd.dist.ShardedBatch` (the wrapper bench.py's N > 1 path is built from): contiguous
    whole-instance shards, one all-gather per requested output, ragged shards included — every rank ends
    up with the rows a single process computes."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29700 + total
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    full, steps, actions = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = _OracleRows(total)
    ref.reset({'step': steps})
    obs, reward, term, _, info = ref.step(torch.tensor(actions))
    assert np.array_equal(full['reward'], reward.numpy())
    assert np.array_equal(full['obs'], obs.numpy())
    assert np.array_equal(full['terminated'], term.double().numpy())
    assert np.array_equal(full['cost'], info['cost'].numpy())


def _overlap_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from opfgym_amd.dist import OverlappedGather
    g = OverlappedGather(world)
    buf = torch.zeros(5, dtype=torch.float64)          # a persistent output buffer, overwritten every "step"
    got = []
    for k in range(6):
        buf.copy_(torch.arange(5, dtype=torch.float64) + 100 * k + 10 * rank)
        prev = g.submit(buf)
        got.append(None if prev is None else prev.clone())
    got.append(g.flush().clone())
    if rank == 0:
        q.put([None if t is None else t.numpy() for t in got])
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_gather_returns_the_previous_step():
    """`OverlappedGather`: step k's rows are gathered while step k+1 runs — `submit` returns the full batch of the
    step before (None at first), `flush` the last one; the producer's buffer may be overwritten right after submit."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_overlap_worker, args=(r, 2, 29655, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0] is None
    for k in range(6):
        want = np.concatenate([np.arange(5) + 100 * k + 10 * r for r in range(2)]).astype(float)
        assert np.array_equal(got[k + 1], want)


def _ragged_overlap_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from opfgym_amd.dist import OverlappedGather, shard_bounds
    total = 7                                             # 7 rows over 2 ranks: shards of 4 and 3
    sizes = [shard_bounds(total, r, world)[1] - shard_bounds(total, r, world)[0] for r in range(world)]
    lo, hi = shard_bounds(total, rank, world)
    g = OverlappedGather(world, sizes)
    got = []
    for k in range(4):
        local = (torch.arange(lo, hi, dtype=torch.float64) + 100 * k).reshape(-1, 1).repeat(1, 2)
        got.append(g.submit(local))
    got.append(g.flush())
    bad = None
    try:
        g.submit(torch.zeros(sizes[rank] + 1, 2, dtype=torch.float64))
    except ValueError as e:
        bad = str(e)
    if rank == 0:
        q.put(([None if t is None else t.numpy() for t in got], bad))
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_gather_with_ragged_shards():
    """a batch the world size does not divide (bench.py's strong-scaling configs on 3, 6 or 7 GPUs) gives
    shards of different sizes; the overlapped gather pads to the largest shard and trims on hand-over instead of
    issuing a collective with mismatched counts."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_ragged_overlap_worker, args=(r, 2, 29657, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, bad = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0] is None and bad is not None and 'shard has' in bad
    for k in range(4):
        want = (np.arange(7.0) + 100 * k).reshape(-1, 1).repeat(2, axis=1)
        assert got[k + 1].shape == (7, 2) and np.array_equal(got[k + 1], want)


def _forced_world_of_one(port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    from opfgym_amd import dist as od
    plain = od.OverlappedGather(1)
    x = torch.arange(6, dtype=torch.float64).reshape(3, 2)
    short = (od.init_from_env('gloo'), dist.is_initialized(), plain.submit(x) is x, plain.flush(), od.all_gather_rows(x, 1) is x)
    r, w, _ = od.init_from_env('gloo', force_collective=True)
    g = od.OverlappedGather(1)
    first = g.submit(x)
    second = g.submit(x + 10)
    last = g.flush()
    rows = od.all_gather_rows(x, 1)
    q.put((short[0][:2], short[1], short[2], short[3], short[4], (r, w), dist.is_initialized(), dist.get_world_size(), first,
           second.numpy(), last.numpy(), rows is x, rows.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_a_forced_world_of_one_runs_the_collectives_instead_of_short_circuiting():
    """`init_from_env(force_collective=True)` (bench.py's OPFX_FORCE_COLLECTIVE=1: how the RCCL branch is executed on a one-GPU box, tests/test_gpu_bench.py): a single rank
    initialises its process group and `OverlappedGather` / `all_gather_rows` go through the collective (staging buffers,
    previous-step hand-over) — without it the same calls hand the local tensor straight back."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_forced_world_of_one, args=(29671, q))
    p.start()
    got = q.get(timeout=120)
    p.join(timeout=60)
    assert p.exitcode == 0
    rw0, init0, same0, flush0, rows_same0, rw, init1, world, first, second, last, rows_same, rows = got
    assert rw0 == (0, 1) and not init0 and same0 and flush0 is None and rows_same0          # short circuit
    assert rw == (0, 1) and init1 and world == 1
    x = np.arange(6.0).reshape(3, 2)
    assert first is None and np.array_equal(second, x) and np.array_equal(last, x + 10)     # the previous step, then the last
    assert not rows_same and np.array_equal(rows, x)
