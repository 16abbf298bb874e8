"""GPU: device-memory footprint and lifetime of the objects behind the C ABI (`opfx_ctx`, `opfx_env`, the reset
programme) and of the two caches above it — the topology twins of a bus-bus-switch environment
(`BatchedOpfEnv._topology_variant`) and the plug-in's per-topology plan cache (`BatchedPowerFlowSolver._cache`).
Free device bytes are read with hipMemGetInfo (`torch.cuda.mem_get_info`) after emptying torch's caching allocator, so
that what is counted is what the library itself and the live tensors hold."""
import gc

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
MB = 1 << 20


def _free_bytes():
    import torch
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    return torch.cuda.mem_get_info(0)[0]


def test_300_create_destroy_cycles_leave_device_memory_flat():
    """opfx_ctx_create -> opfx_env_create -> opfx_env_set_reset -> one reset + step -> opfx_env_destroy ->
    opfx_ctx_destroy, 300 times on one compiled plan: the free device memory afterwards is what it was after the first
    cycles, within 1 MB (a leaked arena, queue counter or event pair per cycle would show as 300 of them)."""
    import torch
    from opfgym_amd import envs
    env = envs.VoltageControl(simbench_network_name='mv-small', batch_size=64, defer_device=True, seed=0)
    actions = None

    def cycle():
        nonlocal actions
        env.attach_device()
        if actions is None:
            actions = torch.rand(64, env.n_actions, dtype=torch.float64, device='cuda:0')
        env.reset()
        _, reward, _, _, info = env.step(actions)
        conv = info['converged'].bool()
        ok = float(conv.double().mean()) > 0.9 and bool(torch.isfinite(reward[conv]).all())
        env.close()
        env.ctx = None                                   # (Context.__del__ -> opfx_ctx_destroy)
        env.buf, env.x = {}, None
        return ok

    for _ in range(5):
        assert cycle()
    free0 = _free_bytes()
    for _ in range(300):
        assert cycle()
    free1 = _free_bytes()
    assert abs(free0 - free1) <= MB, (free0 - free1) / MB


def test_topology_twins_have_a_bounded_footprint_and_are_built_once():
    """Six busbar couplers as actuators = up to 64 topologies, each a twin environment (own case, plan, context,
    descriptor, batch-1 buffers).  Stepping a batch through all 64: the device bytes per twin are reported and bounded,
    and a second pass over the same 64 topologies builds nothing and allocates nothing."""
    import torch
    from opfgym_amd import envs
    from opfgym_amd.batched_env import BatchedOpfEnv
    from test_gpu_env import _split_busbars

    class Couplers(BatchedOpfEnv):
        def __init__(self, **kw):
            base = envs.NetworkReconfiguration(simbench_network_name='hv-small-sw', batch_size=1, defer_device=True)
            net, profiles = base.definition.net, base.definition.profiles
            couplers = _split_busbars(net, [2, 3, 4, 7, 8, 9])
            net.switch['controllable'] = False
            for col, v in (('controllable', True), ('min_closed', 0), ('max_closed', 1), ('min_min_closed', 0), ('max_max_closed', 1)):
                net.switch.loc[couplers, col] = v
            obs_keys = [('load', 'p_mw', net.load.index), ('res_bus', 'vm_pu', net.bus.index)]
            act_keys = [('switch', 'closed', net.switch.index[net.switch.controllable.to_numpy(bool)])]
            BatchedOpfEnv.__init__(self, net, act_keys, obs_keys, profiles=profiles, **kw)

    B = 128
    env = Couplers(batch_size=B, device='cuda:0', seed=1)
    assert len(env._bb_switches) == 6 and env.n_actions == 6
    states = np.array([[(k >> j) & 1 for j in range(6)] for k in range(B)], dtype=float)       # every topology, twice
    actions = torch.as_tensor(0.9 * states + 0.05, device='cuda:0')
    env.reset()
    free0 = _free_bytes()
    out = env.step(actions)
    torch.cuda.synchronize()
    assert len(env._topology_variants) == 64
    free1 = _free_bytes()
    per_twin = (free0 - free1) / 64
    print(f'\n{per_twin / 1024:.0f} KiB of device memory per topology twin (40-bus grid, 64 twins)')
    # (a twin holds three device arenas — context, environment, reset programme — and hipMalloc hands memory out in 2 MB
    #  granules on this stack: 6 MB per twin although the arrays in them are a few hundred KB; before the arenas were
    #  chunked it was 14 MB, one granule share per array)
    assert 0 < per_twin <= 8 * MB, per_twin / MB
    conv1, rew1 = out[4]['converged'].clone(), out[1].clone()
    assert int(conv1.sum()) >= B // 2                                       # (most coupler settings leave a solvable grid)
    twins = {k: id(v) for k, v in env._topology_variants.items()}
    env.reset()
    out2 = env.step(actions)
    torch.cuda.synchronize()
    assert {k: id(v) for k, v in env._topology_variants.items()} == twins       # nothing rebuilt
    assert abs(_free_bytes() - free1) <= MB                                      # nothing allocated
    assert torch.equal(out2[4]['converged'], conv1)
    env.close()
    del env, out, out2
    assert _free_bytes() >= free0 - MB                                           # close() gave every twin back


def test_plugin_plan_cache_holds_256_topologies_and_frees_the_rest():
    """The batch-1 plug-in keeps one compiled plan + context per topology (the reference's N-1 loop comes back to each
    one every step); 700 distinct switch states go through it: 256 are held, the 444 oldest were destroyed — the device
    memory after 700 is what it was after 256 to within a few allocator granules (444 leaked contexts would be ~100 MB)."""
    from opfgym_amd import grids
    from opfgym_amd.solver_plugin import BatchedPowerFlowSolver
    net = grids.get_grid('hv-small-sw')[0]
    solver = BatchedPowerFlowSolver()
    sw = list(net.switch.index[:10])                                # 2^10 = 1024 states of ten line switches
    free_at = {}
    free_start = _free_bytes()
    n_ok = 0
    for k in range(700):
        net.switch.loc[sw, 'closed'] = [bool((k >> j) & 1) for j in range(10)]
        try:
            solver(net)
            n_ok += 1
        except Exception as exc:                                    # (a switch state may island load: not this test's subject)
            assert 'converge' in str(exc).lower(), exc
        if k + 1 in (256, 700):
            free_at[k + 1] = _free_bytes()
    assert len(solver._cache) == 256 and n_ok >= 350
    per_plan = (free_start - free_at[256]) / 256
    print(f'\n{per_plan / 1024:.0f} KiB of device memory per cached plan (40-bus grid)')
    assert 0 < per_plan <= 2 * MB
    # (the evicted contexts were destroyed: what 700 plans left behind is what 256 hold, to within a few 2 MB granules of the
    #  allocator — a leak of the 444 evicted ones would be 444 x per_plan)
    assert abs(free_at[256] - free_at[700]) <= 16 * MB < 0.25 * 444 * per_plan, (free_at[256] - free_at[700]) / MB
    solver._cache.clear()
    assert _free_bytes() >= free_start - 4 * MB
