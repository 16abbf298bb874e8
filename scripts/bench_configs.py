"""Throughput of the other BASELINE.json configurations on one GPU (developer/report
script; the contract benchmark is bench.py).  Prints one JSON line per config."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
import numpy as np, torch
from opfgym_amd import capi, envs

CONFIGS = {
    'config2_voltage_control_mv_urban': (envs.VoltageControl, dict(simbench_network_name='1-MV-urban--0-sw'), 8192),
    'config3_eco_dispatch_hv_mixed': (envs.EcoDispatch, dict(simbench_network_name='1-HV-mixed--0-sw'), 8192),
    'config4_qmarket_mv_urban_shard': (envs.QMarket, dict(simbench_network_name='1-MV-urban--0-sw'), 8192),
    'config5_n1_voltage_control_hv_urban': (envs.SecurityConstrainedVoltageControl,
                                            dict(simbench_network_name='1-HV-urban--0-sw',
                                                 n_minus_one_lines='first8_non_islanding'), 4096),
    'config1_max_renewable_lv_rural': (envs.MaxRenewable, dict(simbench_network_name='1-LV-rural1--0-sw',
                                                               min_sgen_power=0.005, min_storage_power=0.005), 8192),
}
only = sys.argv[1:] or list(CONFIGS)
for name in only:
    cls, kw, B = CONFIGS[name]
    kw = dict(kw)
    if kw.get('n_minus_one_lines') == 'first8_non_islanding':
        # contingencies = lines whose removal does not island (here: the first 8 of them)
        from opfgym_amd import grids
        from opfgym_amd.case import net_to_case
        from helpers import non_bridge_branches
        case_ = net_to_case(grids.get_grid(kw['simbench_network_name'])[0])
        ok = [int(case_.br_elem[b]) for b in non_bridge_branches(case_) if case_.br_kind[b] == 0]
        kw['n_minus_one_lines'] = tuple(ok[:8])
    env = cls(batch_size=B, device='cuda:0', seed=0, **kw)
    rng = np.random.default_rng(0)
    t0 = time.perf_counter(); env.reset(options={'step': rng.choice(env.train_steps, B)}); torch.cuda.synchronize()
    t_reset = time.perf_counter() - t0
    actions = torch.as_tensor(rng.random((B, env.n_actions)), device='cuda:0')
    for _ in range(2):
        out = env.step(actions)
    io = env._io(actions, False)
    ms = capi.C.c_float()
    reps = 10
    capi.check(capi.lib().opfx_time_steps(env._env_handle, B, capi.C.byref(io), capi.C.byref(env.solve_opts), reps,
                                          capi._stream(), capi.C.byref(ms)))
    k_ms = ms.value / reps
    info = env.plan.info
    n_solves = 1 + len(env.contingencies)
    print(json.dumps({'config': name, 'batch': B, 'nb': info['nb'], 'n_actions': env.n_actions,
                      'n_obs': env.n_obs_raw, 'solves_per_step': n_solves, 'kernel_ms': round(k_ms, 4),
                      'env_steps_per_s': round(B / k_ms * 1e3), 'nr_solves_per_s': round(B * n_solves / k_ms * 1e3),
                      'converged': float(out[4]['converged'].double().mean()),
                      'mean_it_base_case': float(out[4]['iterations'].double().mean()),
                      'first_reset_s': round(t_reset, 3), 'lds_bytes': None}))
    env.close()
