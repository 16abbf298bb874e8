"""Benchmark environments on the batched GPU backend.

Same problem definitions (grid preparation, action/observation keys, cost
tables, per-reset sampling tails) as the reference's
`opfgym/envs/{voltage_control,q_market,eco_dispatch,max_renewable}.py` and
`opfgym/examples/security_constrained.py`, expressed against
:class:`opfgym_amd.batched_env.BatchedOpfEnv`.  The `_sampling` tails become
vector ops executed by the reset kernel (`_sampling_ops`).
"""
from __future__ import annotations

import numpy as np

from . import net as ppn
from .batched_env import BatchedOpfEnv, MultiStageOpfEnv, OpsBuilder, SecurityConstrainedOpfEnv
from .simbench_build import build_simbench_net, gens_to_fixed_sgens, non_islanding_lines


class VoltageControl(BatchedOpfEnv):
    """voltage_control.py:8-133: reactive set-points of the bigger sgens and
    storages; loss + (optional) reactive market costs; voltage band, line/trafo
    loading and slack reactive exchange constraints."""

    def __init__(self, simbench_network_name='1-MV-semiurb--1-sw', load_scaling=1.5, gen_scaling=1.3,
                 cos_phi=0.95, max_q_exchange=0.5, min_sgen_power=0.5, min_storage_power=0.5,
                 market_based=False, *args, **kwargs):
        self.min_sgen_power, self.min_storage_power = min_sgen_power, min_storage_power
        self.cos_phi, self.market_based, self.max_q_exchange = cos_phi, market_based, max_q_exchange
        net, profiles = self._define_opf(simbench_network_name, gen_scaling=gen_scaling,
                                         load_scaling=load_scaling, *args, **kwargs)
        obs_keys = [('sgen', 'p_mw', net.sgen.index), ('storage', 'p_mw', net.storage.index),
                    ('load', 'p_mw', net.load.index), ('load', 'q_mvar', net.load.index)]      # :42-47
        if market_based:
            obs_keys.append(('poly_cost', 'cq2_eur_per_mvar2', net.poly_cost.index))           # :49-53
        act_keys = [('sgen', 'q_mvar', net.sgen.index[net.sgen.controllable]),
                    ('storage', 'q_mvar', net.storage.index[net.storage.controllable])]        # :56-57
        kwargs = {k: v for k, v in kwargs.items() if k not in ('net', 'profiles', 'grid_seed')}
        super().__init__(net, act_keys, obs_keys, profiles=profiles, *args, **kwargs)

    def _build_net(self, simbench_network_name, *args, **kwargs):
        """Grid + profiles the problem is defined on (hook for variants on other stand-in grids)."""
        return build_simbench_net(simbench_network_name, *args, **kwargs)

    def _define_opf(self, simbench_network_name, *args, **kwargs):
        net, profiles = self._build_net(simbench_network_name, *args, **kwargs)
        net.load['controllable'] = False
        net.sgen['controllable'] = net.sgen.max_max_p_mw > self.min_sgen_power                 # :68
        net.sgen['max_s_mva'] = net.sgen['max_max_p_mw'] / self.cos_phi                         # :70
        net.sgen['max_max_q_mvar'] = net.sgen['max_s_mva']
        net.sgen['min_min_q_mvar'] = -net.sgen['max_s_mva']
        net.storage['controllable'] = net.storage.max_max_p_mw > self.min_storage_power if len(net.storage) \
            else np.zeros(0, bool)
        net.storage['max_s_mva'] = net.storage['max_max_p_mw'].abs() if len(net.storage) else np.zeros(0)
        net.storage['max_max_q_mvar'] = net.storage['max_s_mva']
        net.storage['min_min_q_mvar'] = -net.storage['max_s_mva']
        net.ext_grid['max_q_mvar'] = self.max_q_exchange                                        # :80-81
        net.ext_grid['min_q_mvar'] = -self.max_q_exchange
        self.loss_costs = 0.03
        for idx in net.sgen.index[net.sgen.controllable]:                                       # :87-100
            ppn.create_poly_cost(net, idx, 'sgen', cp1_eur_per_mw=self.loss_costs, cq2_eur_per_mvar2=0)
        for idx in net.storage.index[net.storage.controllable.astype(bool)]:
            ppn.create_poly_cost(net, idx, 'storage', cp1_eur_per_mw=-self.loss_costs, cq2_eur_per_mvar2=0)
        for idx in net.ext_grid.index:
            ppn.create_poly_cost(net, idx, 'ext_grid', cp1_eur_per_mw=self.loss_costs, cq2_eur_per_mvar2=0)
        ppn.finalize(net)
        assert len(net.gen) == 0                                                                # :102
        self.max_price = 0.03
        net.poly_cost['min_cq2_eur_per_mvar2'] = 0
        net.poly_cost['max_cq2_eur_per_mvar2'] = self.max_price
        return net, profiles

    def _sampling_ops(self, ops: OpsBuilder) -> None:
        net = self.net
        if self.market_based:                                                                   # :115-119
            pc = net.poly_cost
            for unit_type in ('sgen', 'ext_grid', 'storage'):
                idx = pc[pc.et == unit_type].index
                rows = pc.index.get_indexer(idx)
                ops.uniform('poly_cost', 'cq2_eur_per_mvar2', idx,
                            pc['min_cq2_eur_per_mvar2'].to_numpy(float)[rows],
                            pc['max_cq2_eur_per_mvar2'].to_numpy(float)[rows])
        for unit_type in ('sgen', 'storage'):                                                   # :123-125
            if not len(net[unit_type]):
                continue
            sc = net[unit_type].scaling.to_numpy(float)
            ops.affine(unit_type, 'max_p_mw', 'p_mw', sc, 1e-9)
            ops.affine(unit_type, 'min_p_mw', 'p_mw', sc, -1e-9)
        for unit_type in ('sgen', 'storage'):                                                   # :128-133
            if not len(net[unit_type]):
                continue
            ops.sqrt_diff(unit_type, 'max_q_mvar', 'max_p_mw', net[unit_type].max_s_mva.to_numpy(float))
            ops.neg(unit_type, 'min_q_mvar', 'max_q_mvar')
            ops.set_const(unit_type, 'q_mvar', 0.0)


class QMarket(VoltageControl):
    """q_market.py:22-35: market-based VoltageControl with other defaults."""

    def __init__(self, simbench_network_name='1-MV-rural--0-sw', gen_scaling=1.0, load_scaling=1.5,
                 min_sgen_power=0.2, cos_phi=0.95, max_q_exchange=0.1, market_based=True,
                 *args, **kwargs):
        super().__init__(simbench_network_name=simbench_network_name, load_scaling=load_scaling,
                         gen_scaling=gen_scaling, cos_phi=cos_phi, max_q_exchange=max_q_exchange,
                         market_based=market_based, min_sgen_power=min_sgen_power, *args, **kwargs)


class EcoDispatch(BatchedOpfEnv):
    """eco_dispatch.py:7-123: active power of sgens/gens at sampled prices."""

    def __init__(self, simbench_network_name='1-HV-urban--0-sw', gen_scaling=1.0, load_scaling=1.5,
                 max_price_eur_gwh=0.5, min_power=0, *args, **kwargs):
        self.max_price_eur_gwh, self.min_power = max_price_eur_gwh, min_power
        net, profiles = self._define_opf(simbench_network_name, gen_scaling=gen_scaling,
                                         load_scaling=load_scaling, *args, **kwargs)
        obs_keys = [('load', 'p_mw', net.load.index), ('load', 'q_mvar', net.load.index),
                    ('poly_cost', 'cp1_eur_per_mw', net.poly_cost.index),
                    ('pwl_cost', 'cp1_eur_per_mw', net.pwl_cost.index),
                    ('sgen', 'p_mw', net.sgen.index[~net.sgen.controllable]),
                    ('storage', 'p_mw', net.storage.index), ('storage', 'q_mvar', net.storage.index)]  # :44-51
        act_keys = [('sgen', 'p_mw', net.sgen.index[net.sgen.controllable]),
                    ('gen', 'p_mw', net.gen.index[net.gen.controllable])]                       # :54-55
        kwargs = {k: v for k, v in kwargs.items() if k not in ('net', 'profiles', 'grid_seed')}
        super().__init__(net, act_keys, obs_keys, profiles=profiles, *args, **kwargs)

    def _define_opf(self, simbench_network_name, *args, **kwargs):
        net, profiles = build_simbench_net(simbench_network_name, *args, **kwargs)
        net.ext_grid['vm_pu'] = 1.0                                                             # :64-65
        net.gen['vm_pu'] = 1.0
        net.load['controllable'] = False
        net.ext_grid['min_p_mw'] = 0                                                            # :70-72
        net.ext_grid['max_p_mw'] = net.sgen.max_max_p_mw.max()
        net.sgen['min_p_mw'] = 0                                                                # :75-78
        net.sgen['max_p_mw'] = net.sgen['max_max_p_mw']
        net.gen['min_p_mw'] = 0
        net.gen['max_p_mw'] = net.gen['max_max_p_mw']
        net.sgen['controllable'] = net.sgen.max_max_p_mw > self.min_power                       # :81-83
        net.sgen['min_min_p_mw'] = 0
        net.gen['controllable'] = True
        for unit_type in ('gen', 'sgen'):                                                       # :86-88
            net[unit_type]['max_q_mvar'] = 0.0
            net[unit_type]['min_q_mvar'] = 0.0
        for idx in net.ext_grid.index:                                                          # :95-99
            ppn.create_pwl_cost(net, idx, 'ext_grid', points=[[0, 10000, 1]])
        for idx in net.sgen.index[net.sgen.controllable]:
            ppn.create_poly_cost(net, idx, 'sgen', cp1_eur_per_mw=0)
        for idx in net.gen.index[net.gen.controllable]:
            ppn.create_poly_cost(net, idx, 'gen', cp1_eur_per_mw=0)
        ppn.finalize(net)
        net.poly_cost['min_cp1_eur_per_mw'] = 0                                                 # :101-107
        net.poly_cost['max_cp1_eur_per_mw'] = self.max_price_eur_gwh
        net.pwl_cost['cp1_eur_per_mw'] = 0.0
        net.pwl_cost['min_cp1_eur_per_mw'] = 0
        net.pwl_cost['max_cp1_eur_per_mw'] = self.max_price_eur_gwh
        return net, profiles

    def _sampling_ops(self, ops: OpsBuilder) -> None:
        net = self.net                                                                          # :115-123
        for tbl in ('poly_cost', 'pwl_cost'):
            df = net[tbl]
            ops.uniform(tbl, 'cp1_eur_per_mw', df.index, df['min_cp1_eur_per_mw'].to_numpy(float),
                        df['max_cp1_eur_per_mw'].to_numpy(float))


class MaxRenewable(BatchedOpfEnv):
    """max_renewable.py:7-105: maximise renewable feed-in."""

    def __init__(self, simbench_network_name='1-HV-mixed--1-sw', gen_scaling=0.8, load_scaling=0.8,
                 min_storage_power=10, min_sgen_power=24, *args, **kwargs):
        self.min_sgen_power, self.min_storage_power = min_sgen_power, min_storage_power
        net, profiles = self._define_opf(simbench_network_name, gen_scaling=gen_scaling,
                                         load_scaling=load_scaling, *args, **kwargs)
        nctrl_st = net.storage.index[~net.storage.controllable.astype(bool)]
        obs_keys = [('sgen', 'max_p_mw', net.sgen.index), ('load', 'p_mw', net.load.index),
                    ('load', 'q_mvar', net.load.index), ('storage', 'p_mw', nctrl_st)]          # :38-43
        state_keys = [('sgen', 'p_mw', net.sgen.index), ('load', 'p_mw', net.load.index),
                      ('load', 'q_mvar', net.load.index), ('storage', 'p_mw', nctrl_st)]        # :46-51
        act_keys = [('sgen', 'p_mw', net.sgen.index[net.sgen.controllable]),
                    ('storage', 'p_mw', net.storage.index[net.storage.controllable.astype(bool)])]  # :54-57
        kwargs = {k: v for k, v in kwargs.items() if k not in ('net', 'profiles', 'grid_seed')}
        super().__init__(net, act_keys, obs_keys, state_keys=state_keys, profiles=profiles,
                         *args, **kwargs)

    def _define_opf(self, simbench_network_name, *args, **kwargs):
        net, profiles = build_simbench_net(simbench_network_name, *args, **kwargs)
        if len(net.ext_grid) > 1:                                                               # :67-69
            net.ext_grid = net.ext_grid.iloc[0:1]
        net.trafo['max_loading_percent'] = 100                                                  # :72
        net.load['controllable'] = False
        net.ext_grid['vm_pu'] = 1.0
        net.storage['controllable'] = net.storage.max_max_p_mw > self.min_storage_power if len(net.storage) \
            else np.zeros(0, bool)                                                              # :78-86
        net.storage['q_mvar'] = 0.0
        net.storage['max_q_mvar'] = 0.0
        net.storage['min_q_mvar'] = 0.0
        net.storage['max_p_mw'] = net.storage['max_max_p_mw'] if len(net.storage) else np.zeros(0)
        net.storage['min_p_mw'] = net.storage['min_min_p_mw'] if len(net.storage) else np.zeros(0)
        net.sgen['controllable'] = net.sgen.max_max_p_mw > self.min_sgen_power                  # :88-92
        net.sgen['min_p_mw'] = 0.0
        net.sgen['q_mvar'] = 0.0
        net.sgen['max_q_mvar'] = 0.0
        net.sgen['min_q_mvar'] = 0.0
        for idx in net.sgen.index:                                                              # :94-97
            ppn.create_poly_cost(net, idx, 'sgen', cp1_eur_per_mw=-30 / 1000)
        ppn.finalize(net)
        return net, profiles

    def _sampling_ops(self, ops: OpsBuilder) -> None:
        ops.affine('sgen', 'max_p_mw', 'p_mw', self.net.sgen.scaling.to_numpy(float), 1e-6)     # :105


class LoadShedding(BatchedOpfEnv):
    """load_shedding.py:15-145: active power of the bigger loads and storages at sampled
    shedding / storage prices; storage costs are piece-wise linear (efficiency)."""

    def __init__(self, simbench_network_name='1-MV-comm--2-sw', gen_scaling=1.6, load_scaling=2.2,
                 min_load_power=0.6, min_storage_power=1.0, max_p_exchange=8.0, storage_efficiency=0.95,
                 *args, **kwargs):
        self.min_load_power, self.min_storage_power = min_load_power, min_storage_power
        self.max_p_exchange, self.storage_efficiency = max_p_exchange, storage_efficiency
        net, profiles = self._define_opf(simbench_network_name, gen_scaling=gen_scaling,
                                         load_scaling=load_scaling, *args, **kwargs)
        nctrl_st = net.storage.index[~net.storage.controllable.astype(bool)]
        obs_keys = [('sgen', 'p_mw', net.sgen.index), ('load', 'max_p_mw', net.load.index),
                    ('load', 'q_mvar', net.load.index), ('storage', 'p_mw', nctrl_st),
                    ('poly_cost', 'cp1_eur_per_mw', net.poly_cost.index),
                    ('pwl_cost', 'cp1_eur_per_mw', net.pwl_cost.index)]                         # :46-53
        state_keys = [('sgen', 'p_mw', net.sgen.index), ('load', 'p_mw', net.load.index),
                      ('load', 'q_mvar', net.load.index), ('storage', 'p_mw', nctrl_st)]        # :56-63
        act_keys = [('load', 'p_mw', net.load.index[net.load.controllable]),
                    ('storage', 'p_mw', net.storage.index[net.storage.controllable.astype(bool)])]  # :66-67
        # sampled price -> the two segment prices (points updated per reset, :134-141)
        self.pwl_price_columns = {'neg_price_eur_per_mw': 0, 'pos_price_eur_per_mw': 1}
        kwargs = {k: v for k, v in kwargs.items() if k not in ('net', 'profiles', 'grid_seed')}
        super().__init__(net, act_keys, obs_keys, state_keys=state_keys, profiles=profiles, *args, **kwargs)

    def _define_opf(self, simbench_network_name, *args, **kwargs):
        net, profiles = build_simbench_net(simbench_network_name, *args, **kwargs)
        net.load['controllable'] = net.load.max_max_p_mw > self.min_load_power                  # :79-82
        net.load['min_min_p_mw'] = 0
        net.load['min_p_mw'] = 0
        max_storage_power = np.maximum(net.storage['min_min_p_mw'].abs(), net.storage['max_max_p_mw'].abs())
        net.storage['min_p_mw'] = -max_storage_power                                            # :85-91
        net.storage['max_p_mw'] = max_storage_power
        net.storage['min_min_p_mw'] = -max_storage_power
        net.storage['max_max_p_mw'] = max_storage_power
        net.storage['controllable'] = net.storage.max_max_p_mw > self.min_storage_power
        net.sgen['controllable'] = False
        net.ext_grid['max_p_mw'] = self.max_p_exchange                                          # :96-97
        net.ext_grid['min_p_mw'] = -np.inf
        for idx in net.load.index[net.load.controllable]:                                       # :99-105
            ppn.create_poly_cost(net, idx, 'load', cp1_eur_per_mw=0)
        for idx in net.storage.index[net.storage.controllable.astype(bool)]:
            ppn.create_pwl_cost(net, idx, 'storage', points=[[-1000, 0, 1], [0, 1000, 1]])
        ppn.finalize(net)
        net.poly_cost['min_cp1_eur_per_mw'] = -10                                               # :109-117
        net.poly_cost['max_cp1_eur_per_mw'] = 0
        net.pwl_cost['cp1_eur_per_mw'] = 0.0
        net.pwl_cost['min_cp1_eur_per_mw'] = 0
        net.pwl_cost['max_cp1_eur_per_mw'] = 2
        net.ext_grid['vm_pu'] = 1.0
        return net, profiles

    def _sampling_ops(self, ops: OpsBuilder) -> None:
        net = self.net
        for tbl in ('poly_cost', 'pwl_cost'):                                                   # :127-130
            df = net[tbl]
            ops.uniform(tbl, 'cp1_eur_per_mw', df.index, df['min_cp1_eur_per_mw'].to_numpy(float),
                        df['max_cp1_eur_per_mw'].to_numpy(float))
        if len(net.pwl_cost):                                                                   # :133-141
            ops.div('pwl_cost', 'pos_price_eur_per_mw', 'cp1_eur_per_mw', self.storage_efficiency)
            ops.affine('pwl_cost', 'neg_price_eur_per_mw', 'cp1_eur_per_mw', self.storage_efficiency, 0.0)
        ops.affine('load', 'max_p_mw', 'p_mw', net.load.scaling.to_numpy(float), 1e-9)           # :144
        for unit_type in ('load', 'storage'):                                                   # :147-149
            if len(net[unit_type]):
                sc = net[unit_type].scaling.to_numpy(float)
                ops.affine(unit_type, 'max_q_mvar', 'q_mvar', sc, 1e-9)
                ops.affine(unit_type, 'min_q_mvar', 'q_mvar', sc, -1e-9)


class SecurityConstrained(SecurityConstrainedOpfEnv):
    """examples/security_constrained.py:10-49: all sgen P as actions, loss cost at
    the slack, N-1 security for the listed lines."""

    def __init__(self, simbench_network_name='1-HV-urban--0-sw', n_minus_one_lines=(1, 3, 7),
                 *args, **kwargs):
        n_minus_one_keys = (('line', 'in_service', np.array(n_minus_one_lines)),)               # :13
        net, profiles = self._define_opf(simbench_network_name, *args, **kwargs)
        obs_keys = [('load', 'p_mw', net.load.index), ('load', 'q_mvar', net.load.index)]       # :20-23
        act_keys = [('sgen', 'p_mw', net.sgen.index)]                                           # :26
        kwargs = {k: v for k, v in kwargs.items() if k not in ('net', 'profiles', 'grid_seed')}
        super().__init__(net, act_keys, obs_keys, n_minus_one_keys=n_minus_one_keys, profiles=profiles,
                         *args, **kwargs)

    def _define_opf(self, simbench_network_name, *args, **kwargs):
        net, profiles = build_simbench_net(simbench_network_name, *args, **kwargs)
        net.sgen['controllable'] = True                                                         # :37-41
        net.sgen['max_p_mw'] = net.sgen['max_max_p_mw']
        net.sgen['min_p_mw'] = net.sgen['min_min_p_mw']
        net.sgen['max_q_mvar'] = 0
        net.sgen['min_q_mvar'] = 0
        for unit_type in ('load', 'gen', 'storage'):
            net[unit_type]['controllable'] = False
        for idx in net.ext_grid.index:                                                          # :48-49
            ppn.create_poly_cost(net, idx, 'ext_grid', cp1_eur_per_mw=0.01)
        ppn.finalize(net)
        return net, profiles


class MultiStageOpf(MultiStageOpfEnv):
    """examples/multi_stage.py:19-63: all sgen P as actions over several consecutive
    SimBench time steps, cost of the power drawn from the external grid."""

    def __init__(self, simbench_network_name='1-LV-urban6--0-sw', steps_per_episode=4, *args, **kwargs):
        net, profiles = self._define_opf(simbench_network_name, *args, **kwargs)
        obs_keys = [('load', 'p_mw', net.load.index), ('load', 'q_mvar', net.load.index)]       # :33-36
        act_keys = [('sgen', 'p_mw', net.sgen.index)]                                           # :39
        kwargs = {k: v for k, v in kwargs.items() if k not in ('net', 'profiles', 'grid_seed')}
        super().__init__(net, act_keys, obs_keys, profiles=profiles, steps_per_episode=steps_per_episode,
                         *args, **kwargs)

    def _define_opf(self, simbench_network_name, *args, **kwargs):
        net, profiles = build_simbench_net(simbench_network_name, *args, **kwargs)
        net.sgen['controllable'] = True                                                         # :50-54
        net.sgen['min_p_mw'] = net.sgen['min_min_p_mw']
        net.sgen['max_p_mw'] = net.sgen['max_max_p_mw']
        net.sgen['min_q_mvar'] = 0
        net.sgen['max_q_mvar'] = 0
        for unit_type in ('load', 'gen', 'storage'):
            net[unit_type]['controllable'] = False
        for idx in net.ext_grid.index:                                                          # :61-62
            ppn.create_poly_cost(net, idx, 'ext_grid', cp1_eur_per_mw=1)
        ppn.finalize(net)
        return net, profiles


class NetworkReconfiguration(BatchedOpfEnv):
    """examples/network_reconfiguration.py:16-72: selected line switches and all transformer tap
    changers as (discrete) actuators, loss cost at the slack.  The switch states and tap positions
    change Ybus VALUES per instance (opfx_env_desc.bmod_*); the plan is compiled once with every
    controllable switch closed."""

    def __init__(self, simbench_network_name='1-HV-urban--0-sw', controllable_switch_idxs=(1, 3),
                 *args, **kwargs):
        self.controllable_switch_idxs = np.array(controllable_switch_idxs)                      # :20
        net, profiles = self._define_opf(simbench_network_name, *args, **kwargs)
        obs_keys = [('sgen', 'p_mw', net.sgen.index), ('load', 'p_mw', net.load.index),         # :27-31
                    ('load', 'q_mvar', net.load.index)]
        act_keys = [('switch', 'closed', net.switch.index[net.switch.controllable.to_numpy(bool)]),   # :34-35
                    ('trafo', 'tap_pos', net.trafo.index[net.trafo.controllable.to_numpy(bool)])]
        kwargs = {k: v for k, v in kwargs.items() if k not in ('net', 'profiles', 'grid_seed')}
        super().__init__(net, act_keys, obs_keys, profiles=profiles, *args, **kwargs)

    def _define_opf(self, simbench_network_name, *args, **kwargs):
        net, profiles = build_simbench_net(simbench_network_name, *args, **kwargs)
        net.switch['controllable'] = False                                                      # :45-46
        net.switch.loc[self.controllable_switch_idxs, 'controllable'] = True
        net.switch['min_closed'] = 0                                                            # :49-53
        net.switch['max_closed'] = 1
        net.switch['min_min_closed'] = 0
        net.switch['max_max_closed'] = 1
        net.trafo['controllable'] = True                                                        # :56-60
        net.trafo['min_tap_pos'] = -1
        net.trafo['max_tap_pos'] = 1
        net.trafo['min_min_tap_pos'] = -1
        net.trafo['max_max_tap_pos'] = 1
        for unit_type in ('load', 'sgen', 'gen', 'storage'):                                    # :63-64
            net[unit_type]['controllable'] = False
        for idx in net.ext_grid.index:                                                          # :67-68
            ppn.create_poly_cost(net, idx, 'ext_grid', cp1_eur_per_mw=1)
        ppn.finalize(net)
        return net, profiles


class MixedContinuousDiscrete(BatchedOpfEnv):
    """examples/mixed_continuous_discrete.py:22-104: reactive power of all sgens (continuous) and the
    transformer taps (discrete) as actuators, quadratic voltage deviation as objective, slack
    voltage sampled per instance."""

    def __init__(self, simbench_network_name='1-LV-urban6--0-sw', cos_phi=0.95, *args, **kwargs):
        from .objectives import QuadraticDeviation
        self.cos_phi = cos_phi                                                                   # :26
        net, profiles = self._define_opf(simbench_network_name, *args, **kwargs)
        obs_keys = [('ext_grid', 'vm_pu', net.ext_grid.index), ('sgen', 'p_mw', net.sgen.index),  # :32-37
                    ('load', 'p_mw', net.load.index), ('load', 'q_mvar', net.load.index)]
        act_keys = [('sgen', 'q_mvar', net.sgen.index), ('trafo', 'tap_pos', net.trafo.index)]   # :40-41
        kwargs = {k: v for k, v in kwargs.items() if k not in ('net', 'profiles', 'grid_seed')}
        objective = kwargs.pop('objective_function', None) or QuadraticDeviation('bus', 'vm_pu', 1.0)   # :17-19,44
        super().__init__(net, act_keys, obs_keys, profiles=profiles, objective_function=objective, *args, **kwargs)

    def _define_opf(self, simbench_network_name, *args, **kwargs):
        net, profiles = build_simbench_net(simbench_network_name, *args, **kwargs)
        net.trafo['controllable'] = True                                                        # :53-55
        net.trafo['min_tap_pos'] = -2
        net.trafo['max_tap_pos'] = 2
        net.sgen['controllable'] = True                                                         # :58-64
        net.sgen['max_s_mva'] = net.sgen['max_max_p_mw'] / self.cos_phi
        net.sgen['max_max_q_mvar'] = (net.sgen['max_s_mva'] ** 2 - net.sgen['max_max_p_mw'] ** 2) ** 0.5
        net.sgen['min_min_q_mvar'] = -net.sgen['max_max_q_mvar']
        net.sgen['max_q_mvar'] = net.sgen['max_max_q_mvar']
        net.sgen['min_q_mvar'] = -net.sgen['max_max_q_mvar']
        for unit_type in ('load', 'gen', 'storage'):                                            # :67-68
            net[unit_type]['controllable'] = False
        net.ext_grid['min_vm_pu'] = 0.95                                                        # :71-72
        net.ext_grid['max_vm_pu'] = 1.05
        ppn.finalize(net)
        return net, profiles

    def _sampling_ops(self, ops: OpsBuilder) -> None:
        net = self.net
        eg = net.ext_grid                                                                       # :80-84
        ops.uniform('ext_grid', 'vm_pu', eg.index, eg['min_vm_pu'].to_numpy(float), eg['max_vm_pu'].to_numpy(float))
        sc = net.sgen.scaling.to_numpy(float)                                                   # :88-90
        ops.affine('sgen', 'max_p_mw', 'p_mw', sc, 1e-9)
        ops.affine('sgen', 'min_p_mw', 'p_mw', sc, -1e-9)


class ConstraintSatisfaction(BatchedOpfEnv):
    """examples/pure_constraint_satisfaction.py:8-53: no objective at all; sgen active power must
    keep the slack import, a tight voltage band and the line loadings within bounds."""

    def __init__(self, **kwargs):
        net, profiles = self._define_opf(**{k: kwargs[k] for k in ('grid_seed',) if k in kwargs})
        obs_keys = [('load', 'p_mw', net.load.index), ('load', 'q_mvar', net.load.index)]       # :15-18
        act_keys = [('sgen', 'p_mw', net.sgen.index)]                                           # :21
        kwargs = {k: v for k, v in kwargs.items() if k not in ('net', 'profiles', 'grid_seed')}
        super().__init__(net, act_keys, obs_keys, profiles=profiles, **kwargs)

    def _define_opf(self, **kwargs):
        net, profiles = build_simbench_net('1-LV-rural1--0-sw', **kwargs)                         # :26
        net.sgen['controllable'] = True                                                         # :28-32
        net.sgen['min_p_mw'] = 0
        net.sgen['max_p_mw'] = net.sgen['max_max_p_mw']
        net.sgen['min_q_mvar'] = 0
        net.sgen['max_q_mvar'] = 0
        for unit_type in ('load', 'gen', 'storage'):                                            # :35-36
            net[unit_type]['controllable'] = False
        net.ext_grid['max_p_mw'] = 1                                                            # :39-42
        net.bus['max_vm_pu'] = 1.02
        net.bus['min_vm_pu'] = 0.98
        net.line['max_loading_percent'] = 60
        ppn.finalize(net)
        return net, profiles


class PartiallyObservable(BatchedOpfEnv):
    """examples/partial_obs.py:13-65: only some loads are observed; the state keys (what is sampled)
    still cover all of them."""

    def __init__(self, simbench_network_name='1-LV-rural1--0-sw', observable_loads=np.arange(10),
                 *args, **kwargs):
        net, profiles = self._define_opf(simbench_network_name, *args, **kwargs)
        if isinstance(observable_loads, str) and observable_loads == 'all':                      # :21-22
            observable_loads = net.load.index
        obs_keys = [('load', 'p_mw', observable_loads), ('load', 'q_mvar', observable_loads)]    # :26-29
        state_keys = [('load', 'p_mw', net.load.index), ('load', 'q_mvar', net.load.index)]      # :33-36
        act_keys = [('sgen', 'p_mw', net.sgen.index)]                                           # :39
        kwargs = {k: v for k, v in kwargs.items() if k not in ('net', 'profiles', 'grid_seed')}
        super().__init__(net, act_keys, obs_keys, state_keys=state_keys, profiles=profiles, *args, **kwargs)

    def _define_opf(self, simbench_network_name, *args, **kwargs):
        net, profiles = build_simbench_net(simbench_network_name, *args, **kwargs)
        net.sgen['controllable'] = True                                                         # :48-52
        net.sgen['min_p_mw'] = 0
        net.sgen['max_p_mw'] = net.sgen['max_max_p_mw']
        net.sgen['min_q_mvar'] = 0
        net.sgen['max_q_mvar'] = 0
        for unit_type in ('load', 'gen', 'storage'):                                            # :55-56
            net[unit_type]['controllable'] = False
        for idx in net.ext_grid.index:                                                          # :59-60
            ppn.create_poly_cost(net, idx, 'ext_grid', cp1_eur_per_mw=1)
        ppn.finalize(net)
        return net, profiles


class NonSimbenchNet(BatchedOpfEnv):
    """examples/non_simbench_net.py:13-67: a pandapower OPF case without time series: generator
    active power as actions, load states drawn from a normal distribution around the case values
    (`std_dev_*` columns).  The reference loads `pp.networks.case_ieee30()`; that data set is not
    available offline, so the default here is the OPF-ready 9-bus case (`net=` takes any other)."""

    def __init__(self, train_data='normal_around_mean', test_data='normal_around_mean', net=None,
                 *args, **kwargs):
        assert 'simbench' not in train_data and 'simbench' not in test_data                     # :18
        net = self._define_opf(net)
        obs_keys = [('load', 'p_mw', net.load.index), ('load', 'q_mvar', net.load.index)]       # :24-27
        act_keys = [('gen', 'p_mw', net.gen.index)]                                             # :30
        kwargs = {k: v for k, v in kwargs.items() if k not in ('profiles', 'grid_seed')}
        super().__init__(net, act_keys, obs_keys, train_data=train_data, test_data=test_data, *args, **kwargs)

    def _define_opf(self, net=None):
        from . import grids
        net = net if net is not None else grids.case9_opf()                                      # :38
        net.gen['min_q_mvar'] = 0                                                               # :41-42
        net.gen['max_q_mvar'] = 0
        rng_ = 0.3                                                                              # :45-50
        net.load['min_min_p_mw'] = net.load['p_mw'] * (1 - rng_)
        net.load['max_max_p_mw'] = net.load['p_mw'] * (1 + rng_)
        net.load['min_min_q_mvar'] = net.load['q_mvar'] * (1 - rng_)
        net.load['max_max_q_mvar'] = net.load['q_mvar'] * (1 + rng_)
        net.load['mean_p_mw'] = net.load['p_mw']                                                # :53-56
        net.load['std_dev_p_mw'] = rng_ * net.load['p_mw']
        net.load['mean_q_mvar'] = net.load['q_mvar']
        net.load['std_dev_q_mvar'] = rng_ * net.load['q_mvar']
        net.ext_grid['mean_p_mw'] = net.load['mean_p_mw'].sum() - net.gen['p_mw'].sum()         # :59-60
        net.ext_grid['mean_q_mvar'] = net.load['mean_q_mvar'].sum() - (net.gen['max_q_mvar'] - net.gen['max_q_mvar']).sum()
        ppn.finalize(net)
        return net


class AddCustomConstraint(BatchedOpfEnv):
    """examples/custom_constraint.py:19-74: sgen reactive power as actions, the default constraints
    plus an apparent-power limit per sgen.  (The reference example hands its list over as
    `constraints=...`, a keyword `OpfEnv.__init__` swallows, so its custom constraint is never active;
    here it goes to `custom_constraints`, as the example intends.)"""

    def __init__(self, simbench_network_name='1-LV-urban6--0-sw', cos_phi=0.95, constraint_kwargs=None,
                 custom_constraint=None, *args, **kwargs):
        """`custom_constraint`: replaces the example's apparent-power constraint object (e.g. by one with a
        Python value callable, which then runs through the host fallback)."""
        from . import constraints as cons
        self.cos_phi = cos_phi
        net, profiles = self._define_opf(simbench_network_name, *args, **kwargs)
        obs_keys = [('load', 'p_mw', net.load.index), ('load', 'q_mvar', net.load.index)]       # :27-30
        act_keys = [('sgen', 'q_mvar', net.sgen.index)]                                         # :32
        constraint_kwargs = constraint_kwargs or {}
        constraints_list = cons.create_default_constraints(net, constraint_kwargs)               # :35-37
        constraints_list.append(custom_constraint or cons.Constraint(                           # :40-45
            'sgen', 's_mva', get_values=cons.ApparentPower('sgen'),
            get_boundaries=lambda net_: {'max': net_.sgen.max_max_p_mw / 0.95}, **constraint_kwargs))
        kwargs = {k: v for k, v in kwargs.items() if k not in ('net', 'profiles', 'grid_seed')}
        super().__init__(net, act_keys, obs_keys, profiles=profiles, custom_constraints=constraints_list,
                         *args, **kwargs)

    def _define_opf(self, simbench_network_name, *args, **kwargs):
        net, profiles = build_simbench_net(simbench_network_name, *args, **kwargs)
        net.sgen['controllable'] = True                                                         # :57-59
        net.sgen['min_q_mvar'] = -0.3
        net.sgen['max_q_mvar'] = 0.3
        net.sgen['max_s_mva'] = net.sgen['max_max_p_mw'] / self.cos_phi                          # :62
        for unit_type in ('load', 'gen', 'storage'):                                            # :65-66
            net[unit_type]['controllable'] = False
        for idx in net.ext_grid.index:                                                          # :68-69
            ppn.create_poly_cost(net, idx, 'ext_grid', cp1_eur_per_mw=1)
        ppn.finalize(net)
        return net, profiles

    def _sampling_ops(self, ops: OpsBuilder) -> None:
        sc = self.net.sgen.scaling.to_numpy(float)                                              # :79-80
        ops.affine('sgen', 'max_p_mw', 'p_mw', sc, 1e-9)
        ops.affine('sgen', 'min_p_mw', 'p_mw', sc, -1e-9)


class SecurityConstrainedVoltageControl(VoltageControl):
    """BASELINE config 5: VoltageControl problem definition with the N-1 wrapper
    of security_constrained.py (no such class in the reference; composed as
    SURVEY.md Appendix A.5 describes).  `n_minus_one_lines='all'`: every in-service line whose outage
    does not island."""

    def __init__(self, simbench_network_name='1-HV-urban--0-sw', n_minus_one_lines=(1, 3, 7),
                 not_converged_penalty=1, *args, **kwargs):
        def keys(net):
            lines = non_islanding_lines(net) if isinstance(n_minus_one_lines, str) else np.array(n_minus_one_lines)
            return (('line', 'in_service', lines),)
        super().__init__(simbench_network_name, *args, n_minus_one_keys=keys,
                         not_converged_penalty=not_converged_penalty, **kwargs)

    def _build_net(self, simbench_network_name, *args, **kwargs):
        net, profiles = build_simbench_net(simbench_network_name, *args, **kwargs)
        gens_to_fixed_sgens(net, profiles)
        return net, profiles
