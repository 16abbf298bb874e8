"""Scenario table shared by the golden-vector generator (runs the reference's
environment classes under stubs, in the build container only) and by the parity
tests (run this repo's oracle / GPU path on the same inputs)."""

# name -> (environment class name, constructor kwargs, number of samples, seed)
SCENARIOS = {
    'vc_mv_small': ('VoltageControl', dict(simbench_network_name='mv-small'), 6, 1),
    'vc_mv_urban': ('VoltageControl', dict(simbench_network_name='1-MV-urban--0-sw'), 16, 2),
    'qm_mv_small': ('QMarket', dict(simbench_network_name='mv-small'), 6, 3),
    'qm_mv_urban': ('QMarket', dict(simbench_network_name='1-MV-urban--0-sw'), 16, 4),
    'eco_hv_small': ('EcoDispatch', dict(simbench_network_name='hv-small'), 6, 5),
    'maxren_lv': ('MaxRenewable', dict(simbench_network_name='1-LV-rural1--0-sw',
                                       min_sgen_power=0.005, min_storage_power=0.005), 6, 6),
    'sc_hv_small': ('SecurityConstrained', dict(simbench_network_name='hv-small'), 5, 7),
    'loadshed_mv_small': ('LoadShedding', dict(simbench_network_name='mv-small', min_load_power=0.8,
                                               min_storage_power=0.3, max_p_exchange=6.0), 5, 18),
    # option coverage on the small MV grid
    'vc_replacement': ('VoltageControl', dict(
        simbench_network_name='mv-small', reward_function='replacement',
        reward_function_params=dict(valid_reward=0.7, penalty_weight=0.3, clip_range=(-1.5, 1.0))), 8, 8),
    'vc_parameterized': ('VoltageControl', dict(
        simbench_network_name='mv-small', reward_function='parameterized',
        reward_function_params=dict(valid_reward=0.4, invalid_penalty=0.2, invalid_objective_share=0.5,
                                    penalty_weight=None),
        constraint_params=dict(penalty_factor=2.0, penalty_power=1.5, violation_count_penalty=0.1,
                               only_worst_case_violations=True)), 8, 9),
    'vc_resobs_diff': ('VoltageControl', dict(
        simbench_network_name='mv-small', add_res_obs=True, diff_objective=True, add_act_obs=True,
        add_mean_obs=True, clipped_action_penalty=0.5), 4, 10),
    'vc_noscale_scaled': ('VoltageControl', dict(
        simbench_network_name='mv-small', autoscale_actions=False, voltage_band=0.02, max_loading=40,
        reward_function='summation',
        reward_function_params=dict(reward_scaling='minmax11', scaling_params=dict(
            min_objective=-2.0, max_objective=0.5, min_penalty=-3.0, max_penalty=0.0))), 4, 11),
    'vc_noisy': ('VoltageControl', dict(
        simbench_network_name='mv-small', train_data='noisy_simbench',
        sampling_params=dict(noise_factor=0.2)), 4, 12),
    'vc_full_uniform': ('VoltageControl', dict(
        simbench_network_name='mv-small', train_data='full_uniform', test_data='full_uniform'), 4, 13),
    'vc_normal_noise': ('VoltageControl', dict(
        simbench_network_name='mv-small', train_data='noisy_simbench',
        sampling_params=dict(noise_factor=0.15, noise_distribution='normal')), 4, 15),
    'vc_interpolate': ('VoltageControl', dict(
        simbench_network_name='mv-small', sampling_params=dict(interpolate_steps=True)), 4, 16),
    'vc_normal_mean': ('VoltageControl', dict(
        simbench_network_name='mv-small', train_data='normal_around_mean', test_data='normal_around_mean',
        sampling_params=dict(relative_std=0.3)), 4, 17),
    # autoscale_violation=False: ext-grid violations scaled by 1/|mean| (constraints.py:179-192), the others not
    'vc_no_autoscale': ('VoltageControl', dict(
        simbench_network_name='mv-small', max_q_exchange=0.05,
        constraint_params=dict(autoscale_violation=False)), 4, 30),
    # train_data='mixed' (opf_env.py:242-251) with the source forced by the probabilities, one scenario each
    'vc_mixed_simbench': ('VoltageControl', dict(
        simbench_network_name='mv-small', train_data='mixed', test_data='mixed',
        sampling_params=dict(data_probabilities=(1.0, 1.0, 1.0))), 3, 27),
    'vc_mixed_uniform': ('VoltageControl', dict(
        simbench_network_name='mv-small', train_data='mixed', test_data='mixed',
        sampling_params=dict(data_probabilities=(0.0, 1.0, 1.0))), 3, 28),
    'vc_mixed_normal': ('VoltageControl', dict(
        simbench_network_name='mv-small', train_data='mixed', test_data='mixed',
        sampling_params=dict(data_probabilities=(0.0, 0.0, 1.0), relative_std=0.25)), 3, 29),
    # multi-step episodes with incremental actions (opf_env.py:451-458, 406-414)
    'vc_multistep_diff': ('VoltageControl', dict(
        simbench_network_name='mv-small', steps_per_episode=3, diff_action_step_size=0.2), 3, 14),
}

SCENARIOS.update({
    # observation variants (opf_env.py:535-536, 806-810)
    'vc_bus_wise': ('VoltageControl', dict(simbench_network_name='mv-small', bus_wise_obs=True,
                                           add_mean_obs=True), 3, 19),
    # multi-stage episodes over consecutive time steps (multi_stage.py, examples/multi_stage.py)
    'multistage_lv': ('MultiStageOpf', dict(simbench_network_name='1-LV-rural1--0-sw', steps_per_episode=4), 4, 20),
})

SCENARIOS.update({
    # discrete actuators that change Ybus per instance: line switches + transformer taps
    # (examples/network_reconfiguration.py; opf_env.py:476-481)
    'reconf_hv_small_sw': ('NetworkReconfiguration', dict(simbench_network_name='hv-small-sw',
                                                          controllable_switch_idxs=(1, 3)), 8, 21),
    # the same plus three shunts in steps as actuators (('shunt', 'step'): rounded like tap positions, opf_env.py:476-481)
    # two busbar couplers (bus-bus switches 44, 45 of the prepared grid) among the controllable switches: four topologies
    'busbar_hv_small_sw': ('BusbarCouplers', dict(simbench_network_name='hv-small-sw',
                                                  controllable_switch_idxs=(1, 3, 44, 45)), 12, 44),
    'shunt_hv_small_sw': ('SwitchedShunts', dict(simbench_network_name='hv-small-sw',
                                                 controllable_switch_idxs=(1, 3)), 10, 43),
    # continuous + discrete actuators, objective_function seam, per-instance slack voltage
    # (examples/mixed_continuous_discrete.py)
    'mixed_lv': ('MixedContinuousDiscrete', dict(simbench_network_name='1-LV-rural1--0-sw'), 8, 22),
    # the remaining example environments of the reference (examples/*.py)
    'constraint_sat_lv': ('ConstraintSatisfaction', dict(), 5, 23),
    'partial_obs_lv': ('PartiallyObservable', dict(simbench_network_name='1-LV-rural1--0-sw'), 4, 24),
    'nonsimbench_case9': ('NonSimbenchNet', dict(), 5, 25),
    # custom constraint with value/boundary callables (examples/custom_constraint.py, constraints.py:62-65)
    'custom_constraint_lv': ('AddCustomConstraint', dict(simbench_network_name='1-LV-rural1--0-sw'), 6, 26),
})

SCENARIOS.update({
    # BASELINE configs 3 and 5 on their own grids: EcoDispatch on the 306-bus meshed HV
    # stand-in (wave teams of two), N-1 VoltageControl on the 372-bus stand-in with every non-islanding
    # line as contingency (250 of them; wave teams of four)
    'eco_hv_mixed': ('EcoDispatch', dict(simbench_network_name='1-HV-mixed--0-sw'), 16, 31),
    # a grid with a three-winding transformer: Trafo3wOverloadConstraint among the defaults (constraints.py:164-172,210)
    'vc_mv_3w': ('VoltageControl', dict(simbench_network_name='mv-3w'), 5, 33),
    'sc_vc_hv_urban': ('SecurityConstrainedVoltageControl', dict(simbench_network_name='1-HV-urban--0-sw',
                                                                 n_minus_one_lines='all'), 6, 32),
    # the same N-1 problem next to voltage collapse (loads scaled 4.6 instead of 1.5: the base case still solves, a part of
    # the 250 contingency cases does not): rows with FAILED contingencies — the +penalty sign of defect D6, `valids` all
    # False, security_constrained.py:59-63 — recorded from the reference itself (FAILED_CONTINGENCY_ROWS below)
    'sc_vc_hv_urban_stress': ('SecurityConstrainedVoltageControl', dict(simbench_network_name='1-HV-urban--0-sw',
                                                                        n_minus_one_lines='all', load_scaling=4.6), 3, 36),
    # the same N-1 problem with a reactive exchange band an HV grid can meet (the reference's default of +-0.5 Mvar is an
    # MV figure: on the 372-bus grid no action keeps all 251 cases inside it): valid AND invalid states of config 5's kind
    'sc_vc_hv_urban_wide': ('SecurityConstrainedVoltageControl', dict(simbench_network_name='1-HV-urban--0-sw',
                                                                      n_minus_one_lines='all', max_q_exchange=150.0), 4, 34),
})

SCENARIOS.update({
    # generators SHARING buses (two and three on one bus, one out of service, one beside the ext_grid) with different
    # reactive ranges and reactive prices on their cost rows: `res_gen.q_mvar` per generator as pypower's pfsoln splits
    # the bus total (objective.py:48-54), limits binding under enforce_q_lims
    'eco_hv_small_shared': ('EcoDispatchSharedBus', dict(simbench_network_name='hv-small'), 8, 51),
})

# Fixtures that must hold all-valid AND invalid states (the reward classes' `valid` branch, reward.py:246-252,254-305,
# needs reference-generated rows of both kinds): name -> (all-valid rows among the n samples, action levels searched).
# Random actions practically never give a valid state (the ext-grid band is narrow), so the generator looks for one
# along ONE scalar: the action level a = clip(level + 0.1 (u - 0.5)), u the sample's own uniform draw — the first level
# of the list whose step the reference itself reports as all-valid is recorded; a candidate without one is dropped.
def _levels(lo, hi, step):
    return [round(lo + step * k, 6) for k in range(int(round((hi - lo) / step)) + 1)]


VALID_ROWS = {
    'vc_mv_urban': (8, _levels(0.40, 0.70, 0.005)),
    'qm_mv_urban': (8, _levels(0.40, 0.70, 0.005)),
    'eco_hv_mixed': (8, _levels(0.0, 1.0, 0.0125)),
    'vc_replacement': (4, _levels(0.30, 0.70, 0.01)),
    'vc_parameterized': (4, _levels(0.30, 0.70, 0.01)),
    'sc_vc_hv_urban_wide': (2, [0.5, 0.49, 0.51]),
}

# Fixtures that must hold rows in which a contingency's power flow FAILS while the base case converges: name -> how many of
# the n samples; the generator counts the LoadflowNotConverged raised inside the reference's N-1 loop.
FAILED_CONTINGENCY_ROWS = {'sc_vc_hv_urban_stress': 2}
# explicit time steps to try first (else drawn from the training steps): heavy-load steps of the stand-in profiles
CANDIDATE_STEPS = {'sc_vc_hv_urban_stress': [30976, 14521, 21626]}
# solver settings of the PRODUCT environment that replays a fixture (not arguments of the reference's classes).  The fixtures
# are recorded with every power flow started flat (the stub's runpp); next to collapse WHICH contingencies converge depends
# on the start, so the replay must start the contingencies the same way
PRODUCT_KWARGS = {'sc_vc_hv_urban_stress': dict(contingency_start='flat')}

# E12 (`estimate_reward_distribution`, reward.py:181-216): fixture name -> (scenario whose environment is sampled, samples)
E12_SCENARIOS = {'e12_vc_mv_small': ('vc_mv_small', 64), 'e12_sc_hv_small': ('sc_hv_small', 24),
                 'e12_vc_noisy': ('vc_noisy', 32)}

# scenarios whose episodes take several steps: the generator records EPISODE_STEPS steps per reset
EPISODE_STEPS = {'vc_multistep_diff': 3, 'multistage_lv': 4}
# explicit start steps (else drawn from the training steps): 670 runs into the first validation week at 672
EPISODE_START_STEPS = {'multistage_lv': [5000, 20000, 670, 33000]}

# table columns snapshotted after reset (when present in the reference net)
TRACKED = [('load', 'p_mw'), ('load', 'q_mvar'), ('sgen', 'p_mw'), ('sgen', 'q_mvar'),
           ('storage', 'p_mw'), ('storage', 'q_mvar'), ('gen', 'p_mw'),
           ('sgen', 'max_p_mw'), ('sgen', 'min_p_mw'), ('sgen', 'max_q_mvar'), ('sgen', 'min_q_mvar'),
           ('storage', 'max_q_mvar'), ('storage', 'min_q_mvar'),
           ('poly_cost', 'cq2_eur_per_mvar2'), ('poly_cost', 'cp1_eur_per_mw'),
           ('pwl_cost', 'cp1_eur_per_mw'), ('load', 'max_p_mw'), ('ext_grid', 'vm_pu')]
# (switch states / tap positions are not snapshotted: they are outputs of the step, see 'tab_after__*')
