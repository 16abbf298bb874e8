"""Test helper: dress a net of `opfgym_amd.net` the way a REAL pandapower 2.13 net arrives.

pandapower (`create_empty_network` + the `create_*` functions; `>=2.13.1,<3.0`, pyproject.toml:32 of the
reference) gives every element table about twice the columns the converters need, with its own dtypes:
`uint32` bus references, `object` columns holding None / NaN / strings (`name`, `std_type`, `type`, `zone`,
`tap_side`), boolean flags, NaN `tap_pos` for transformers without a tap changer, integer-valued floats, and a
set of element tables that are present but EMPTY (xward, dcline, ... — and ward, impedance, motor where the net has none).  The column lists
below are restated from pandapower's published `create.py` / `create_empty_network` (from memory — pandapower is
not installed here), so they describe the SHAPE the converters must survive, nothing numeric.
"""
import numpy as np
import pandas as pd

# element tables pandapower creates empty; columns with their dtypes
EMPTY_TABLES = {
    'ward': [('name', object), ('bus', 'u4'), ('ps_mw', 'f8'), ('qs_mvar', 'f8'), ('qz_mvar', 'f8'), ('pz_mw', 'f8'),
             ('in_service', bool)],
    'xward': [('name', object), ('bus', 'u4'), ('ps_mw', 'f8'), ('qs_mvar', 'f8'), ('qz_mvar', 'f8'), ('pz_mw', 'f8'),
              ('r_ohm', 'f8'), ('x_ohm', 'f8'), ('vm_pu', 'f8'), ('slack_weight', 'f8'), ('in_service', bool)],
    'impedance': [('name', object), ('from_bus', 'u4'), ('to_bus', 'u4'), ('rft_pu', 'f8'), ('xft_pu', 'f8'),
                  ('rtf_pu', 'f8'), ('xtf_pu', 'f8'), ('sn_mva', 'f8'), ('in_service', bool)],
    'dcline': [('name', object), ('from_bus', 'u4'), ('to_bus', 'u4'), ('p_mw', 'f8'), ('loss_percent', 'f8'),
               ('loss_mw', 'f8'), ('vm_from_pu', 'f8'), ('vm_to_pu', 'f8'), ('max_p_mw', 'f8'), ('min_q_from_mvar', 'f8'),
               ('min_q_to_mvar', 'f8'), ('max_q_from_mvar', 'f8'), ('max_q_to_mvar', 'f8'), ('in_service', bool)],
    'motor': [('name', object), ('bus', 'i8'), ('pn_mech_mw', 'f8'), ('loading_percent', 'f8'), ('cos_phi', 'f8'),
              ('cos_phi_n', 'f8'), ('efficiency_percent', 'f8'), ('efficiency_n_percent', 'f8'), ('lrc_pu', 'f8'),
              ('vn_kv', 'f8'), ('scaling', 'f8'), ('in_service', bool), ('rx', 'f8')],
    'asymmetric_load': [('name', object), ('bus', 'u4'), ('p_a_mw', 'f8'), ('q_a_mvar', 'f8'), ('p_b_mw', 'f8'),
                        ('q_b_mvar', 'f8'), ('p_c_mw', 'f8'), ('q_c_mvar', 'f8'), ('sn_mva', 'f8'), ('scaling', 'f8'),
                        ('in_service', bool), ('type', object)],
    'asymmetric_sgen': [('name', object), ('bus', 'i8'), ('p_a_mw', 'f8'), ('q_a_mvar', 'f8'), ('p_b_mw', 'f8'),
                        ('q_b_mvar', 'f8'), ('p_c_mw', 'f8'), ('q_c_mvar', 'f8'), ('sn_mva', 'f8'), ('scaling', 'f8'),
                        ('in_service', bool), ('type', object), ('current_source', bool)],
    'svc': [('name', object), ('bus', 'u4'), ('x_l_ohm', 'f8'), ('x_cvar_ohm', 'f8'), ('set_vm_pu', 'f8'),
            ('thyristor_firing_angle_degree', 'f8'), ('controllable', bool), ('in_service', bool),
            ('min_angle_degree', 'f8'), ('max_angle_degree', 'f8')],
    'tcsc': [('name', object), ('from_bus', 'u4'), ('to_bus', 'u4'), ('x_l_ohm', 'f8'), ('x_cvar_ohm', 'f8'),
             ('set_p_to_mw', 'f8'), ('thyristor_firing_angle_degree', 'f8'), ('controllable', bool), ('in_service', bool)],
    'measurement': [('name', object), ('measurement_type', object), ('element_type', object), ('element', 'u4'),
                    ('value', 'f8'), ('std_dev', 'f8'), ('side', object)],
    'controller': [('object', object), ('in_service', bool), ('order', 'f8'), ('level', object),
                   ('initial_run', bool), ('recycle', object)],
    'characteristic': [('object', object)],
    'group': [('name', object), ('element_type', object), ('element', object), ('reference_column', object)],
}

# columns pandapower adds to the tables the converters read: (column, dtype, value or callable(n, rng))
EXTRA_COLUMNS = {
    'bus': [('zone', object, None), ('geo', object, None)],
    'line': [('std_type', object, lambda n, rng: rng.choice(['NAYY 4x150 SE', '243-AL1/39-ST1A 110.0', None], n)),
             ('type', object, lambda n, rng: rng.choice(['cs', 'ol'], n)), ('endtemp_degree', 'f8', np.nan),
             ('alpha', 'f8', np.nan), ('temperature_degree_celsius', 'f8', np.nan), ('geo', object, None)],
    'trafo': [('std_type', object, lambda n, rng: rng.choice(['63 MVA 110/20 kV', '0.4 MVA 20/0.4 kV', None], n)),
              ('tap_min', 'f8', lambda n, rng: np.full(n, -9.0)), ('tap_max', 'f8', lambda n, rng: np.full(n, 9.0)),
              ('oltc', bool, False), ('tap_dependent_impedance', object, None), ('vk0_percent', 'f8', np.nan),
              ('vkr0_percent', 'f8', np.nan), ('mag0_percent', 'f8', np.nan), ('si0_hv_partial', 'f8', np.nan),
              ('vector_group', object, None), ('power_station_unit', object, None)],
    'trafo3w': [('std_type', object, None), ('tap_min', 'f8', -10.0), ('tap_max', 'f8', 10.0),
                ('tap_step_degree', 'f8', np.nan), ('tap_at_star_point', bool, False), ('vector_group', object, None)],
    'load': [('const_z_percent', 'f8', 0.0), ('const_i_percent', 'f8', 0.0), ('sn_mva', 'f8', np.nan),
             ('type', object, lambda n, rng: rng.choice(['wye', 'delta', None], n)), ('zone', object, None)],
    'sgen': [('sn_mva', 'f8', np.nan), ('type', object, lambda n, rng: rng.choice(['PV', 'WP', 'wye', None], n)),
             ('current_source', bool, True), ('k', 'f8', np.nan), ('rx', 'f8', np.nan), ('generator_type', object, None),
             ('lrc_pu', 'f8', np.nan), ('max_ik_ka', 'f8', np.nan), ('kappa', 'f8', np.nan)],
    'storage': [('sn_mva', 'f8', np.nan), ('soc_percent', 'f8', 50.0), ('min_e_mwh', 'f8', 0.0),
                ('max_e_mwh', 'f8', np.inf), ('type', object, None)],
    'gen': [('sn_mva', 'f8', np.nan), ('slack', bool, False), ('slack_weight', 'f8', 0.0), ('type', object, None),
            ('vn_kv', 'f8', np.nan), ('xdss_pu', 'f8', np.nan), ('rdss_ohm', 'f8', np.nan), ('cos_phi', 'f8', np.nan),
            ('pg_percent', 'f8', np.nan), ('power_station_trafo', 'f8', np.nan)],
    'ext_grid': [('slack_weight', 'f8', 1.0), ('s_sc_max_mva', 'f8', np.nan), ('s_sc_min_mva', 'f8', np.nan),
                 ('rx_min', 'f8', np.nan), ('rx_max', 'f8', np.nan), ('r0x0_max', 'f8', np.nan), ('x0x_max', 'f8', np.nan)],
    'shunt': [('name', object, None), ('max_step', 'u4', lambda n, rng: np.full(n, 3, dtype=np.uint32))],
    'switch': [('type', object, lambda n, rng: rng.choice(['CB', 'LBS', 'DS', None], n)), ('name', object, None),
               ('z_ohm', 'f8', 0.0), ('in_ka', 'f8', np.nan)],
}

UINT_COLUMNS = {'load': ['bus'], 'storage': ['bus'], 'gen': ['bus'], 'ext_grid': ['bus'], 'shunt': ['bus', 'step'],
                'line': ['from_bus', 'to_bus', 'parallel'], 'trafo': ['hv_bus', 'lv_bus', 'parallel'],
                'trafo3w': ['hv_bus', 'mv_bus', 'lv_bus']}


def dress_as_pandapower(net, seed=0):
    """In place: the full pandapower 2.13 column set and dtypes on every table of `net`; returns `net`."""
    rng = np.random.default_rng(seed)
    for tbl, cols in EXTRA_COLUMNS.items():
        if tbl not in net:
            continue
        df = net[tbl]
        n = len(df)
        for col, dtype, value in cols:
            if col in df.columns:
                continue
            vals = value(n, rng) if callable(value) else np.full(n, value, dtype=object if dtype is object else None)
            df[col] = pd.Series(list(vals), index=df.index, dtype=dtype if dtype is not object else object)
    # pandapower's own dtypes: unsigned bus references, NaN tap data for transformers without a tap changer,
    # None where a string is not set, tap positions as integer-valued floats next to NaN
    for tbl, cols in UINT_COLUMNS.items():
        if tbl in net and len(net[tbl]):
            for col in cols:
                if col in net[tbl].columns:
                    net[tbl][col] = net[tbl][col].astype(np.uint32)
    tr = net['trafo']
    if len(tr):
        no_tap = np.array([not isinstance(s, str) for s in tr['tap_side']])
        for col in ('tap_pos', 'tap_neutral', 'tap_min', 'tap_max', 'tap_step_percent'):
            tr[col] = tr[col].astype(float)
            tr.loc[tr.index[no_tap], col] = np.nan
        tr['tap_side'] = pd.Series([s if isinstance(s, str) else None for s in tr['tap_side']], index=tr.index, dtype=object)
        if 'tap_step_degree' not in tr.columns:
            tr['tap_step_degree'] = np.nan
        if 'tap_phase_shifter' not in tr.columns:
            tr['tap_phase_shifter'] = False
        tr['tap_phase_shifter'] = pd.Series([bool(v) if v is not None and v == v else False for v in tr['tap_phase_shifter']],
                                            index=tr.index, dtype=bool)
    for tbl in ('bus', 'line', 'trafo', 'load', 'sgen', 'gen', 'storage', 'ext_grid'):
        if tbl in net and len(net[tbl]):
            net[tbl]['name'] = pd.Series([f'{tbl} {i}' if i % 3 else None for i in range(len(net[tbl]))],
                                         index=net[tbl].index, dtype=object)
    for tbl, cols in EMPTY_TABLES.items():
        if tbl in net and len(net[tbl]):              # (ward / impedance / motor rows of the net itself: pandapower's columns and dtypes on them)
            df = net[tbl]
            for c, d in cols:
                if c not in df.columns:
                    df[c] = pd.Series([None if d is object else np.nan] * len(df), index=df.index, dtype=object if d is object else 'f8')
                elif d in ('u4', 'i8'):
                    df[c] = df[c].astype(np.uint32 if d == 'u4' else np.int64)
            continue
        net[tbl] = pd.DataFrame({c: pd.Series(dtype=(object if d is object else d)) for c, d in cols})
    net['std_types'] = {'line': {}, 'trafo': {}, 'trafo3w': {}}
    net['version'] = '2.13.1'
    net['format_version'] = '2.12.0'
    net['converged'] = False
    net['user_pf_options'] = {}
    return net


def unmodelled_variants():
    """(label, function(net) that adds content `pp.runpp` would model and the converters do not)."""
    def add_row(tbl, **vals):
        def f(net):
            df = net[tbl]
            row = {c: vals.get(c, (None if df[c].dtype == object else (False if df[c].dtype == bool else 0))) for c in df.columns}
            net[tbl] = pd.concat([df, pd.DataFrame([row])], ignore_index=True)
        return f

    def set_col(tbl, col, value):
        def f(net):
            net[tbl].loc[net[tbl].index[0], col] = value
        return f
    b = 1
    return [
        ('asymmetric_load', add_row('asymmetric_load', bus=b, p_a_mw=0.01, in_service=True)),
        ('svc', add_row('svc', bus=b, x_l_ohm=1.0, x_cvar_ohm=-10.0, set_vm_pu=1.0, in_service=True)),
        ('load.const_z_percent', set_col('load', 'const_z_percent', 30.0)),
        ('load.const_i_percent', set_col('load', 'const_i_percent', 10.0)),
        ('switch.z_ohm', lambda net: net['switch'].__setitem__('z_ohm', np.where(net['switch']['et'] == 'l', 0.05, 0.0))),   # (at LINE switches; bus-bus: modelled)
        ('switch.et', set_col('switch', 'et', 't3')),
        ('gen.slack', set_col('gen', 'slack', True)),
        ('trafo.tap_dependent_impedance', set_col('trafo', 'tap_dependent_impedance', True)),
    ]
