"""Dump a pandapower net — its element tables, pandapower's internal pypower case and pandapower's
own power-flow results — into one .npz fixture.

NOT runnable in the build container (pandapower / simbench are absent there); meant for a machine
that has them:

    python scripts/export_pandapower_case.py 1-MV-urban--0-sw fixtures/1-MV-urban--0-sw.npz

What the fixture pins once it exists under fixtures/ (tests/test_oracle_pf.py,
tests/test_gpu_solve.py; both skip while the directory is empty):
  * `baseMVA, bus, branch, gen[, br_g]` (ppci) + `res_vm, res_va`: the solvers against pandapower's
    voltages at 1e-6 p.u. (matrix route: oracle `pd2ppc.ppc_from_matrices`, product
    `ppci_io.case_from_ppc`);
  * `tab__<table>__<column>` (+ `idx__<table>`): the element tables themselves, so that BOTH table
    converters (oracle `pd2ppc.build_ppc`, product `case.net_to_case`) are checked against
    pandapower's `res_bus / res_line / res_trafo / res_ext_grid` (`out__*`), i.e. row P2 of SURVEY §8a.
`load_tables` (no pandapower needed) rebuilds the net container from such a file.
"""
import sys

import numpy as np

TABLES = ('bus', 'line', 'trafo', 'trafo3w', 'load', 'sgen', 'storage', 'gen', 'ext_grid', 'shunt', 'switch')
RESULTS = (('res_bus', ('vm_pu', 'va_degree')), ('res_line', ('loading_percent',)),
           ('res_trafo', ('loading_percent',)), ('res_trafo3w', ('loading_percent',)),
           ('res_ext_grid', ('p_mw', 'q_mvar')), ('res_gen', ('p_mw', 'q_mvar', 'vm_pu')))


def dump_tables(net):
    """Element tables as plain arrays: numeric columns as float64, everything else as strings."""
    out = {'scalar__sn_mva': np.array(float(net['sn_mva'])), 'scalar__f_hz': np.array(float(net['f_hz']))}
    for tbl in TABLES:
        if tbl not in net or not len(net[tbl]):
            continue
        df = net[tbl]
        out[f'idx__{tbl}'] = np.asarray(df.index, dtype=np.int64)
        for col in df.columns:
            vals = df[col].to_numpy()
            if vals.dtype == bool:
                out[f'tab__{tbl}__{col}'] = vals.astype(np.int8)
                continue
            try:
                out[f'tab__{tbl}__{col}'] = vals.astype(np.float64)
            except (TypeError, ValueError):
                out[f'str__{tbl}__{col}'] = np.array(['' if v is None or v != v else str(v) for v in vals])
    return out


def load_tables(z):
    """Inverse of dump_tables: a net container (opfgym_amd.net.Net) from the arrays of a fixture."""
    import pandas as pd
    from opfgym_amd.net import Net
    net = Net('fixture', f_hz=float(z['scalar__f_hz']), sn_mva=float(z['scalar__sn_mva']))
    cols = {}
    for key in z.files if hasattr(z, 'files') else z:
        kind, _, rest = key.partition('__')
        if kind in ('tab', 'str'):
            tbl, _, col = rest.partition('__')
            arr = np.asarray(z[key])
            if kind == 'str':
                arr = np.array([None if v == '' else str(v) for v in arr], dtype=object)
            elif arr.dtype == np.int8:
                arr = arr.astype(bool)
            cols.setdefault(tbl, {})[col] = arr
    for tbl, data in cols.items():
        net[tbl] = pd.DataFrame(data, index=np.asarray(z[f'idx__{tbl}']))
        for col in ('bus', 'from_bus', 'to_bus', 'hv_bus', 'mv_bus', 'lv_bus', 'element'):
            if col in net[tbl].columns:
                net[tbl][col] = net[tbl][col].astype(np.int64)
    return net


def main(code, out):
    import pandapower as pp
    import simbench as sb
    net = sb.get_simbench_net(code)
    pp.runpp(net, enforce_q_lims=True)
    ppci = net._ppc['internal']
    data = dict(baseMVA=np.array(ppci['baseMVA'], dtype=float), bus=ppci['bus'].real.astype(float),
                branch=ppci['branch'].real.astype(float), gen=ppci['gen'].real.astype(float),
                res_vm=ppci['bus'][:, 7].real.astype(float), res_va=ppci['bus'][:, 8].real.astype(float),
                enforce_q_lims=np.array(1))
    try:
        from pandapower.pypower.idx_brch import BR_G
        data['br_g'] = ppci['branch'][:, BR_G].real.astype(float)
    except ImportError:
        data['br_g'] = -ppci['branch'][:, 4].imag.astype(float)      # older versions: BR_B = b - j g
    data.update(dump_tables(net))
    for tbl, cols in RESULTS:
        if tbl in net and len(net[tbl]):
            for col in cols:
                if col in net[tbl].columns:
                    data[f'out__{tbl}__{col}'] = net[tbl][col].to_numpy(dtype=float)
    data['pandapower_version'] = np.array(pp.__version__)
    np.savez_compressed(out, **data)
    print(f'wrote {out}: {data["bus"].shape[0]} buses, {data["branch"].shape[0]} branches')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
