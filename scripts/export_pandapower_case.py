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
`load_tables` (no pandapower needed; opfgym_amd.definition.arrays_to_net) rebuilds the net container from such a file.
"""
import sys

import numpy as np

import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opfgym_amd.definition import arrays_to_net as load_tables, tables_to_arrays as dump_tables  # noqa: E402,F401
from opfgym_amd import capi
capi.set_default_debug(capi.debug_from_env())      # this harness is steered through OPFX_* variables (see capi.debug_from_env)

RESULTS = (('res_bus', ('vm_pu', 'va_degree')), ('res_line', ('loading_percent',)),
           ('res_trafo', ('loading_percent',)), ('res_trafo3w', ('loading_percent',)),
           ('res_ext_grid', ('p_mw', 'q_mvar')), ('res_gen', ('p_mw', 'q_mvar', 'vm_pu')))


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
