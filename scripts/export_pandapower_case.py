"""Dump a pandapower net's internal pypower case and its power-flow result.

NOT runnable in the build container (pandapower / simbench are absent there);
meant for a machine that has them:

    python scripts/export_pandapower_case.py 1-MV-urban--0-sw out.npz

Writes baseMVA, bus, branch, gen (ppci: consecutive bus numbers, in-service
elements only), br_g when the pandapower version carries a BR_G column, and the
solved voltages of `pp.runpp(net, enforce_q_lims=True)` as res_vm / res_va.
`opfgym_amd.ppci_io.load_exported_case(out.npz)` turns it into a Case; solving
that with opfx_solve and comparing against res_vm/res_va is the 1e-6 p.u. parity
check against pandapower proper that cannot be made inside this repository.
"""
import sys

import numpy as np


def main(code, out):
    import pandapower as pp
    import simbench as sb
    net = sb.get_simbench_net(code)
    pp.runpp(net, enforce_q_lims=True)
    ppci = net._ppc['internal']
    data = dict(baseMVA=np.array(ppci['baseMVA'], dtype=float), bus=ppci['bus'].real.astype(float),
                branch=ppci['branch'].real.astype(float), gen=ppci['gen'].real.astype(float),
                res_vm=ppci['bus'][:, 7].real.astype(float), res_va=ppci['bus'][:, 8].real.astype(float))
    try:
        from pandapower.pypower.idx_brch import BR_G
        data['br_g'] = ppci['branch'][:, BR_G].real.astype(float)
    except ImportError:
        data['branch_b_imag'] = ppci['branch'][:, 4].imag.astype(float)
    np.savez_compressed(out, **data)
    print(f'wrote {out}: {data["bus"].shape[0]} buses, {data["branch"].shape[0]} branches')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
