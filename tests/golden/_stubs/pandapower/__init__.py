"""Throw-away stand-in for `pandapower` used ONLY while generating golden
vectors from the reference's Python layer (tests/golden/make_golden.py): the
net container is opfgym_amd.net.Net and `runpp` is the repo's SciPy oracle."""
from opfgym_amd import net as _ppn
from opfgym_amd.net import Net as pandapowerNet  # noqa: F401
from oracle import pf_oracle as _po

from . import powerflow, optimal_powerflow, networks  # noqa: F401


def runpp(net, enforce_q_lims=False, **kwargs):
    kwargs.pop('lightsim2grid', None)
    return _po.runpp(net, enforce_q_lims=enforce_q_lims,
                     not_converged_exception=powerflow.LoadflowNotConverged, **kwargs)


def runopp(net, **kwargs):
    raise optimal_powerflow.OPFNotConverged('no OPF solver in the stub')


def diagnostic(net, **kwargs):
    return {}


def create_poly_cost(net, element, et, cp1_eur_per_mw, **kwargs):
    idx = _ppn.create_poly_cost(net, element, et, cp1_eur_per_mw, **kwargs)
    _ppn.finalize(net)
    return idx


def create_pwl_cost(net, element, et, points, power_type='p', **kwargs):
    idx = _ppn.create_pwl_cost(net, element, et, points, power_type=power_type, **kwargs)
    _ppn.finalize(net)
    return idx
