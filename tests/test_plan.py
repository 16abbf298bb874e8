"""CPU: the C-ABI library loads, exports every symbol declared in
include/opfx.h, and the host-side symbolic plan (ordering, block pattern,
level schedule) is correct: walking the compiled programme in numpy
(tests/plan_emulator.py) reproduces the SciPy oracle's Newton iterates."""
import re
import os

import numpy as np
import pytest

from opfgym_amd import capi, grids
from opfgym_amd.case import bus_injections, net_to_case
from helpers import OracleSide
from plan_emulator import emulate_newton, emulate_newton_lane_program, load_plan

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(built_lib):
    header = open(os.path.join(ROOT, 'include', 'opfx.h')).read()
    declared = set(re.findall(r'\b(opfx_[a-z_]+)\s*\(', header))
    declared -= {'opfx_case', 'opfx_plan', 'opfx_ctx', 'opfx_env'}
    assert declared == set(capi.EXPORTS)
    for name in declared:
        assert hasattr(built_lib, name), name
    ver = [capi.C.c_int() for _ in range(3)]
    built_lib.opfx_version(*[capi.C.byref(v) for v in ver])
    assert (ver[0].value, ver[1].value) == capi.ABI_VERSION == (0, 3)
    # the header and the library agree on the version; the developer entry points live in a header of their own
    assert re.search(r'#define OPFX_VERSION_MAJOR (\d+)', header).group(1) == str(ver[0].value)
    assert re.search(r'#define OPFX_VERSION_MINOR (\d+)', header).group(1) == str(ver[1].value)
    dbg = open(os.path.join(ROOT, 'include', 'opfx_debug.h')).read()
    declared_dbg = set(re.findall(r'^int (opfx_[a-z_]+)\s*\(', dbg, re.M))
    assert declared_dbg == set(capi.DEBUG_EXPORTS)
    for name in declared_dbg:
        assert hasattr(built_lib, name), name
    # nothing else is exported, and the library reads no environment variable (opfx.h: "no hidden global state")
    import subprocess
    nm = subprocess.run(['nm', '-D', '--defined-only', capi.LIB_PATH], capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in nm.splitlines() if ' T ' in ln and ln.split()[-1].startswith('opfx_')}
    assert exported == declared | declared_dbg, exported ^ (declared | declared_dbg)
    undefined = subprocess.run(['nm', '-D', '--undefined-only', capi.LIB_PATH], capture_output=True, text=True).stdout
    assert 'getenv' not in undefined
    for src in ('opfx.hip', 'plan.cpp'):
        assert 'getenv' not in open(os.path.join(ROOT, 'opfgym_amd', 'csrc', src)).read(), src


def test_structs_carry_their_size_and_a_wrong_size_is_refused(built_lib):
    """include/opfx.h, VERSIONING: every struct that crosses the boundary starts with `struct_size`; a binding built
    against another header (a shorter or longer struct) gets OPFX_ERR_INVALID with text instead of a library that reads
    past the end of the caller's struct (`init` was inserted into the middle of opfx_solve_opts)."""
    C = capi.C
    header = open(os.path.join(ROOT, 'include', 'opfx.h')).read()
    for name in ('opfx_case', 'opfx_plan_info', 'opfx_solve_opts', 'opfx_env_desc', 'opfx_step_io', 'opfx_profile_desc',
                 'opfx_reset_desc', 'opfx_reset_io'):
        body = re.search(r'typedef struct %s \{(.*?)\} %s;' % (name, name), header, re.S).group(1)
        first = [ln.strip() for ln in body.splitlines() if ln.strip() and not ln.strip().startswith(('/*', '*'))][0]
        assert first.startswith('uint32_t struct_size;'), (name, first)
    for cls in (capi.CaseStruct, capi.PlanInfo, capi.SolveOpts, capi.EnvDesc, capi.StepIO, capi.ProfileDesc, capi.ResetDesc,
                capi.ResetIO, capi.DebugOpts):
        assert cls._fields_[0][0] == 'struct_size' and cls().struct_size == C.sizeof(cls)
    # the ctypes layouts equal the C compiler's: sizeof of every struct through a tiny C program
    import subprocess, tempfile
    names = ['opfx_case', 'opfx_plan_info', 'opfx_solve_opts', 'opfx_env_desc', 'opfx_step_io', 'opfx_profile_desc',
             'opfx_reset_desc', 'opfx_reset_io', 'opfx_debug_opts']
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, 's.c')
        open(src, 'w').write('#include <stdio.h>\n#include "opfx_debug.h"\nint main(void) {' +
                             ''.join(f'printf("%zu\\n", sizeof({n}));' for n in names) + 'return 0; }\n')
        subprocess.run(['gcc', '-std=c99', '-pedantic', '-Wall', '-Werror', '-I', os.path.join(ROOT, 'include'), src, '-o',
                        os.path.join(td, 's')], check=True)
        sizes = [int(x) for x in subprocess.run([os.path.join(td, 's')], capture_output=True, text=True).stdout.split()]
    assert sizes == [C.sizeof(c) for c in (capi.CaseStruct, capi.PlanInfo, capi.SolveOpts, capi.EnvDesc, capi.StepIO,
                                          capi.ProfileDesc, capi.ResetDesc, capi.ResetIO, capi.DebugOpts)]
    # a case with a wrong size: refused, with the expected size in the text
    case = net_to_case(grids.two_bus())
    plan = capi.Plan(case)                                   # (the right size works)
    cs = capi.CaseStruct()
    cs.nb, cs.nbr = 2, 1
    h = C.c_void_p()
    for wrong in (0, C.sizeof(capi.CaseStruct) - 16, C.sizeof(capi.CaseStruct) + 8):
        cs.struct_size = wrong
        assert built_lib.opfx_plan_create(C.byref(cs), C.byref(h)) == -1
        msg = built_lib.opfx_last_error().decode()
        assert 'struct_size' in msg and str(C.sizeof(capi.CaseStruct)) in msg, msg
    info = capi.PlanInfo()
    info.struct_size = C.sizeof(capi.PlanInfo) + 4
    assert built_lib.opfx_plan_get_info(plan.handle, C.byref(info)) == -1
    # an out-struct of an older, shorter layout gets its prefix only (members are appended, never inserted)
    info = capi.PlanInfo()
    info.struct_size = 6 * 4
    info.nnz_y = -7
    assert built_lib.opfx_plan_get_info(plan.handle, C.byref(info)) == 0
    assert (info.nb, info.nbr, info.nref, info.npq) == (2, 1, 1, 1) and info.nnz_y == -7
    dbg = capi.DebugOpts()
    dbg.struct_size = 8
    assert built_lib.opfx_plan_create_debug(C.byref(capi.Plan(case).case_struct), C.byref(dbg), C.byref(h)) == -1
    assert 'opfx_debug_opts' in built_lib.opfx_last_error().decode()


def test_developer_switches_travel_in_an_explicit_struct(built_lib, monkeypatch):
    """include/opfx_debug.h: what used to be OPFX_* environment variables of the LIBRARY is a struct handed to the *_debug
    constructors; the binding takes it explicitly (`debug=`: a struct or a dict of member names) and never reads the
    process environment for it — `debug_from_env()` is a helper for scripts."""
    case = net_to_case(grids.get_grid('1-HV-mixed--0-sw')[0])
    default = capi.Plan(case).info
    d = capi.DebugOpts()
    d.plan_search = -1
    first = capi.Plan(case, debug=d).info
    # the process environment steers NOTHING by itself: a Plan built without `debug=` is the default plan whatever OPFX_*
    # says; `debug_from_env()` is the explicit helper of scripts and harnesses that turns such variables into the struct
    monkeypatch.setenv('OPFX_PLAN_SEARCH', '0')
    assert capi.Plan(case).info == default
    assert capi.debug_from_env().plan_search == -1 and capi.Plan(case, debug=capi.debug_from_env()).info == first
    assert capi.Plan(case, debug=dict(plan_search=-1)).info == first
    monkeypatch.delenv('OPFX_PLAN_SEARCH')
    d = capi.DebugOpts()
    d.plan_no_tail = 1
    assert capi.Plan(case, debug=d).info['tail_m'] == 0 and default['tail_m'] > 0
    assert not capi.debug_from_env({}).any()
    e = capi.debug_from_env({'OPFX_TEAM': '2', 'OPFX_QUEUE': '0', 'OPFX_FORCE_MEM': '1', 'OPFX_PACKED': '1'})
    assert (e.team, e.queue, e.force_mem, e.packed) == (2, -1, 1, 1)


def test_bad_arguments_return_status_not_exceptions(built_lib):
    h = capi.C.c_void_p()
    assert built_lib.opfx_plan_create(None, capi.C.byref(h)) == -1
    assert b'null' in built_lib.opfx_last_error()
    net = grids.two_bus()
    case = net_to_case(net)
    case.bus_type = np.array([1, 1], dtype=np.int32)           # no slack
    with pytest.raises(capi.OpfxError, match='no REF'):
        capi.Plan(case)


@pytest.mark.parametrize('code', ['case9', '1-LV-rural1--0-sw', 'mv-small', '1-MV-urban--0-sw', 'hv-small'])
def test_schedule_reproduces_oracle(built_lib, code):
    net = grids.case9() if code == 'case9' else grids.get_grid(code)[0]
    case = net_to_case(net)
    plan = capi.Plan(case)
    p, q, *_ = bus_injections(net, case)
    p, q = p / case.base_mva, q / case.base_mva
    ref = OracleSide(net, case).solve(p, q)              # the oracle's own case of the same net
    v, conv, it, nrm = emulate_newton(plan, p, q)
    assert conv and ref['converged'] and it == ref['iterations']
    assert np.abs(v - ref['V']).max() < 1e-11


@pytest.mark.parametrize('code', ['case9', '1-LV-rural1--0-sw', '1-MV-urban--0-sw', 'hv-small'])
def test_lane_programme_reproduces_oracle(built_lib, code):
    """The register/stream form of the schedule that kernel `newton2` executes
    (ELL rows + overflow entries, flat update items, solve items with inline
    U-terms, relative-|V| unknowns) converges to the oracle's solution in the
    same number of iterations."""
    net = grids.case9() if code == 'case9' else grids.get_grid(code)[0]
    case = net_to_case(net)
    plan = capi.Plan(case)
    p, q, *_ = bus_injections(net, case)
    p, q = p / case.base_mva, q / case.base_mva
    ref = OracleSide(net, case).solve(p, q)
    v, conv, it, nrm = emulate_newton_lane_program(plan, p, q)
    assert conv and it == ref['iterations']
    assert np.abs(v - ref['V']).max() < 1e-9          # both stop at ||F|| < 1e-8


@pytest.mark.parametrize('code,team', [('hv-small', 2), ('hv-small', 4), ('1-HV-mixed--0-sw', 4), ('1-HV-mixed--0-sw', 2),
                                       ('1-HV-urban--0-sw', 4), ('1-MV-urban--0-sw', 2)])
def test_team_stream_reproduces_oracle(built_lib, code, team):
    """The wave-team form of the factor/solve stream (kernel `newton2_coop`): rounds dealt to 2 or 4 wavefronts,
    barriers only where the plan says so (none between consecutive one-round groups), the dense tail's back
    substitution as a register chain of wavefront 0 between the two parts of the stream.  The emulation walks
    exactly that and checks, between any two barriers, that no wavefront reads a location another one adds to —
    the condition under which every interleaving the hardware may choose gives the same sums."""
    net = grids.get_grid(code)[0]
    case = net_to_case(net)
    plan = capi.Plan(case)
    p, q, *_ = bus_injections(net, case)
    p, q = p / case.base_mva, q / case.base_mva
    ref = OracleSide(net, case).solve(p, q)
    v, conv, it, nrm = emulate_newton_lane_program(plan, p, q, team=team)
    assert conv and it == ref['iterations']
    assert np.abs(v - ref['V']).max() < 1e-9


@pytest.mark.parametrize('code,team', [('hv-small', 2), ('hv-small', 4), ('1-HV-mixed--0-sw', 2), ('1-HV-mixed--0-sw', 4),
                                       ('1-HV-urban--0-sw', 4)])
def test_shared_slots_give_the_same_power_flow_with_less_lds(built_lib, code, team):
    """opfx_debug_opts.plan_share_slots (round 5, plan.cpp share_slots): fill blocks born late live in the ids (= LDS slots) of
    lower blocks that are dead by then; a zero store riding on a team item of an earlier level prepares the slot.  The
    emulation walks the team stream with those stores and its race check covers them (no wavefront may use, between two
    barriers, a slot another one zeroes); the raw schedule is walked as well.  Same iterates, fewer blocks."""
    from plan_emulator import emulate_newton
    net = grids.get_grid(code)[0]
    case = net_to_case(net)
    plain = capi.Plan(case)
    shared = capi.Plan(case, debug=dict(plan_share_slots=1))
    ns = shared.info['n_shared']
    assert plain.info['n_shared'] == 0 and ns > 0
    if case.nb < 200:            # (from 200 buses on the plan is SEARCHED, and with less LDS another elimination order may win)
        assert shared.info['n_blk'] == plain.info['n_blk'] - ns and shared.info['n_full'] == plain.info['n_full'] - ns
        assert abs(plain.info['lds_doubles'] - shared.info['lds_doubles'] - 4 * ns) <= 4      # (counts are rounded up to even)
        assert len(shared.array('fill_blk')) == plain.info['n_fill'] - ns
    assert shared.info['lds_doubles'] < plain.info['lds_doubles']
    zl, zb = shared.array('zero_lev'), shared.array('zero_blk')
    assert len(zl) == len(zb) == ns and (zb < shared.info['n_full']).all() and (zl >= 0).all() and (zl < shared.info['n_levels']).all()
    fill = shared.array('fill_blk')                               # (ids whose FIRST tenant is a fill block: one range, zeroed in phase A)
    assert (np.diff(fill) == 1).all() and fill[-1] == shared.info['n_full'] - 1
    p, q, *_ = bus_injections(net, case)
    p, q = p / case.base_mva, q / case.base_mva
    ref = OracleSide(net, case).solve(p, q)
    v, conv, it, nrm = emulate_newton_lane_program(shared, p, q, team=team)
    assert conv and it == ref['iterations']
    assert np.abs(v - ref['V']).max() < 1e-9
    v0, conv0, it0, _ = emulate_newton_lane_program(plain, p, q, team=team)
    assert it0 == it and np.abs(v - v0).max() < 1e-10
    if code != '1-HV-urban--0-sw':                                # (the raw schedule of the largest grid takes a minute in numpy)
        vr, convr, itr, _ = emulate_newton(shared, p, q)
        assert convr and itr == it and np.abs(vr - ref['V']).max() < 1e-9


@pytest.mark.parametrize('code,team', [('1-MV-urban--0-sw', 0), ('hv-small', 2), ('1-HV-mixed--0-sw', 4)])
def test_second_columns_carry_every_update_term_once(built_lib, code, team):
    """Round 4: a factor item may carry a SECOND column of the same multiplier -A_ik A_kk^-1 (device word 2).  Each stream
    takes a level with second columns only where that saves a round per wavefront; plan_no_pairs = 1 / 2 switches them
    off / on everywhere.  All three plans hold every update term of the schedule exactly once and give the oracle's
    iterates."""
    net = grids.get_grid(code)[0]
    case = net_to_case(net)
    p, q, *_ = bus_injections(net, case)
    p, q = p / case.base_mva, q / case.base_mva
    ref = OracleSide(net, case).solve(p, q)
    counts = {}
    for mode in (0, 1, 2):
        d = capi.DebugOpts()
        d.plan_no_pairs = mode
        d.plan_search = -1              # (one elimination order for the three modes: the search would weigh their rounds)
        plan = capi.Plan(case, debug=d)
        if team:
            it4 = plan.array(f'LP_TEAM{team}').astype(np.int64).reshape(-1, 4) & 0xFFFFFFFF
            first, second = it4[:, 0], it4[:, 2]            # (back-substitution items have right-hand-side targets)
        else:
            first = plan.array('LP_B').astype(np.int64).reshape(-1, 2)[:, 0] & 0xFFFFFFFF
            second = plan.array('LP_B3').astype(np.int64) & 0xFFFFFFFF
        live = (first & 0xFFFF) != 0xFFFF
        blk = live & ((first & 0x8000) == 0)
        two = live & ((second & 0xFFFF) != 0xFFFF)
        assert not (two & ~blk).any()                          # only a block target has a second column
        counts[mode] = (int(blk.sum() + two.sum()), int(two.sum()))
        v, conv, it, _ = emulate_newton_lane_program(plan, p, q, team=team)
        assert conv and it == ref['iterations'] and np.abs(v - ref['V']).max() < 1e-9
    first_order = load_plan(capi.Plan(case, debug=dict(plan_search=-1)))
    n_block_terms = int((np.repeat(first_order['tgt_blk'], np.diff(first_order['tgt_sptr'])) >= 0).sum())
    assert counts[0][0] == counts[1][0] == counts[2][0] == n_block_terms
    assert counts[1][1] == 0 and counts[0][1] <= counts[2][1] and counts[2][1] > 0      # (a small grid on a team: none by default)


@pytest.mark.parametrize('code,team,theta', [('1-MV-urban--0-sw', 0, 1e-2), ('hv-small', 0, 1e-1), ('hv-small', 2, 1e-1), ('1-HV-mixed--0-sw', 4, 1e-1),
                                             ('1-HV-mixed--0-sw', 2, 1.0), ('1-HV-urban--0-sw', 4, 1.0), ('case9', 0, 1.0)])
def test_chord_stream_reaches_the_same_fixed_point(built_lib, code, team, theta):
    """Chord steps (opfx_solve_opts.jacobian_reuse_tol = theta): once an iteration's mismatch is below theta the later
    ones keep its factorisation and walk the CHORD stream — the forward substitution alone (right-hand-side items of
    every level, reading A_ik and A_kk as the factorisation left them) and the same back substitution.  Emulated as the
    CHORD kernels run it, single wavefront and wave teams (with the race check between barriers): same fixed point
    as full Newton, at least one iteration without a factorisation, never fewer iterations than full Newton."""
    net = grids.case9() if code == 'case9' else grids.get_grid(code)[0]
    case = net_to_case(net)
    plan = capi.Plan(case)
    p, q, *_ = bus_injections(net, case)
    p, q = p / case.base_mva, q / case.base_mva
    ref = OracleSide(net, case).solve(p, q)
    trace = []
    v, conv, it, nrm = emulate_newton_lane_program(plan, p, q, team=team, reuse_tol=theta, trace=trace)
    assert conv and nrm < 1e-8 and it >= ref['iterations'] and it <= ref['iterations'] + 3
    assert np.abs(v - ref['V']).max() < 1e-8
    kinds = [f for _, f in trace]
    assert kinds[0] and not all(kinds), trace                      # the first iteration factorises; a later one does not
    info = plan.info
    assert info['lp_rounds_f'] <= info['lp_rounds_b'] and info['lp_rounds_f_pad'] % 4 == 0
    if team:
        assert info[f'team_rounds_chord_{team}'] <= info[f'team_rounds_{team}']
        assert info[f'team_barriers_chord_{team}'] <= info[f'team_barriers_{team}']
    # theta = 0 is full Newton: the oracle's iteration count exactly
    v0, conv0, it0, _ = emulate_newton_lane_program(plan, p, q, team=team, reuse_tol=0.0)
    assert conv0 and it0 == ref['iterations']


@pytest.mark.parametrize('n,team', [(27, 4), (30, 2), (40, 4)])
def test_team_stream_on_a_complete_graph(built_lib, n, team):
    """Tails longer than the fuzz grids produce: a complete graph eliminates one pivot per level from the start, so
    the register chain runs at its largest instantiation (25..32 pivots) and, beyond 32 pivots, shares the
    back substitution with ordinary single-round groups."""
    from helpers import dense_ppc, oracle_ppc_solve
    from opfgym_amd.ppci_io import case_from_ppc
    base, bus, branch, gen = dense_ppc(n)
    case, p, q, _, _ = case_from_ppc(base, bus, branch, gen)
    plan = capi.Plan(case)
    assert plan.info['tail_m'] == min(n - 1, 32)                        # (the slack bus is no pivot)
    ref = oracle_ppc_solve(base, bus, branch, gen)
    v, conv, it, nrm = emulate_newton_lane_program(plan, p, q, team=team)
    assert conv and ref['converged'] and it == ref['iterations']
    assert np.abs(v - ref['V']).max() < 1e-9


def test_dense_tail_tables(built_lib):
    """The tail the register chain works on: the final levels with one pivot each; its table names, for every
    pair of tail pivots e < s, the U-block (row e, column s) and — lower triangle, the forward direction of the chain in
    chord iterations — the L-block (row s, column e); all of them four-value blocks."""
    net = grids.get_grid('1-HV-mixed--0-sw')[0]
    plan = capi.Plan(net_to_case(net))
    P, info = load_plan(plan), plan.info
    m = info['tail_m']
    assert 8 <= m <= 32 and info['team_kb_4'] < info['team_rounds_4']
    per_level = np.diff(P['lev_pptr'])
    assert (per_level[-m:] == 1).all() and per_level[-m - 1] > 1
    M = (m + 7) & ~7
    tb = plan.array('tail_bus').view(np.uint32)
    ids = plan.array('tail_ids').reshape(m + 1, M)
    bus = (tb[:m] & 0xFFFF).astype(int)
    assert bus.tolist() == P['piv_bus'][-m:].tolist()
    assert ((tb[:m] >> 16).astype(int) == P['diag_blk'][bus]).all()
    where = {(int(r), int(c)): b for b, (r, c) in enumerate(zip(P['blk_row'], P['blk_col']))}
    n_in = 0
    for e in range(m):
        for s in range(M):
            if e != s and e < m and s < m and (bus[e], bus[s]) in where:
                assert ids[e, s] == where[(bus[e], bus[s])] and ids[e, s] < info['n_full']
                n_in += 1
            else:                             # (also a pair of tail pivots without a block: the tail need not fill in completely)
                assert ids[e, s] == 0xFFFF
    assert n_in >= 0.9 * m * (m - 1)                # (the last separator fills in almost completely)
    # without a tail (radial grid) the stream is one part
    plan2 = capi.Plan(net_to_case(grids.get_grid('1-MV-urban--0-sw')[0]))
    assert plan2.info['tail_m'] == 0 and plan2.info['team_kb_2'] == plan2.info['team_rounds_2']


def test_plan_search_keeps_the_cheapest_elimination_order(built_lib, monkeypatch):
    """For grids of the wave-team kernels `opfx_plan_create` tries several tie-breaking rules of the minimum-degree
    ordering and keeps the plan with the fewest rounds per wavefront (plan.cpp: plan_cost); below 200 buses there is
    no search.  Same grid, same plan every time."""
    case = net_to_case(grids.get_grid('1-HV-mixed--0-sw')[0])
    cost = lambda i: i['team_rounds_4'] + 0.25 * i['team_barriers_4']
    searched = capi.Plan(case).info
    again = capi.Plan(case).info
    assert searched == again
    first = capi.Plan(case, debug=dict(plan_search=-1)).info
    assert cost(searched) <= cost(first)
    few = capi.Plan(case, debug=dict(plan_search=3)).info
    assert cost(searched) <= cost(few) <= cost(first)
    small = net_to_case(grids.get_grid('1-MV-urban--0-sw')[0])
    assert capi.Plan(small, debug=dict(plan_search=-1)).info == capi.Plan(small).info


def test_plan_structure_invariants(built_lib):
    net, _ = grids.get_grid('1-HV-mixed--0-sw')
    case = net_to_case(net)
    plan = capi.Plan(case)
    P = load_plan(plan)
    info = plan.info
    nb = info['nb']
    # every non-REF bus is a pivot exactly once; REF buses never
    piv = P['piv_bus']
    assert sorted(piv.tolist()) == sorted(np.flatnonzero(case.bus_type != 3).tolist())
    # pivots of one level are pairwise non-adjacent in the filled graph of that level
    blocks = set(zip(P['blk_row'].tolist(), P['blk_col'].tolist()))
    assert len(blocks) == info['n_blk']                      # no duplicate block
    for lev in range(info['n_levels']):
        ps = piv[P['lev_pptr'][lev]:P['lev_pptr'][lev + 1]]
        for a in ps:
            for u in range(P['piv_uptr'][np.flatnonzero(piv == a)[0]], P['piv_uptr'][np.flatnonzero(piv == a)[0] + 1]):
                assert P['u_col'][u] not in ps
    # symmetric pattern, diagonal present
    for (i, j) in blocks:
        assert (j, i) in blocks
    assert all(P['diag_blk'][i] >= 0 for i in range(nb) if case.bus_type[i] != 3)
    # Ybus of the plan equals the dense assembly of the case
    g, b = plan.ybus()
    y = np.zeros((nb, nb), complex)
    for i in range(nb):
        for e in range(P['y_ptr'][i], P['y_ptr'][i + 1]):
            y[i, P['y_col'][e]] = g[e] + 1j * b[e]
    assert np.abs(y - case.ybus_dense()).max() < 1e-12 * np.abs(y).max()   # summation order only
    ev = lambda v: (v + 1) & ~1
    assert info['lds_doubles'] == 4 * ev(nb) + 2 * ev(info['n_blk']) + 2 * ev(info['n_full'])
    assert info['n_fill'] <= info['n_full'] <= info['n_blk']
    # blocks beyond n_full are off-diagonal Jacobian blocks of PQ rows that no update targets
    tgt = set(int(b) for b in P['tgt_blk'] if b >= 0)
    for b in range(info['n_full'], info['n_blk']):
        i, j = int(P['blk_row'][b]), int(P['blk_col'][b])
        assert i != j and case.bus_type[i] == 1 and b not in tgt and b not in set(P['fill_blk'].tolist())


def test_ctx_create_fails_loudly_without_gpu(built_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    plan = capi.Plan(net_to_case(grids.two_bus()))
    with pytest.raises(capi.OpfxError):
        capi.Context(plan, 0)


def test_islanding_outages_flagged():
    """OPFX_ARR_BR_ISLAND marks exactly the branches whose outage cuts a bus off every REF bus."""
    from opfgym_amd import capi, grids
    from opfgym_amd.case import net_to_case
    from helpers import non_bridge_branches
    for code in ('1-MV-urban--0-sw', '1-HV-urban--0-sw', 'hv-small'):
        net, _ = grids.get_grid(code)
        case = net_to_case(net)
        plan = capi.Plan(case)
        island = plan.array('br_island')
        assert len(island) == case.nbr
        assert set(np.flatnonzero(island == 0).tolist()) == set(non_bridge_branches(case).tolist())


@pytest.mark.parametrize('code', ['hv-small', '1-HV-mixed--0-sw', '1-MV-urban--0-sw', 'mv-3w', 'hv-small-sw'])
def test_dc_start_arrays_are_the_dc_power_flow_of_the_oracle(code, built_lib):
    """opfx_solve_opts.init = OPFX_INIT_DC: the plan lays B' (pypower makeBdc) out on the Ybus pattern next to the constant
    part of the DC right-hand side; reassembled from the descriptor arrays and solved densely it gives the angles of the
    oracle's DC start (oracle/pf_oracle.py:_dc_angles, its own branch table)."""
    from helpers import OracleSide
    from opfgym_amd import capi, grids
    from opfgym_amd.case import bus_injections, net_to_case
    from oracle import pf_oracle as po
    net, _ = grids.get_grid(code)
    case = net_to_case(net)
    plan = capi.Plan(case)
    assert plan.info['has_dc'] == 1
    ka, ra, nb = plan.info['lp_ell_width'], plan.info['lp_rounds_a'], case.nb
    dc = plan.darray('LP_DC').reshape(ra, ka + 2, 64)
    ent = (plan.array('LP_A_ENT').astype(np.int64) & 0xFFFFFFFF).reshape(ra, ka, 64)
    hdc, hent = plan.darray('LP_H_DC'), plan.array('LP_H_ENT').astype(np.int64) & 0xFFFFFFFF
    hrow = plan.array('LP_H_ROW').astype(np.int64) & 0xFFFFFFFF
    b, cst = np.zeros((nb, nb)), np.zeros(nb)
    for i in range(nb):
        r, lane = divmod(i, 64)
        for k in range(ka):
            j = int(ent[r, k, lane] & 0xFFFF)
            if j != i:
                b[i, j] += dc[r, k, lane]
        b[i, i], cst[i] = dc[r, ka, lane], dc[r, ka + 1, lane]
    for q in range(len(hent)):
        if (hent[q] & 0xFFFF) != 0xFFFF:
            b[int(hrow[q]), int(hent[q] & 0xFFFF)] += hdc[q]
    assert np.allclose(b, b.T, rtol=0, atol=1e-12) and np.allclose(b.sum(axis=1), 0, rtol=0, atol=1e-9 * np.abs(b).max())
    p, *_ = bus_injections(net, case)
    ref, free = np.flatnonzero(case.bus_type == 3), np.flatnonzero(case.bus_type != 3)
    theta = np.zeros(nb)
    theta[ref] = case.va_set[ref]
    theta[free] = np.linalg.solve(b[np.ix_(free, free)], (p / case.base_mva - cst)[free])
    side = OracleSide(net, case)
    v0 = po.start_voltage(side.ppc, 'dc')[side.bus_map]
    assert np.abs(np.angle(v0 * np.exp(-1j * theta))).max() < 1e-10


def test_the_boundary_is_usable_from_plain_c99(built_lib, tmp_path):
    """tests/native/abi_c99.c: the header compiled as C99 (-pedantic -Werror) and linked against libopfx.so — OPFX_INIT'ed
    structs, a two-bus case through plan creation / read-back, refusal of a wrong struct_size, the developer entry point,
    and opfx_ctx_create returning a status (OPFX_ERR_NO_DEVICE here) instead of crashing.  What a cgo / JNI / cffi binding
    of the reference's maintainers would do, without Python in between."""
    import shutil
    import subprocess
    if shutil.which('gcc') is None:
        pytest.skip('needs gcc')
    exe = tmp_path / 'abi_c99'
    lib_dir = os.path.dirname(capi.LIB_PATH)
    r = subprocess.run(['gcc', '-std=c99', '-pedantic', '-Wall', '-Wextra', '-Werror', '-I', os.path.join(ROOT, 'include'),
                        os.path.join(ROOT, 'tests', 'native', 'abi_c99.c'), '-o', str(exe), '-L', lib_dir, '-lopfx',
                        f'-Wl,-rpath,{lib_dir}', '-Wl,--allow-shlib-undefined'], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and 'abi ok' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
