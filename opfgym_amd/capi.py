"""ctypes binding of libopfx (include/opfx.h).

The reference project is pure Python and reaches native solvers through
third-party wheels; `cffi` is not available here, so the thin FFI layer is
`ctypes` over the C ABI.  There is deliberately NO fallback: if the library is
missing or a GPU entry point fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# OPFX_LIB: developer override (the diagnostic build with cycle stamps, __graft_entry__.build_stamps)
LIB_PATH = os.path.abspath(os.environ['OPFX_LIB']) if os.environ.get('OPFX_LIB') else os.path.join(_HERE, 'libopfx.so')

OK = 0
PQ, PV, REF = 1, 2, 3
SRC_X, SRC_RESULT = 0, 1
COST_UNIT, COST_EXT_GRID, COST_GEN = 0, 1, 2
REWARD_SUMMATION, REWARD_REPLACEMENT, REWARD_PARAMETERIZED, REWARD_ONLY_OBJECTIVE = 0, 1, 2, 3
OP_SET_CONST, OP_AFFINE, OP_SQRT_DIFF, OP_NEG, OP_UNIFORM, OP_NORMAL, OP_CLIP, OP_DIV, OP_NORMINV, OP_TRUNCNORM = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9

ARRAYS = ['Y_PTR', 'Y_COL', 'Y_BLK', 'DIAG_BLK', 'FILL_BLK', 'LEV_TPTR', 'TGT_BLK',
          'TGT_SPTR', 'SRC_IK', 'SRC_KK', 'SRC_KJ', 'LEV_PPTR', 'PIV_BUS', 'PIV_UPTR',
          'U_BLK', 'U_COL', 'BLK_ROW', 'BLK_COL', 'LP_A_ENT', 'LP_A_DBLK', 'LP_H_ENT', 'LP_H_ROW',
          'LP_B', 'LP_C', 'BR_ISLAND', 'ISL_PTR', 'ISL_BUS', 'LP_TEAM2', 'LP_TEAM4', 'TAIL_BUS', 'TAIL_IDS', 'LP_B2',
          'LP_BCC', 'LP_TEAMC2', 'LP_TEAMC4', 'LP_B3', 'ZERO_LEV', 'ZERO_BLK']
DARRAYS = ['LP_A_Y', 'LP_A_YDIAG', 'LP_H_Y', 'LP_DC', 'LP_H_DC']

_pd = C.POINTER(C.c_double)
_pi = C.POINTER(C.c_int32)
_pu8 = C.POINTER(C.c_uint8)


class OpfxError(RuntimeError):
    pass


# the version of include/opfx.h these ctypes structs were written for; lib() refuses a library of another major.minor
ABI_VERSION = (0, 3)


class Sized(C.Structure):
    """A struct of include/opfx.h: first member `struct_size`, stamped on construction (opfx.h, VERSIONING)."""

    def __init__(self, *args, **kw):
        super().__init__(C.sizeof(type(self)), *args, **kw)


class CaseStruct(Sized):
    _fields_ = [('struct_size', C.c_uint32), ('nb', C.c_int32), ('nbr', C.c_int32), ('base_mva', C.c_double),
                ('bus_type', _pi), ('vm_set', _pd), ('va_set', _pd), ('gs', _pd), ('bs', _pd),
                ('br_f', _pi), ('br_t', _pi), ('br_y', _pd), ('br_kf', _pd), ('br_kt', _pd),
                ('br_bdc', _pd), ('br_pfinj', _pd), ('elim_last', _pi)]


class PlanInfo(Sized):
    _fields_ = [('struct_size', C.c_uint32)] + [(n, C.c_int32) for n in (
        'nb', 'nbr', 'nref', 'npv', 'npq', 'nnz_y', 'nnz_j', 'n_blk', 'n_fill', 'n_levels',
        'n_targets', 'n_sources', 'n_uterms', 'max_level_width', 'lds_doubles',
        'lp_rounds_a', 'lp_rounds_h', 'lp_rounds_b', 'lp_rounds_c', 'n_full',
        'team_rounds_2', 'team_rounds_4', 'team_barriers_2', 'team_barriers_4', 'n_groups',
        'team_kb_2', 'team_kb_4', 'tail_m', 'lp_ell_width', 'has_dc', 'lp_rounds_f', 'team_rounds_chord_2',
        'team_rounds_chord_4', 'team_barriers_chord_2', 'team_barriers_chord_4', 'team_kb_chord_2', 'team_kb_chord_4',
        'lp_rounds_f_pad', 'n_shared')]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_ if n != 'struct_size'}


class SolveOpts(Sized):
    _fields_ = [('struct_size', C.c_uint32), ('reserved0', C.c_int32),
                ('tol', C.c_double), ('max_iter', C.c_int32), ('enforce_q_lims', C.c_int32),
                ('init', C.c_int32), ('contingency_start', C.c_int32), ('jacobian_reuse_tol', C.c_double)]

    def __init__(self, tol=1e-8, max_iter=10, enforce_q_lims=0, init=0, contingency_start=0, jacobian_reuse_tol=0.0):
        super().__init__(0, float(tol), int(max_iter), int(enforce_q_lims), int(init), int(contingency_start),
                         float(jacobian_reuse_tol))


INIT = {'flat': 0, 'dc': 1}


class EnvDesc(Sized):
    _fields_ = [
        ('struct_size', C.c_uint32), ('nx', C.c_int32),
        ('pinj_ptr', _pi), ('pinj_slot', _pi), ('pinj_coef', _pd),
        ('qinj_ptr', _pi), ('qinj_slot', _pi), ('qinj_coef', _pd),
        ('qg_min', _pd), ('qg_max', _pd),
        ('na', C.c_int32),
        ('act_slot', _pi), ('act_scaling', _pd), ('act_lo_slot', _pi), ('act_hi_slot', _pi),
        ('act_lo_const', _pd), ('act_hi_const', _pd),
        ('clamp_lo_slot', _pi), ('clamp_hi_slot', _pi), ('clamp_lo_const', _pd), ('clamp_hi_const', _pd),
        ('clamp_enabled', C.c_int32),
        ('diff_action_step_size', C.c_double), ('clipped_action_penalty', C.c_double),
        ('npoly', C.c_int32), ('npwl', C.c_int32), ('nseg', C.c_int32),
        ('cost_kind', _pi), ('cost_pidx', _pi), ('cost_qidx', _pi), ('cost_scale', _pd),
        ('pwl_is_q', _pi), ('cost_coef', _pd),
        ('nprice', C.c_int32), ('price_slot', _pi), ('price_coef', _pi),
        ('nc', C.c_int32), ('con_ptr', _pi), ('con_src', _pi), ('con_min', _pd), ('con_max', _pd),
        ('con_autoscale', _pd), ('con_penalty_factor', _pd), ('con_penalty_power', _pd),
        ('con_count_penalty', _pd), ('con_worst_case', _pi),
        ('reward_kind', C.c_int32), ('penalty_weight', C.c_double),
        ('clip_lo', C.c_double), ('clip_hi', C.c_double),
        ('objective_factor', C.c_double), ('objective_bias', C.c_double),
        ('penalty_factor', C.c_double), ('penalty_bias', C.c_double),
        ('valid_reward', C.c_double), ('invalid_penalty', C.c_double),
        ('invalid_objective_share', C.c_double), ('diff_objective', C.c_int32),
        ('nobs', C.c_int32), ('obs_kind', _pi), ('obs_idx', _pi),
        ('steps_per_episode', C.c_int32),
        ('n_cont', C.c_int32), ('cont_branch', _pi), ('not_converged_penalty', C.c_double),
        ('act_kind', _pi),
        ('n_bmod', C.c_int32), ('bmod_branch', _pi), ('bmod_slot', _pi), ('bmod_lo', _pi),
        ('bmod_n', _pi), ('bmod_ptr', _pi), ('bmod_y', _pd),
        ('vset_slot', _pi),
        ('n_qterm', C.c_int32), ('qterm_idx', _pi), ('qterm_target', _pd), ('qterm_weight', _pd),
        ('n_xres', C.c_int32), ('xres_kind', _pi), ('xres_p', _pi), ('xres_q', _pi), ('xres_scale', _pd),
                ('xres_r', _pi), ('cost_bus', _pi),
                ('xres_offset', _pd), ('cost_pres', _pi), ('cost_qres', _pi)]        # (appended in 0.3.1)


ACT_CONTINUOUS, ACT_INTEGER, ACT_BOOLEAN = 0, 1, 2
XRES_P, XRES_S, XRES_MAX3, XRES_AFFINE = 0, 1, 2, 3


class StepIO(Sized):
    _fields_ = [('struct_size', C.c_uint32), ('reserved0', C.c_uint32), ('x', C.c_void_p), ('action', C.c_void_p), ('initial_obj', C.c_void_p),
                ('step_in_episode', C.c_void_p), ('outage', C.c_void_p),
                ('obs', C.c_void_p), ('reward', C.c_void_p), ('terminated', C.c_void_p),
                ('truncated', C.c_void_p), ('valids', C.c_void_p), ('violations', C.c_void_p),
                ('penalties', C.c_void_p), ('cost', C.c_void_p), ('objective', C.c_void_p),
                ('results', C.c_void_p), ('mean_correction', C.c_void_p),
                ('converged', C.c_void_p), ('iterations', C.c_void_p), ('max_mismatch', C.c_void_p),
                ('total_iterations', C.c_void_p), ('min_pivot', C.c_void_p), ('min_pivot_bus', C.c_void_p)]


class ProfileDesc(Sized):
    _fields_ = [('struct_size', C.c_uint32), ('n_steps', C.c_int32), ('n_types', C.c_int32), ('n_cols', C.c_int32),
                ('rel', _pd), ('typ', _pi), ('peak', _pd), ('slot', _pi),
                ('col_min', _pd), ('col_max', _pd)]


class ResetDesc(Sized):
    _fields_ = [('struct_size', C.c_uint32), ('n_tables', C.c_int32), ('tables', C.POINTER(ProfileDesc)),
                ('n_ops', C.c_int32), ('op_code', _pi), ('op_dst', _pi), ('op_a', _pi),
                ('op_n', _pi), ('op_c0', _pi), ('op_c1', _pi), ('op_c2', _pi),
                ('n_consts', C.c_int32), ('consts', _pd), ('n_uniform', C.c_int32),
                ('init_off', C.c_int32), ('n_normal', C.c_int32), ('op_mode', _pi)]


class ResetIO(Sized):
    _fields_ = [('struct_size', C.c_uint32), ('reserved0', C.c_uint32), ('step_idx', C.c_void_p), ('noise', C.c_void_p), ('interp', C.c_void_p),
                ('uniform', C.c_void_p), ('normal', C.c_void_p), ('normal_noise_factor', C.c_double),
                ('x', C.c_void_p), ('mode', C.c_void_p), ('action', C.c_void_p), ('obs', C.c_void_p),
                ('keep_state', C.c_int32), ('step_pool', C.c_void_p), ('n_step_pool', C.c_int32),
                ('rng_seed', C.c_uint64), ('step_out', C.c_void_p)]


class DebugOpts(Sized):
    """include/opfx_debug.h: developer switches, handed explicitly to the *_debug constructors (0 = library default)."""
    _fields_ = [('struct_size', C.c_uint32)] + [(n, C.c_int32) for n in (
        'plan_search', 'plan_dcap_slack', 'plan_seed', 'plan_no_bank', 'plan_no_pack', 'plan_no_riders', 'plan_no_tail',
        'plan_no_pairs',
        'team', 'queue', 'packed', 'force_mem', 'kernel_v1', 'waves_per_cu', 'verbose', 'stamps',
        'reset_team', 'plan_share_slots', 'no_rank1_dc')]

    def any(self):
        return any(getattr(self, n) for n, _ in self._fields_ if n != 'struct_size')

    def as_dict(self) -> dict:
        return {n: int(getattr(self, n)) for n, _ in self._fields_ if n != 'struct_size' and getattr(self, n)}


_default_debug = None


def set_default_debug(debug=None) -> None:
    """What `debug=None` means from now on in this process (None: all zero again).  For harness scripts that build many
    objects and are steered from a shell loop: `capi.set_default_debug(capi.debug_from_env())` as their first statement.
    The package itself never calls it; tests pass `debug=` explicitly."""
    global _default_debug
    _default_debug = None if debug is None else debug_opts(debug)


def debug_opts(debug=None) -> DebugOpts:
    """`debug=` arguments of this package as the struct: None (all zero: the library's own choices — unless a harness script
    has called `set_default_debug`), a dict of field names
    (`dict(team=2, queue=-1)`) or a DebugOpts (copied).  The package never reads the process environment for these;
    scripts and A/B harnesses that are steered by OPFX_* variables call `debug_from_env()` themselves and pass the result."""
    if debug is None:
        return DebugOpts.from_buffer_copy(_default_debug) if _default_debug is not None else DebugOpts()
    if isinstance(debug, DebugOpts):
        return DebugOpts.from_buffer_copy(debug)
    d = DebugOpts()
    for k, v in dict(debug).items():
        if k not in {n for n, _ in DebugOpts._fields_} or k == 'struct_size':
            raise KeyError(f'opfx_debug_opts has no member {k!r}')
        setattr(d, k, int(v))
    return d


def debug_from_env(environ=None) -> DebugOpts:
    """Developer switches as OPFX_* variables of the calling process, turned into the explicit struct the library takes:
    the ONE place of this package that reads them, and nothing in the package calls it — scripts/ (profiling, A/B and fuzz
    harnesses, which are steered from shell loops) do, and hand the result on as `debug=`.  Neither the C library nor
    `Plan` / `Context` / `BatchedOpfEnv` read the environment themselves.  Tri-state switches: '1' on, '0' off, unset =
    automatic."""
    e = os.environ if environ is None else environ
    d = DebugOpts()

    def flag(name):
        return 1 if e.get(name) not in (None, '', '0') else 0

    def tri(name):
        return 0 if e.get(name) in (None, '') else (1 if int(e[name]) != 0 else -1)
    if e.get('OPFX_PLAN_SEARCH') not in (None, ''):
        d.plan_search = int(e['OPFX_PLAN_SEARCH']) or -1
    d.plan_dcap_slack = int(e.get('OPFX_PLAN_DCAP_SLACK') or 0)
    d.plan_seed = int(e.get('OPFX_PLAN_SEED') or 0)
    d.plan_no_bank, d.plan_no_pack = flag('OPFX_PLAN_NO_BANK'), flag('OPFX_PLAN_NO_PACK')
    d.plan_no_riders, d.plan_no_tail = flag('OPFX_NO_RIDERS'), flag('OPFX_NO_TAIL')
    d.plan_no_pairs = int(e.get('OPFX_PLAN_NO_PAIRS') or 0)
    d.team = int(e.get('OPFX_TEAM') or 0)
    d.queue, d.packed = tri('OPFX_QUEUE'), tri('OPFX_PACKED')
    d.force_mem, d.kernel_v1 = flag('OPFX_FORCE_MEM'), flag('OPFX_KERNEL_V1')
    d.waves_per_cu = int(e.get('OPFX_WAVES_PER_CU') or 0)
    d.verbose, d.stamps = flag('OPFX_VERBOSE'), flag('OPFX_STAMPS')
    d.reset_team = int(e.get('OPFX_RESET_TEAM') or 0)
    d.plan_share_slots = int(e.get('OPFX_PLAN_SHARE') or 0)
    d.no_rank1_dc = flag('OPFX_NO_RANK1_DC')
    return d


_lib = None

EXPORTS = ['opfx_plan_create', 'opfx_plan_destroy', 'opfx_plan_get_info', 'opfx_plan_get_array',
           'opfx_plan_get_ybus', 'opfx_plan_get_darray', 'opfx_ctx_create', 'opfx_ctx_destroy', 'opfx_last_error',
           'opfx_version', 'opfx_solve', 'opfx_env_create', 'opfx_env_destroy', 'opfx_step',
           'opfx_env_set_reset', 'opfx_reset', 'opfx_time_steps', 'opfx_env_get_info', 'opfx_env_get_storage',
           'opfx_env_get_spec', 'opfx_env_get_row_io', 'opfx_env_prepare']
# include/opfx_debug.h (developer entry points, not part of the boundary)
DEBUG_EXPORTS = ['opfx_plan_create_debug', 'opfx_ctx_create_debug', 'opfx_debug_read_stamps', 'opfx_debug_read_finish']


def lib():
    """Load libopfx.so (built in-tree by __graft_entry__.build()); fail loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OpfxError(f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; '
                        f'g.build()"` (hipcc --offload-arch=gfx950). There is no CPU fallback.')
    # PyTorch-ROCm bundles its own HIP runtime (same SONAME as /opt/rocm's).  It has to be
    # in the process BEFORE libopfx.so so that the loader binds both to ONE runtime: device
    # pointers and hipStream_t handles are then interchangeable between torch and libopfx.
    # (The other order makes torch fail with "No HIP GPUs are available".)
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.opfx_last_error.restype = C.c_char_p
    L.opfx_version.argtypes = [C.POINTER(C.c_int)] * 3
    L.opfx_version.restype = None
    v = [C.c_int(), C.c_int(), C.c_int()]
    L.opfx_version(*[C.byref(x) for x in v])
    if (v[0].value, v[1].value) != ABI_VERSION:
        raise OpfxError(f'{LIB_PATH} is libopfx {v[0].value}.{v[1].value}.{v[2].value}, this binding was written for '
                        f'{ABI_VERSION[0]}.{ABI_VERSION[1]}.x (struct layouts differ): rebuild it with __graft_entry__.build()')
    L.opfx_plan_create_debug.argtypes = [C.POINTER(CaseStruct), C.POINTER(DebugOpts), C.POINTER(vp)]
    L.opfx_ctx_create_debug.argtypes = [vp, C.c_int, C.POINTER(DebugOpts), C.POINTER(vp)]
    L.opfx_plan_create.argtypes = [C.POINTER(CaseStruct), C.POINTER(vp)]
    L.opfx_plan_destroy.argtypes = [vp]
    L.opfx_plan_destroy.restype = None
    L.opfx_plan_get_info.argtypes = [vp, C.POINTER(PlanInfo)]
    L.opfx_plan_get_array.argtypes = [vp, C.c_int, _pi, C.c_int64]
    L.opfx_plan_get_array.restype = C.c_int64
    L.opfx_plan_get_ybus.argtypes = [vp, _pd, _pd]
    L.opfx_plan_get_darray.argtypes = [vp, C.c_int, _pd, C.c_int64]
    L.opfx_plan_get_darray.restype = C.c_int64
    L.opfx_ctx_create.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.opfx_ctx_destroy.argtypes = [vp]
    L.opfx_ctx_destroy.restype = None
    L.opfx_solve.argtypes = [vp, C.c_int64] + [vp] * 5 + [C.POINTER(SolveOpts)] + [vp] * 11
    L.opfx_env_create.argtypes = [vp, C.POINTER(EnvDesc), C.POINTER(vp)]
    L.opfx_env_destroy.argtypes = [vp]
    L.opfx_env_destroy.restype = None
    L.opfx_step.argtypes = [vp, C.c_int64, C.POINTER(StepIO), C.POINTER(SolveOpts), C.c_int32, vp]
    L.opfx_env_set_reset.argtypes = [vp, C.POINTER(ResetDesc)]
    L.opfx_reset.argtypes = [vp, C.c_int64, C.POINTER(ResetIO), vp]
    L.opfx_time_steps.argtypes = [vp, C.c_int64, C.POINTER(StepIO), C.POINTER(SolveOpts), C.c_int32,
                                  vp, C.POINTER(C.c_float)]
    L.opfx_env_get_info.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
    L.opfx_env_get_storage.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.opfx_env_get_spec.argtypes = [vp, C.POINTER(C.c_int32)]
    L.opfx_env_get_row_io.argtypes = [vp, C.POINTER(C.c_int32)]
    L.opfx_env_prepare.argtypes = [vp, C.c_int64]
    _lib = L
    return L


def check(rc, what=''):
    if rc != OK:
        msg = lib().opfx_last_error()
        raise OpfxError(f'{what} failed with status {rc}: {msg.decode() if msg else ""}')


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_pd)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_pi)


class Plan:
    """Host-side compiled grid plan (opfx_plan)."""

    def __init__(self, case, debug: DebugOpts = None, elim_last=None):
        """elim_last: bus indices (of the case) to eliminate after all others — the rescue ordering for a breakdown of
        the static pivoting at those buses (opfx_case.elim_last, `min_pivot_bus`)."""
        self.case = case
        self.elim_last = None if elim_last is None else sorted(int(b) for b in elim_last)
        self.debug = debug_opts(debug)
        keep = []
        cs = CaseStruct()
        cs.nb, cs.nbr, cs.base_mva = case.nb, case.nbr, float(case.base_mva)
        for name, arr, conv in (('bus_type', case.bus_type, _i), ('vm_set', case.vm_set, _d),
                                ('va_set', case.va_set, _d), ('gs', case.gs, _d), ('bs', case.bs, _d),
                                ('br_f', case.f, _i), ('br_t', case.t, _i),
                                ('br_kf', case.kf, _d), ('br_kt', case.kt, _d)):
            a, p = conv(arr)
            keep.append(a)
            setattr(cs, name, p)
        y = np.stack([case.yff.real, case.yff.imag, case.yft.real, case.yft.imag,
                      case.ytf.real, case.ytf.imag, case.ytt.real, case.ytt.imag], axis=1)
        ya, yp = _d(y)
        keep.append(ya)
        cs.br_y = yp
        # (a zero-reactance branch has no finite 1/x: such a case gets no DC start — `has_dc` 0, init='auto' stays flat —
        #  instead of a B' with 1.8e308 in it)
        if getattr(case, 'bdc', None) is not None and getattr(case, 'pfinj', None) is not None \
                and np.isfinite(np.asarray(case.bdc, dtype=float)).all() and np.isfinite(np.asarray(case.pfinj, dtype=float)).all():
            for name, arr in (('br_bdc', case.bdc), ('br_pfinj', case.pfinj)):
                a, p = _d(np.asarray(arr, dtype=float))
                keep.append(a)
                setattr(cs, name, p)
        if self.elim_last:
            flags = np.zeros(case.nb, dtype=np.int32)
            flags[self.elim_last] = 1
            a, p = _i(flags)
            keep.append(a)
            cs.elim_last = p
        self.branch_y = ya
        self.case_struct, self._keep = cs, keep
        h = C.c_void_p()
        if self.debug.any():
            check(lib().opfx_plan_create_debug(C.byref(cs), C.byref(self.debug), C.byref(h)), 'opfx_plan_create_debug')
        else:
            check(lib().opfx_plan_create(C.byref(cs), C.byref(h)), 'opfx_plan_create')
        self.handle = h
        info = PlanInfo()
        check(lib().opfx_plan_get_info(h, C.byref(info)), 'opfx_plan_get_info')
        self.info = info.as_dict()

    def array(self, name) -> np.ndarray:
        which = ARRAYS.index(name.upper())
        n = lib().opfx_plan_get_array(self.handle, which, None, 0)
        if n < 0:
            check(int(n), 'opfx_plan_get_array')
        out = np.zeros(int(n), dtype=np.int32)
        lib().opfx_plan_get_array(self.handle, which, out.ctypes.data_as(_pi), n)
        return out

    def darray(self, name) -> np.ndarray:
        which = DARRAYS.index(name.upper())
        n = lib().opfx_plan_get_darray(self.handle, which, None, 0)
        if n < 0:
            check(int(n), 'opfx_plan_get_darray')
        out = np.zeros(int(n))
        lib().opfx_plan_get_darray(self.handle, which, out.ctypes.data_as(_pd), n)
        return out

    def ybus(self):
        g = np.zeros(self.info['nnz_y'])
        b = np.zeros(self.info['nnz_y'])
        check(lib().opfx_plan_get_ybus(self.handle, g.ctypes.data_as(_pd), b.ctypes.data_as(_pd)))
        return g, b

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                lib().opfx_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class Context:
    """Plan resident on one GPU (opfx_ctx)."""

    def __init__(self, plan: Plan, device: int = 0, debug: DebugOpts = None):
        self.plan = plan
        self.device = device
        self.debug = debug_opts(debug)
        h = C.c_void_p()
        if self.debug.any():
            check(lib().opfx_ctx_create_debug(plan.handle, int(device), C.byref(self.debug), C.byref(h)), 'opfx_ctx_create_debug')
        else:
            check(lib().opfx_ctx_create(plan.handle, int(device), C.byref(h)), 'opfx_ctx_create')
        self.handle = h

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                lib().opfx_ctx_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def _ptr(t):
    """Device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    assert t.is_contiguous()
    return C.c_void_p(t.data_ptr())


def _stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def solve(ctx: Context, p_inj, q_inj, *, qg_min=None, qg_max=None, outage=None, tol=1e-8,
          max_iter=10, enforce_q_lims=False, want=('vm', 'va', 'loading', 's_ref', 'q_gen'), init='flat',
          jacobian_reuse_tol=0.0):
    """Batched power flow on torch CUDA tensors p_inj/q_inj [B, nb] (p.u.)."""
    import torch
    assert p_inj.is_cuda and p_inj.dtype == torch.float64 and p_inj.shape == q_inj.shape
    B, nb = p_inj.shape
    info = ctx.plan.info
    assert nb == info['nb']
    dev = p_inj.device
    out = {}
    if 'vm' in want:
        out['vm'] = torch.empty(B, nb, dtype=torch.float64, device=dev)
    if 'va' in want:
        out['va'] = torch.empty(B, nb, dtype=torch.float64, device=dev)
    if 'loading' in want:
        out['loading'] = torch.empty(B, info['nbr'], dtype=torch.float64, device=dev)
    if 's_ref' in want:
        out['s_ref'] = torch.empty(B, info['nref'], 2, dtype=torch.float64, device=dev)
    if 'q_gen' in want:
        out['q_gen'] = torch.empty(B, nb, dtype=torch.float64, device=dev)
    out['converged'] = torch.empty(B, dtype=torch.uint8, device=dev)
    out['iterations'] = torch.empty(B, dtype=torch.int32, device=dev)
    out['max_mismatch'] = torch.empty(B, dtype=torch.float64, device=dev)
    out['min_pivot'] = torch.empty(B, dtype=torch.float64, device=dev)
    out['min_pivot_bus'] = torch.empty(B, dtype=torch.int32, device=dev)
    opts = SolveOpts(tol, max_iter, int(bool(enforce_q_lims)), INIT[init], 0, jacobian_reuse_tol)
    with torch.cuda.device(dev):
        check(lib().opfx_solve(
            ctx.handle, B, _ptr(p_inj.contiguous()), _ptr(q_inj.contiguous()), _ptr(qg_min), _ptr(qg_max),
            _ptr(outage), C.byref(opts), _ptr(out.get('vm')), _ptr(out.get('va')),
            _ptr(out.get('loading')), _ptr(out.get('s_ref')), _ptr(out.get('q_gen')), _ptr(out['converged']),
            _ptr(out['iterations']), _ptr(out['max_mismatch']), _ptr(out['min_pivot']), _ptr(out['min_pivot_bus']),
            _stream()), 'opfx_solve')
    return out
