/*
 * opfx.h — C ABI of libopfx: MI355X-native batched AC power-flow + OPF-environment
 * evaluation backend for opfgym.
 *
 * This is the drop-in boundary.  Every entry point replaces a
 * piece of the reference's per-step Python path; citations are into
 * /root/reference/opfgym/:
 *
 *   opfx_plan_create / opfx_ctx_create
 *        one-time per grid; replaces the per-call structure work pandapower
 *        redoes inside every `pp.runpp` (opf_env.py:703): bus typing, Ybus
 *        assembly, Jacobian pattern, ordering and symbolic factorisation.
 *   opfx_solve
 *        replaces `self._run_power_flow(self.net)` (opf_env.py:657,
 *        security_constrained.py:53) for B instances at once: Newton-Raphson
 *        on given bus injections, returns V and branch loadings.
 *   opfx_env_create
 *        one-time; flattens what OpfEnv.__init__ wires up (opf_env.py:27-175):
 *        action keys, observation keys, constraints (constraints.py:195-226),
 *        cost tables (objective.py:6-87) and the reward function (reward.py).
 *   opfx_step
 *        replaces OpfEnv.step (opf_env.py:374-419) for B instances: apply
 *        actions (:421-491), power flow (:646-662), objective (:493-500),
 *        violations (:502-513), reward (:515-530), observation (:532-549).
 *   opfx_reset
 *        replaces OpfEnv.reset/_sampling/_set_simbench_state (opf_env.py:177-
 *        372) and the benchmark envs' `_sampling` tails (voltage_control.py:
 *        111-133, eco_dispatch.py:111-123, max_renewable.py:101-105).
 *
 * Conventions
 *   - plain C, no exceptions cross the boundary: every function returns
 *     OPFX_OK (0) or a negative opfx_status; opfx_last_error() gives text.
 *   - all `double*`/`int32_t*` data arguments of opfx_solve/opfx_step/
 *     opfx_reset are DEVICE pointers (HIP), row-major, instance-major
 *     ([B, n]); the caller owns them.  Plan/env descriptors are HOST pointers
 *     and are copied during the call.
 *   - work is enqueued on the caller's stream (`void* stream` = hipStream_t,
 *     NULL = default stream) and is asynchronous; non-convergence is DATA
 *     (converged[B], iterations[B], max_mismatch[B]), never an error code.
 *   - one context per device; calls on one context are serialised by the
 *     caller — in stream order too: launches of one context must not overlap
 *     on the device (they share its scratch rows); no hidden global state:
 *     the library reads no environment variable (developer switches travel in
 *     an explicit struct, include/opfx_debug.h).
 *   - VERSIONING.  Every struct that crosses the boundary starts with
 *     `uint32_t struct_size`, which the CALLER sets to sizeof(the struct) of the
 *     header it was compiled against (OPFX_INIT(x) does it).  The library
 *     refuses a size it does not know (OPFX_ERR_INVALID with text) instead of
 *     reading past the caller's struct; new members are only ever APPENDED, and
 *     a struct that is shorter than the library's but not shorter than the
 *     first layout of this major.minor series is accepted with the missing
 *     members zero (= their documented defaults).  opfx_version() reports the
 *     library's version; bindings compare major and minor on load
 *     (opfgym_amd/capi.py does).
 */
#ifndef OPFX_H
#define OPFX_H

#include <stdint.h>
#include <string.h>   /* memset (OPFX_INIT) */

#ifdef __cplusplus
extern "C" {
#endif

#define OPFX_VERSION_MAJOR 0
#define OPFX_VERSION_MINOR 3      /* 0.2: struct_size in every struct, developer switches out of the environment;
                                   * 0.3: members appended to opfx_case / opfx_step_io / opfx_solve_opts / opfx_plan_info and
                                   *      min_pivot_bus added to opfx_solve after the first 0.2 layout -> a new series */
#define OPFX_VERSION_PATCH 1      /* 0.3.1: xres_offset / cost_pres / cost_qres appended to opfx_env_desc, OPFX_XRES_AFFINE */

/* zero a struct of this header and stamp its size: opfx_solve_opts o; OPFX_INIT(o); o.tol = 1e-8; ... */
#define OPFX_INIT(x) do { memset(&(x), 0, sizeof(x)); (x).struct_size = (uint32_t)sizeof(x); } while (0)

typedef enum opfx_status {
  OPFX_OK = 0,
  OPFX_ERR_INVALID = -1,     /* bad argument / inconsistent descriptor   */
  OPFX_ERR_HIP = -2,         /* a HIP runtime call failed                 */
  OPFX_ERR_NO_DEVICE = -3,   /* no usable GPU                             */
  OPFX_ERR_TOO_LARGE = -4,   /* grid does not fit the LDS-resident kernel */
  OPFX_ERR_SINGULAR = -5     /* structurally singular case (no slack …)   */
} opfx_status;

enum { OPFX_PQ = 1, OPFX_PV = 2, OPFX_REF = 3 };

/* ---- per-unit grid description (one topology, shared by the whole batch) ---
 * Mirrors the pypower/pandapower `ppci` the reference builds inside runpp
 * (SURVEY §8a rows P2/P3). */
typedef struct opfx_case {
  uint32_t struct_size;      /* = sizeof(opfx_case), see VERSIONING above    */
  int32_t nb;                /* buses                                        */
  int32_t nbr;               /* branches (lines + transformers), in service  */
  double base_mva;
  const int32_t* bus_type;   /* [nb] OPFX_PQ / OPFX_PV / OPFX_REF            */
  const double* vm_set;      /* [nb] |V| set-point of PV/REF buses, start value else */
  const double* va_set;      /* [nb] rad: REF angle; start angle for the others      */
  const double* gs;          /* [nb] bus shunt conductance, p.u.             */
  const double* bs;          /* [nb] bus shunt susceptance, p.u.             */
  const int32_t* br_f;       /* [nbr] from bus                               */
  const int32_t* br_t;       /* [nbr] to bus                                 */
  const double* br_y;        /* [nbr*8] yff.re,yff.im,yft.re,yft.im,ytf.re,ytf.im,ytt.re,ytt.im */
  const double* br_kf;       /* [nbr] loading_percent = max(|If|*kf, |It|*kt) */
  const double* br_kt;       /* [nbr]                                        */
  /* optional, for opfx_solve_opts.init = OPFX_INIT_DC (pypower makeBdc): per branch b = 1 / (x * |tap|) in p.u. (0
   * for a branch that couples nothing) and its phase-shift injection b * (-shift [rad]); NULL: no DC start */
  const double* br_bdc;      /* [nbr] */
  const double* br_pfinj;    /* [nbr] */
  /* optional [nb] flags (NULL: none): buses to eliminate AFTER all others.  The block LU pivots statically — a fixed
   * elimination order, the 2x2 diagonal block of each bus as its pivot — where pandapower's SuperLU pivots partially; a
   * bus whose own diagonal block is (numerically) singular at its turn breaks the factorisation although the Jacobian is
   * regular (min_pivot ~ 0, min_pivot_bus names it).  Eliminated last, the same bus meets a diagonal block that carries
   * the Schur complement of everything else: the rescue plan of `on_pivot_breakdown='resolve'` (opfgym_amd/batched_env.py). */
  const int32_t* elim_last;
} opfx_case;

typedef struct opfx_plan opfx_plan;   /* host-side compiled structure        */
typedef struct opfx_ctx opfx_ctx;     /* plan resident on one GPU            */
typedef struct opfx_env opfx_env;     /* environment evaluator on a context  */

typedef struct opfx_plan_info {
  uint32_t struct_size;      /* set by the caller BEFORE opfx_plan_get_info: the library fills that many bytes */
  int32_t nb, nbr, nref, npv, npq;
  int32_t nnz_y;             /* Ybus block entries (incl. diagonal)          */
  int32_t nnz_j;             /* scalar Jacobian non-zeros, pypower layout    */
  int32_t n_blk;             /* 2x2 LU blocks (Jacobian pattern + fill)      */
  int32_t n_fill;            /* fill blocks                                  */
  int32_t n_levels;          /* elimination levels                           */
  int32_t n_targets;         /* forward update work items                    */
  int32_t n_sources;         /* forward update terms                         */
  int32_t n_uterms;          /* backward substitution terms                  */
  int32_t max_level_width;   /* widest level in work items                   */
  int32_t lds_doubles;       /* solver state per instance in LDS (doubles), two-value block storage */
  /* register-resident lane programme (rounds of 64 work items); -1 = not built */
  int32_t lp_rounds_a, lp_rounds_h, lp_rounds_b, lp_rounds_c;
  int32_t n_full;            /* blocks [0,n_full) store 4 values, [n_full,n_blk) two (a,b of [[a,b],[-b,a]]) */
  /* wave-team form of the factor/solve stream, [0]: teams of 2 wavefronts, [1]: of 4 */
  int32_t team_rounds[2];    /* rounds each wavefront of a team walks per NR iteration            */
  int32_t team_barriers[2];  /* workgroup barriers among them (groups that end with one)          */
  int32_t n_groups;          /* independent groups of the stream (elimination levels of B and C)  */
  int32_t team_kb[2];        /* rounds before the dense tail's register chain (= team_rounds without a tail) */
  int32_t tail_m;            /* pivots of the dense tail (final levels with one pivot each), 0 = none */
  int32_t lp_ell_width;      /* off-diagonal Ybus entries per bus row in the row's own lane (LP_A_ENT: [ra][width][64]) */
  int32_t has_dc;            /* 1 = the case carried br_bdc / br_pfinj: opfx_solve_opts.init = OPFX_INIT_DC is available */
  /* chord steps (opfx_solve_opts.jacobian_reuse_tol): rounds of the forward substitution alone (one wavefront), and the
   * rounds / barriers per wavefront of a team's chord iteration (forward + back substitution, no factorisation) */
  int32_t lp_rounds_f;
  int32_t team_rounds_chord[2];
  int32_t team_barriers_chord[2];
  int32_t team_kb_chord[2];  /* rounds of the chord stream before the tail chain (as team_kb) */
  int32_t lp_rounds_f_pad;   /* lp_rounds_f padded to the multiple of four the stream OPFX_ARR_LP_BCC is laid out with; its
                              * back-substitution part follows with lp_rounds_c padded likewise */
  int32_t n_shared;          /* fill blocks that live in the id (= LDS slot) of a block that is dead by then (plan.cpp share_slots;
                              * 0: every block has an id of its own).  Such a plan runs on the wave-team kernels with full Newton
                              * only: the single-wave and chord streams carry no zero-at-birth operations */
} opfx_plan_info;

/* Symbolic analysis on the host (no GPU needed): bus partition, Ybus block
 * CSR, level-scheduled fill-reducing ordering on the bus graph, symbolic block
 * LU, elimination schedule. */
int opfx_plan_create(const opfx_case* c, opfx_plan** out);
void opfx_plan_destroy(opfx_plan* p);
int opfx_plan_get_info(const opfx_plan* p, opfx_plan_info* out);

/* Read-back of the compiled schedule for tests/tools.  `which` selects one of
 * the OPFX_ARR_* int32 arrays; returns its length (or copies up to `cap`
 * entries into `out` when out != NULL). */
enum {
  OPFX_ARR_Y_PTR = 0, OPFX_ARR_Y_COL, OPFX_ARR_Y_BLK, OPFX_ARR_DIAG_BLK,
  OPFX_ARR_FILL_BLK, OPFX_ARR_LEV_TPTR, OPFX_ARR_TGT_BLK, OPFX_ARR_TGT_SPTR,
  OPFX_ARR_SRC_IK, OPFX_ARR_SRC_KK, OPFX_ARR_SRC_KJ, OPFX_ARR_LEV_PPTR,
  OPFX_ARR_PIV_BUS, OPFX_ARR_PIV_UPTR, OPFX_ARR_U_BLK, OPFX_ARR_U_COL,
  OPFX_ARR_BLK_ROW, OPFX_ARR_BLK_COL,
  OPFX_ARR_LP_A_ENT, OPFX_ARR_LP_A_DBLK, OPFX_ARR_LP_H_ENT, OPFX_ARR_LP_H_ROW,
  OPFX_ARR_LP_B, OPFX_ARR_LP_C,
  OPFX_ARR_BR_ISLAND,  /* [nbr] 1 = taking this branch out cuts some bus off every REF bus */
  OPFX_ARR_ISL_PTR, OPFX_ARR_ISL_BUS,  /* CSR [nbr+1] -> the buses that outage cuts off (they are de-energised) */
  OPFX_ARR_LP_TEAM2, OPFX_ARR_LP_TEAM4, /* wave-team streams [round][wave][64][4]: w0, w1, second column, round flags (bits 0-1) | rider bits */
  OPFX_ARR_TAIL_BUS,                    /* [32] bus | diagonal block << 16 of the dense tail's pivots */
  OPFX_ARR_TAIL_IDS,                    /* [tail_m + 1][M] U-block ids inside the tail (0xFFFF none), M = tail_m rounded up to 8 */
  OPFX_ARR_LP_B2,                       /* [rb][64] right-hand-side rider of a factor item: i | k << 16 (0xFFFF both: none) */
  OPFX_ARR_LP_BCC,                      /* chord stream of the single-wave kernels: [rf_pad + rc_pad][64][4] */
  OPFX_ARR_LP_TEAMC2, OPFX_ARR_LP_TEAMC4, /* chord streams of the wave teams, laid out like OPFX_ARR_LP_TEAM2 / 4 */
  OPFX_ARR_LP_B3,                        /* [rb][64] second column of a factor item (same multiplier): target2 | A_kj2 << 16 (0xFFFF both: none) */
  OPFX_ARR_ZERO_LEV, OPFX_ARR_ZERO_BLK   /* shared slots (opfx_plan_info.n_shared): block id ZERO_BLK[q] is set to zero during elimination level
                                          * ZERO_LEV[q] — between the last read of the block that held the id and the first update of the fill
                                          * block that holds it next */
};
/* double arrays of the lane programme: Ybus values per descriptor */
enum { OPFX_DARR_LP_A_Y = 0, OPFX_DARR_LP_A_YDIAG, OPFX_DARR_LP_H_Y,
       OPFX_DARR_LP_DC,   /* [ra][width + 2][64]: B'_ij of the row's ELL entries, B'_ii, and the constant part of the DC
                           * right-hand side of bus i (phase-shift injections + shunt conductance + B'_i,ref theta_ref) */
       OPFX_DARR_LP_H_DC  /* [rh][64]: B'_ij of the overflow entries */ };
int64_t opfx_plan_get_darray(const opfx_plan* p, int which, double* out, int64_t cap);
int64_t opfx_plan_get_array(const opfx_plan* p, int which, int32_t* out, int64_t cap);
/* Ybus values in the plan's CSR order: out_g/out_b [nnz_y]. */
int opfx_plan_get_ybus(const opfx_plan* p, double* out_g, double* out_b);

/* Upload a plan to GPU `device` (hipSetDevice ordinal). */
int opfx_ctx_create(const opfx_plan* p, int device, opfx_ctx** out);
void opfx_ctx_destroy(opfx_ctx* ctx);
const char* opfx_last_error(void);
void opfx_version(int* major, int* minor, int* patch);

/* ---- pure power flow ------------------------------------------------------
 * p_inj/q_inj [B,nb]: net bus injections (generation - demand), p.u.; q_inj is
 * ignored at PV buses.  Outputs (any may be NULL): vm [B,nb] p.u., va [B,nb]
 * rad, loading [B,nbr] percent, s_ref [B,nref*2] power delivered by the slack
 * source at each REF bus (calculated injection minus p_inj/q_inj there), P,Q
 * p.u. interleaved, REF buses in increasing bus order; q_gen [B,nb] reactive
 * output of the generator at each PV bus, p.u. (0 elsewhere); converged [B],
 * iterations [B], max_mismatch [B] (final inf-norm of [dP;dQ], p.u.),
 * min_pivot [B]: the smallest RELATIVE pivot of the block-LU over all iterations,
 * |det| / (|a11 a22| + |a12 a21|) of the 2x2 diagonal block each bus is eliminated
 * with (1 = no cancellation; towards 0 the static pivoting inside the blocks is
 * breaking down and a "not converged" may be numerical, not physical; NaN from the
 * first-generation fallback kernel, which does not monitor it).  min_pivot_bus [B]: the bus that pivot belongs
 * to (-1: none recorded) — where a breakdown of the static pivoting sits, see opfx_case.elim_last.
 * q_inj at a PV bus is the reactive injection of everything EXCEPT the voltage-
 * controlling generator there (so Qgen = Qcalc - q_inj); qg_min/qg_max [nb]
 * p.u. (NULL = unlimited) are the generator capability used by enforce_q_lims.
 * outage: NULL (all branches in service) or [B] int32: the one branch that is
 * out of service in that instance, -1 = none — the N-1 axis of
 * security_constrained.py:44-66.
 * B = 0 (an empty batch) is a no-op that returns OPFX_OK for opfx_solve, opfx_step and
 * opfx_reset alike; the batch buffers may then be NULL. */
enum { OPFX_INIT_FLAT = 0, OPFX_INIT_DC = 1 };
typedef struct opfx_solve_opts {
  uint32_t struct_size;      /* = sizeof(opfx_solve_opts)                    */
  int32_t reserved0;         /* (alignment) 0                                */
  double tol;                /* inf-norm tolerance on the mismatch, p.u. (pandapower tolerance_mva=1e-8) */
  int32_t max_iter;          /* pandapower max_iteration 'auto' -> 10        */
  int32_t enforce_q_lims;    /* opf_env.py:697: PV->PQ switching on Q limits.  0: off.  1: on; a generator whose reactive
                              * range is a single point (eco_dispatch.py:86-88: min_q = max_q = 0) STARTS pinned at it — where
                              * the first pass always puts it — which saves one Newton solve.  2: on, pypower's own path: every
                              * generator starts as a PV bus, violated limits are pinned after the first converged solve
                              * (same result; `iterations` then count both solves, as pandapower's do) */
  int32_t init;              /* start of the base-case Newton iteration: OPFX_INIT_FLAT (0, default): |V| = 1 (set-points at
                              * PV / REF buses), angles = the slack angle carried through the transformer phase shifts;
                              * OPFX_INIT_DC (1): angles from a DC power flow B' theta = P first (pandapower init='dc',
                              * its 'auto' choice whenever voltage angles are calculated, i.e. for grids fed above 70 kV):
                              * one linear solve through the same block-LU schedule.  Same fixed point, other iteration
                              * counts.  A solve with branches OUT OF SERVICE — the `outage` array, an N-1 contingency
                              * (with contingency_start = 1), an open line / transformer switch — starts from the DC power
                              * flow of the grid without them, as pandapower's does; a solve with any other per-instance
                              * modifier (tap position, branch open at one end, shunt step), with several branches out
                              * at once or with a de-energised island starts flat.  A contingency that starts from the
                              * base case's solution (contingency_start = 0) starts THERE: no DC pass on a warm start. */
  int32_t contingency_start; /* opfx_step, N-1 loop: 0 = every contingency solve starts from the base-case
                              * solution (default; same fixed point, one iteration fewer), 1 = from the flat
                              * start, as the reference does by calling pandapower anew
                              * (security_constrained.py:53): identical iteration counts and identical
                              * behaviour next to voltage collapse */
  double jacobian_reuse_tol; /* 0 (default): full Newton, a Jacobian factorisation in every iteration — what pandapower does.
                              * theta > 0 (Shamanskii / chord steps): once the mismatch norm of an iteration is below theta
                              * AND the step before it cut the norm at least tenfold (Newton is past its slow start),
                              * the LATER iterations of that solve keep its factorisation — mismatch, forward and back
                              * substitution only, no Jacobian, no block LU — for as long as each such step cuts the norm
                              * at least tenfold (else the next iteration factorises again).  Same fixed point and same
                              * tolerance test; the iterates differ from full Newton's after the switch, so iteration
                              * counts and the last digits of V may differ from pandapower's.  theta is in p.u. like tol
                              * (base_mva = 1: MW; flat-start mismatches of the benchmark grids are 10 ... 70, theta = 0.1
                              * is what profiles/r04_ab_chord.txt recommends).  Ignored (full Newton) together with
                              * init = OPFX_INIT_DC, on the first-generation fallback kernel and in the memory-resident form */
} opfx_solve_opts;

int opfx_solve(opfx_ctx* ctx, int64_t B, const double* p_inj, const double* q_inj,
               const double* qg_min, const double* qg_max,
               const int32_t* outage, const opfx_solve_opts* opts,
               double* vm, double* va, double* loading, double* s_ref, double* q_gen,
               uint8_t* converged, int32_t* iterations, double* max_mismatch,
               double* min_pivot, int32_t* min_pivot_bus, void* stream);

/* ---- environment evaluation ------------------------------------------------
 * Per-instance state lives in one row-major column store x[B,nx] owned by the
 * caller: every per-instance table column of the reference net that the hot
 * path reads or writes (load/sgen/storage/gen p_mw,q_mvar as TABLE values,
 * i.e. before `scaling`; per-reset limit columns such as max_q_mvar; sampled
 * prices) has a slot in x.  Indices below are slots in x unless stated. */

enum { OPFX_SRC_X = 0, OPFX_SRC_RESULT = 1 };
/* result bank layout (per instance), index space of OPFX_SRC_RESULT sources
 * and of constraint elements: [vm nb | va_degree nb | loading nbr |
 * p_ext nref | q_ext nref | q_gen nb (Mvar generated at PV buses, 0 else)] */

enum { OPFX_COST_UNIT = 0, OPFX_COST_EXT_GRID = 1, OPFX_COST_GEN = 2 };
enum { OPFX_REWARD_SUMMATION = 0, OPFX_REWARD_REPLACEMENT = 1,
       OPFX_REWARD_PARAMETERIZED = 2, OPFX_REWARD_ONLY_OBJECTIVE = 3 };

/* kinds of actuator columns (opfx_env_desc.act_kind) */
enum { OPFX_ACT_CONTINUOUS = 0, OPFX_ACT_INTEGER = 1 /* np.round */, OPFX_ACT_BOOLEAN = 2 /* np.round(..).astype(bool) */ };

enum { OPFX_XRES_P = 0, OPFX_XRES_S = 1, OPFX_XRES_MAX3 = 2, OPFX_XRES_AFFINE = 3 };

typedef struct opfx_env_desc {
  uint32_t struct_size;      /* = sizeof(opfx_env_desc)                      */
  int32_t nx;                /* columns of the per-instance store           */
  /* bus injections: P_i = sum coef*x[slot] over p-list of bus i (p.u.) */
  const int32_t* pinj_ptr;   /* [nb+1] */
  const int32_t* pinj_slot;  const double* pinj_coef;
  const int32_t* qinj_ptr;   /* [nb+1] */
  const int32_t* qinj_slot;  const double* qinj_coef;
  /* reactive capability of PV buses for enforce_q_lims: Mvar → p.u. bounds  */
  const double* qg_min;      /* [nb] p.u. (NaN/inf = unlimited) or NULL      */
  const double* qg_max;      /* [nb] */
  /* actions (opf_env.py:421-491) */
  int32_t na;
  const int32_t* act_slot;   /* [na] column written: x = setpoint/scaling    */
  const double* act_scaling; /* [na] */
  const int32_t* act_lo_slot;/* [na] range source column, or -1 → act_lo_const */
  const int32_t* act_hi_slot;
  const double* act_lo_const;/* [na] */
  const double* act_hi_const;
  const int32_t* clamp_lo_slot; /* [na] -2 none, -1 const, >=0 column (autoscale off / diff mode, :464-470) */
  const int32_t* clamp_hi_slot;
  const double* clamp_lo_const;
  const double* clamp_hi_const;
  int32_t clamp_enabled;          /* bit 0: clamp in step(), bit 1: clamp in reset() (= autoscale off) */
  double diff_action_step_size;   /* 0 → absolute set-points (:451-461)     */
  double clipped_action_penalty;  /* :403-404 */
  /* costs (objective.py:34-77); coefficient table = [npoly*6 | npwl*nseg*3] */
  int32_t npoly, npwl, nseg;
  const int32_t* cost_kind;  /* [npoly+npwl] OPFX_COST_*                     */
  const int32_t* cost_pidx;  /* UNIT: slot of p_mw; EXT_GRID: ref ordinal; GEN: bus */
  const int32_t* cost_qidx;  /* UNIT: slot of q_mvar (or -1)                 */
  const double* cost_scale;  /* UNIT: scaling; others 1                      */
  const int32_t* pwl_is_q;   /* [npwl] */
  const double* cost_coef;   /* [npoly*6 + npwl*nseg*3] cp0,cp1,cp2,cq0,cq1,cq2 | lo,hi,price */
  int32_t nprice;            /* per-instance coefficient overrides           */
  const int32_t* price_slot; /* [nprice] column of x holding the price       */
  const int32_t* price_coef; /* [nprice] index into cost_coef it replaces    */
  /* constraints (constraints.py:70-128): nc groups over result-bank elements */
  int32_t nc;
  const int32_t* con_ptr;    /* [nc+1] → elements                            */
  const int32_t* con_src;    /* [ncel] result-bank index                     */
  const double* con_min;     /* [ncel] NaN = no lower bound                  */
  const double* con_max;     /* [ncel] NaN = no upper bound                  */
  const double* con_autoscale;   /* [nc] multiplier (20, 1/30, … or 1)       */
  const double* con_penalty_factor;
  const double* con_penalty_power;
  const double* con_count_penalty;
  const int32_t* con_worst_case; /* [nc] only_worst_case_violations          */
  /* reward (reward.py:61-98, 219-320) */
  int32_t reward_kind;
  double penalty_weight;     /* NaN → None (plain sum, reward.py:79-80)      */
  double clip_lo, clip_hi;   /* NaN → no clipping                            */
  double objective_factor, objective_bias, penalty_factor, penalty_bias;
  double valid_reward, invalid_penalty, invalid_objective_share;
  int32_t diff_objective;    /* opf_env.py:497-498                           */
  /* observation (opf_env.py:532-549): obs[k] = bank(obs_kind[k])[obs_idx[k]] */
  int32_t nobs;
  const int32_t* obs_kind;   /* OPFX_SRC_X / OPFX_SRC_RESULT                 */
  const int32_t* obs_idx;
  int32_t steps_per_episode; /* opf_env.py:406-414                           */
  /* N-1 security constraints (security_constrained.py:37-68): after the base
   * case, every listed branch is taken out in turn, the case is re-solved and
   * its violations are accumulated (valids &=, violations +=, penalties +=). */
  int32_t n_cont;
  const int32_t* cont_branch;     /* [n_cont] case branch index              */
  double not_converged_penalty;   /* security_constrained.py:27,63-64        */
  /* discrete actuators (opf_env.py:476-481): OPFX_ACT_* per action, NULL = all continuous */
  const int32_t* act_kind;        /* [na] */
  /* Branch state columns — actuators that change Ybus values per instance
   * (examples/network_reconfiguration.py:34-35, mixed_continuous_discrete.py:40-41:
   * ('switch','closed'), ('trafo','tap_pos'), ('line'|'trafo','in_service')).  Row m names a
   * case branch whose four stamps follow an integer state column of the store:
   * stamps = bmod_y[bmod_ptr[m] + clip(round(x[bmod_slot[m]]) - bmod_lo[m], 0, bmod_n[m]-1)];
   * an all-zero table row takes the branch out of service.  The plan is compiled with every
   * such branch present; at most one state column per branch.
   * A row with bmod_branch[m] = -1 - bus is a BUS SHUNT in steps (an ('shunt','step') actuator,
   * opf_env.py:476-481): its table rows hold (0, 0, 0, 0, 0, 0, dG, dB), the DIFFERENCE of the bus's shunt
   * admittance at that step to the compiled case, p.u. */
  int32_t n_bmod;
  const int32_t* bmod_branch;     /* [n_bmod] case branch index, or -1 - bus */
  const int32_t* bmod_slot;       /* [n_bmod] column of the store            */
  const int32_t* bmod_lo;         /* [n_bmod] state value of table row 0     */
  const int32_t* bmod_n;          /* [n_bmod] table rows                     */
  const int32_t* bmod_ptr;        /* [n_bmod] first row in bmod_y            */
  const double* bmod_y;           /* [rows][8] ff, ft, tf, tt as (g, b), p.u. */
  /* per-instance voltage set-points (a sampled ext_grid.vm_pu,
   * examples/mixed_continuous_discrete.py:102-104): REF/PV bus i holds |V| = x[vset_slot[i]]
   * instead of the compiled magnitude; -1 = compiled value; NULL = none */
  const int32_t* vset_slot;       /* [nb] */
  /* objective terms on the result bank — the device form of the `objective_function` seam
   * (opf_env.py:52,80-84) for objectives like (vm_pu - 1)^2 per bus
   * (mixed_continuous_discrete.py:17-19): the objective vector gets, after the cost rows,
   * qterm_weight[k] * (result[qterm_idx[k]] - qterm_target[k])^2 */
  int32_t n_qterm;
  const int32_t* qterm_idx;       /* [n_qterm] result-bank index             */
  const double* qterm_target;     /* [n_qterm] */
  const double* qterm_weight;     /* [n_qterm] */
  /* derived rows appended to the result bank (index nres_base + k, nres_base = 3nb+nbr+2nref):
   * the res_<unit> echoes pandapower writes for controllable units and what custom constraints
   * compute from them (examples/custom_constraint.py:9-11):
   *   OPFX_XRES_P: x[xres_p[k]] * xres_scale[k]        (res_<unit>.p_mw / q_mvar = set-point * scaling)
   *   OPFX_XRES_S: sqrt(P^2 + Q^2) of x[xres_p[k]], x[xres_q[k]] (apparent power)
   *   OPFX_XRES_MAX3: NaN-propagating max of the RESULT-BANK entries xres_p[k], xres_q[k], xres_r[k]:
   *       res_trafo3w.loading_percent = the worst of the three windings of its star equivalent
   *   OPFX_XRES_AFFINE: xres_offset[k] + xres_scale[k] * RESULT-BANK entry xres_p[k] (an entry of the solver's part,
   *       index < nres_base); 0 when xres_q[k] >= 0 names a bus that is de-energised in this instance.  The share of ONE
   *       generator (or ext_grid) in the reactive power generated at its bus, which the solver reports per BUS (q_gen /
   *       q_ext): pypower's `pfsoln` splits the bus total among the generators of a bus in proportion to their reactive
   *       ranges, Q_g = Qmin_g + (Q_bus - sum Qmin) / (sum Qmax - sum Qmin + eps) * (Qmax_g - Qmin_g) — affine in Q_bus
   *       with constants the caller computes once (opfgym_amd/case.py generator_dispatch); likewise the active power of an
   *       ext_grid that is not the first generator of its REF bus (0 * p_ext).  Read by res_gen.q_mvar / res_ext_grid
   *       observations, constraints and cost rows (objective.py:48-54). */
  int32_t n_xres;
  const int32_t* xres_kind;       /* [n_xres] OPFX_XRES_*                    */
  const int32_t* xres_p;          /* [n_xres] column of the store            */
  const int32_t* xres_q;          /* [n_xres] column of the store or -1      */
  const double* xres_scale;       /* [n_xres] */
  const int32_t* xres_r;          /* [n_xres] third result index of OPFX_XRES_MAX3 rows (else unused); NULL = none */
  /* Bus of the unit behind every OPFX_COST_UNIT row, -1 for the other kinds; NULL = none.  pandapower reports zero
   * power for units on a de-energised bus (results_bus.py: set-point x `_is_elements`), so the cost rows of units on
   * an island that a switch state or an outage has cut off vanish from the objective (objective.py:34-54). */
  const int32_t* cost_bus;        /* [npoly+npwl] */
  /* ---- appended in 0.3.1 (a caller compiled against 0.3.0 leaves them out: NULL) ---- */
  const double* xres_offset;      /* [n_xres] constant term of OPFX_XRES_AFFINE rows (else unused); NULL = zeros */
  /* Result-bank entries (derived rows allowed) that replace what an OPFX_COST_EXT_GRID / OPFX_COST_GEN row reads from the
   * solver's per-BUS values: cost_pres[r] for the active power of an ext_grid row, cost_qres[r] for the reactive power of
   * an ext_grid or generator row; -1 = the bus value (p_ext / q_ext of the REF bus, q_gen of the generator's bus).  For
   * units that share their bus with other generators (see OPFX_XRES_AFFINE).  NULL = none. */
  const int32_t* cost_pres;       /* [npoly+npwl] */
  const int32_t* cost_qres;       /* [npoly+npwl] */
} opfx_env_desc;

int opfx_env_create(opfx_ctx* ctx, const opfx_env_desc* d, opfx_env** out);
void opfx_env_destroy(opfx_env* env);

typedef struct opfx_step_io {
  uint32_t struct_size;      /* = sizeof(opfx_step_io)                       */
  uint32_t reserved0;        /* (alignment) 0                                */
  /* inputs */
  double* x;                 /* [B,nx] in/out: action set-points are written back */
  const double* action;      /* [B,na] in [0,1] (clipped inside, opf_env.py:429)  */
  const double* initial_obj; /* [B] or NULL (diff_objective)                 */
  const int32_t* step_in_episode; /* [B] or NULL (→ 1)                       */
  const int32_t* outage;     /* [B] or NULL: branch out of service in the base case, -1 none */
  /* outputs, any may be NULL */
  double* obs;               /* [B,nobs]                                     */
  double* reward;            /* [B]                                          */
  uint8_t* terminated;       /* [B]                                          */
  uint8_t* truncated;        /* [B]                                          */
  uint8_t* valids;           /* [B,nc]                                       */
  double* violations;        /* [B,nc]                                       */
  double* penalties;         /* [B,nc]  info['unscaled_penalties']           */
  double* cost;              /* [B]     info['cost']                         */
  double* objective;         /* [B]     sum of the (negated) cost vector     */
  double* results;           /* [B,nres] result bank                         */
  double* mean_correction;   /* [B]                                          */
  uint8_t* converged;        /* [B]                                          */
  int32_t* iterations;       /* [B]  NR iterations of the base case          */
  double* max_mismatch;      /* [B]                                          */
  int32_t* total_iterations; /* [B]  NR iterations summed over the base case and every contingency solve */
  double* min_pivot;         /* [B]  smallest relative 2x2 pivot of all solves of the step, see opfx_solve */
  int32_t* min_pivot_bus;    /* [B]  the bus it belongs to (-1: none), see opfx_solve */
} opfx_step_io;

/* One env.step() for B instances: apply actions → injections → NR → results →
 * objective → violations → reward → observation, one kernel launch.
 * mode: 0 = full step; 1 = evaluate the current set-points without applying
 * `action` (`apply_action=False`, opf_env.py:197,386); 2 = apply `action` and
 * write the table observation only, no power flow (reset of an environment
 * whose observation needs no results, opf_env.py:207,218); 3 = table observation of the
 * current x only (no action, no power flow; multi_stage.py:56); 4 = full step with the action
 * applied the way reset applies its initial action (opf_env.py:207-216: absolute set-points
 * even when the environment steps incrementally, clamping only without autoscaling).
 * Modes 2 and 4 are the two forms of reset; 5 = as 4 (absolute set-points) but a complete evaluation,
 * contingencies included: what `estimate_reward_distribution` does per sample (reward.py:186-193:
 * `_apply_actions(action)` without a step size, power flow, objective, `calculate_violations`). */
int opfx_step(opfx_env* env, int64_t B, const opfx_step_io* io,
              const opfx_solve_opts* opts, int32_t mode, void* stream);

/* ---- reset: SimBench-state sampling on the device (opf_env.py:317-372) -----
 * Profile tables are uploaded once in SimBench's factored form: column j of a
 * table is rel[:, typ[j]] * peak[j]; per-column min/max are precomputed once
 * (defect D10 of the reference recomputes them on every reset). */
typedef struct opfx_profile_desc {
  uint32_t struct_size;      /* = sizeof(opfx_profile_desc): also the stride of opfx_reset_desc.tables */
  int32_t n_steps, n_types, n_cols;
  const double* rel;         /* [n_steps, n_types] row-major                 */
  const int32_t* typ;        /* [n_cols] */
  const double* peak;        /* [n_cols] */
  const int32_t* slot;       /* [n_cols] destination column in x             */
  const double* col_min;     /* [n_cols] clip range (opf_env.py:364-369)     */
  const double* col_max;
} opfx_profile_desc;

/* post-sampling column programme: the envs' `_sampling` tails as a short list
 * of element-wise VECTOR ops executed in order on the instance's row of x; op k
 * works on columns dst[k]+j, a[k]+j for j < n[k]; c0/c1/c2 are per-element
 * constant vectors given as offsets into `consts` (-1 = not used):
 *   OPFX_OP_SET_CONST   x[dst] = c0
 *   OPFX_OP_AFFINE      x[dst] = x[a]*c0 + c1            (max_p = p*scaling + eps)
 *   OPFX_OP_SQRT_DIFF   x[dst] = sqrt(c0^2 - x[a]^2)     (q_max = sqrt(S^2 - P^2))
 *   OPFX_OP_NEG         x[dst] = -x[a]
 *   OPFX_OP_UNIFORM     x[dst] = (c0 + u*(c1-c0)) / c2,  u = uniform[b, a+j]
 *                       (opf_env.py:278-284; here `a` indexes the draw vector)
 *   OPFX_OP_NORMAL      x[dst] = c0 + c1*z,  z = normal[b, a+j]   (opf_env.py:311-312)
 *   OPFX_OP_CLIP        x[dst] = min(max(x[a], c0), c1)           (opf_env.py:313-314)
 *   OPFX_OP_DIV         x[dst] = x[a] / c0                        (load_shedding.py:137)
 *   OPFX_OP_NORMINV     x[dst] = c0 + c1*Phi^-1(x[a])   (truncated normal by inverse CDF, opf_env.py:306-309:
 *                       x[a] holds a probability drawn uniformly between Phi(lower) and Phi(upper))
 *   OPFX_OP_TRUNCNORM   x[dst] = ppf of the standard normal truncated to [c0, c1] at the probability x[a] — what
 *                       scipy.stats.truncnorm.ppf(x[a], c0, c1) returns, evaluated in log space so that bounds far
 *                       out in one tail (the reference passes raw MW values as standardised bounds) stay finite
 */
enum { OPFX_OP_SET_CONST = 0, OPFX_OP_AFFINE = 1, OPFX_OP_SQRT_DIFF = 2,
       OPFX_OP_NEG = 3, OPFX_OP_UNIFORM = 4, OPFX_OP_NORMAL = 5, OPFX_OP_CLIP = 6,
       OPFX_OP_DIV = 7, OPFX_OP_NORMINV = 8, OPFX_OP_TRUNCNORM = 9 };
typedef struct opfx_reset_desc {
  uint32_t struct_size;      /* = sizeof(opfx_reset_desc)                    */
  int32_t n_tables;
  const opfx_profile_desc* tables;
  int32_t n_ops;
  const int32_t* op_code; const int32_t* op_dst; const int32_t* op_a; const int32_t* op_n;
  const int32_t* op_c0; const int32_t* op_c1; const int32_t* op_c2;   /* offsets into consts */
  int32_t n_consts;
  const double* consts;
  int32_t n_uniform;         /* uniform draws consumed per instance          */
  int32_t init_off;          /* offset into consts of an nx-long row template copied
                                into x before the tables are applied, or -1  */
  int32_t n_normal;          /* standard-normal draws consumed per instance  */
  /* `train_data='mixed'` (opf_env.py:242-251): every reset draws ONE of the data sources per
   * instance.  op_mode[k] is the set of sources (bit 0 SimBench profiles, bit 1 uniform, bit 2
   * normal) under which op k runs; NULL = every op always.  The profile tables apply under source 0. */
  const int32_t* op_mode;    /* [n_ops] or NULL */
} opfx_reset_desc;

int opfx_env_set_reset(opfx_env* env, const opfx_reset_desc* d);

/* Random numbers are INPUTS (the caller draws them, on the device or from a seeded host
 * generator, so that runs are reproducible from the caller's RNG):
 *   step_idx [B] int32   SimBench time step per instance;
 *   noise [B,n_noise] or NULL, one value per profile column in table order:
 *       normal_noise_factor == 0: multiplicative factors (opf_env.py:354-356);
 *       normal_noise_factor  > 0: standard-normal draws z, value += |value|*factor*z (:359-360);
 *   interp [B,n_tables] or NULL: r in [0,1) per table, value = row(step)*r + row(step+1)*(1-r)
 *       for step < n_steps-1 (`interpolate_steps`, opf_env.py:345-349);
 *   uniform [B,n_uniform], normal [B,n_normal]: draws consumed by the OPFX_OP_UNIFORM /
 *       OPFX_OP_NORMAL ops in order; NULL: the kernel makes them itself — counter-based, value j of
 *       instance b is a function of (rng_seed, b, j) only — which saves the caller a launch per reset.
 * Fills x [B,nx]. */
typedef struct opfx_reset_io {
  uint32_t struct_size;      /* = sizeof(opfx_reset_io)                      */
  uint32_t reserved0;        /* (alignment) 0                                */
  const int32_t* step_idx;
  const double* noise;
  const double* interp;
  const double* uniform;
  const double* normal;
  double normal_noise_factor;
  double* x;
  const int32_t* mode;       /* [B] data source per instance (0, 1, 2) or NULL = no source selection */
  /* optional, the reset of an environment whose observation needs no power flow in ONE launch
   * (opf_env.py:201-207,217-218): with `obs` given the kernel also applies `action` (the initial
   * action, absolute set-points; NULL = keep the sampled values) and writes the table part of the
   * observation (result entries NaN) — what opfx_step mode 2 / 3 do in a second launch */
  const double* action;      /* [B,na] or NULL */
  double* obs;               /* [B,nobs] or NULL */
  /* 0: every row starts from the compiled table template (independent episodes); 1: from the
   * instance's current row in x, as the reference's single net does between episodes and between
   * the stages of a multi-stage episode (multi_stage.py:49-56: only the sampled columns change, the
   * set-points of the last action stay) */
  int32_t keep_state;
  /* optional: draw the time step of every instance inside the kernel instead of reading step_idx —
   * a uniformly random entry of `step_pool` (the train / validation / test steps, opf_env.py:327-333) from a
   * counter-based generator keyed by (rng_seed, instance), written to `step_out`.  Saves the two launches a
   * host-side gather needs in front of every reset. */
  const int32_t* step_pool;  /* [n_step_pool] or NULL = use step_idx */
  int32_t n_step_pool;
  uint64_t rng_seed;         /* change it for every reset */
  int32_t* step_out;         /* [B] the steps drawn (may be NULL) */
} opfx_reset_io;

int opfx_reset(opfx_env* env, int64_t B, const opfx_reset_io* io, void* stream);

/* Timing helper for bench.py: runs `reps` back-to-back opfx_step launches on
 * `stream` between two hipEvents recorded on that stream and returns the
 * elapsed milliseconds (device time of the timed region). */
int opfx_time_steps(opfx_env* env, int64_t B, const opfx_step_io* io,
                    const opfx_solve_opts* opts, int32_t reps, void* stream,
                    float* elapsed_ms);

/* Launch configuration of the environment's fused step kernel (report/diagnostics; bench.py prints it):
 * wavefronts that share one instance (1, 2 or 4), LDS bytes per instance, and instances resident per CU
 * (0 until the first opfx_step has queried the occupancy).  Any out pointer may be NULL. */
int opfx_env_get_info(const opfx_env* env, int32_t* waves_per_instance, int64_t* lds_bytes_per_instance,
                      int32_t* instances_per_cu);

/* Block storage the environment's kernels run with (report/diagnostics): LU blocks in all, and how many of them
 * are stored with four values (the rest keep the two values (a, b) of [[a, b], [-b, a]], see opfx_plan_info.n_full;
 * equal to n_blk when the two-value form is not used: it is chosen only where the LDS it saves lets a CU hold more
 * instances).  Any out pointer may be NULL. */
int opfx_env_get_storage(const opfx_env* env, int32_t* n_blk, int32_t* n_four_value);

/* Specialisation of the plain step kernel this environment's launches use (report/diagnostics; the last template argument
 * of k_step in a profile): bit 0 = the plan has no PV bus, bit 1 = the environment has no per-instance branch modifier of
 * any kind (no switch / tap / shunt-step column, no N-1 contingency, no per-instance |V| set-point).  What the environment
 * fixes for the whole batch is compiled out of the kernel instead of being tested per instance; a call that brings an
 * `outage` array runs without bit 1; a single-wave launch on a grid of more than 256 buses runs without bit 1 as well (the
 * no-modifier single-wave kernels keep |V| and the angle of four bus rounds in registers).  (The full-Newton kernels,
 * plain and with the DC start; the chord-step and memory-resident kernels and the first-generation fallback are not
 * specialised.) */
int opfx_env_get_spec(const opfx_env* env, int32_t* spec);
/* (0.3.1) How much of an instance's row x[b, 0 .. nx) opfx_step reads: the columns [0, *columns_read) — up to the last one a
 * descriptor of the environment names (injections, actuators and their limits, observations, prices, state columns).
 * opfx_reset writes the whole row.  A caller that keeps columns per instance which the step never looks at (intermediates of
 * its reset programme, limit columns of units that are no actuators) saves their HBM traffic by laying them out LAST
 * (opfgym_amd.BatchedOpfEnv does). */
int opfx_env_get_row_io(const opfx_env* env, int32_t* columns_read);
/* (0.3.1) Allocate the context's per-workgroup scratch rows for launches of up to B instances NOW.  Without it the first
 * opfx_step / opfx_reset (and any later one with a larger launch grid) allocates them itself: a synchronising hipMalloc on
 * the hot path, which must not happen inside a stream capture and belongs outside a timed region.  (The block-value rows of
 * the memory-resident kernels, grids past the LDS, are still allocated by their first launch.) */
int opfx_env_prepare(opfx_env* env, int64_t B);

#ifdef __cplusplus
}
#endif
#endif /* OPFX_H */
