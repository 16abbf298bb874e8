// libopfx device side (header shared by the kernel translation units k_*.hip and the host side opfx.hip).
//
//
// Execution model: ONE WAVEFRONT (64 lanes) PER GRID INSTANCE, instance state
// resident in LDS, persistent waves striding over the batch.
//   * all structure (Ybus CSR, Jacobian block pattern, level-scheduled block-LU
//     elimination programme) is shared by the whole batch, read-only, and
//     stays in L1/L2; per-instance HBM traffic is inputs + outputs only;
//   * inside an instance the 64 lanes parallelise over buses (mismatch,
//     Jacobian rows, voltage update, results), over the independent work items
//     of an elimination level (numeric 2x2-block LU + forward substitution),
//     over the pivots of a level (back substitution), and reduce the mismatch
//     inf-norm / constraint sums with cross-lane shuffles;
//   * no inter-workgroup communication and no global atomics; update terms that share a target meet in LDS
//     atomics (ds_add_f64): one wavefront issues them in a fixed order, so the single-wave kernels are
//     bit-reproducible; the wave teams (large grids) are reproducible to rounding only.
//
// Replaces: pypower `newtonpf` (P4), the q-limit outer loop
// (P5), `pfsoln`/result extraction (P6) and, in MODE_ENV, OpfEnv._apply_actions
// (opf_env.py:421-491), get_pandapower_costs (objective.py:6-87),
// Constraint.get_violation_metrics (constraints.py:70-128),
// RewardFunction.__call__ (reward.py:61-98) and OpfEnv._get_obs
// (opf_env.py:532-549) — all inside one kernel launch per env.step().
#ifndef OPFX_DEV_H
#define OPFX_DEV_H
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "opfx.h"
#include "opfx_debug.h"
#include "plan.h"

#define HIP_TRY(expr)                                                         \
  do {                                                                        \
    hipError_t err__ = (expr);                                                \
    if (err__ != hipSuccess) {                                                \
      opfx_set_error(std::string(#expr) + ": " + hipGetErrorString(err__));   \
      return OPFX_ERR_HIP;                                                    \
    }                                                                         \
  } while (0)

namespace {

constexpr int WAVE = 64;
// SPEC: what the environment fixes for the WHOLE batch, as a template parameter of the step kernels, so that the tests for
// it disappear at compile time instead of being skipped at run time (round 5; the single-wave kernel is instruction-issue
// bound, only fewer instructions make it faster: BASELINE config 2 0.242 -> 0.230 ms with both bits):
//   SPEC_NO_PV   no PV bus in the plan (voltage_control.py:102 asserts a grid without generators): the bus-type tests for PV
//                rows, the q-limit loop and its bus types go;
//   SPEC_NO_MOD  no per-instance branch modifier of any kind — no switch / tap / shunt-step column, no outage array, no N-1
//                contingency, no per-instance |V| set-point: the modifier plumbing, islands and de-energised buses go.
// opfx_step picks the instantiation per launch (do_step); -DOPFX_FORCE_SPEC=n (probe builds) forces n's bits into every one.
constexpr int SPEC_NO_PV = 1, SPEC_NO_MOD = 2;
#ifndef OPFX_FORCE_SPEC
#define OPFX_FORCE_SPEC 0
#endif
#ifndef OPFX_SPEC_MASK                      // (probe builds: -DOPFX_SPEC_MASK=0|1|2 keeps only these bits of what a launch would pick)
#define OPFX_SPEC_MASK 3
#endif
// Developer probe (-DOPFX_DUP=<bit mask>, scripts/ab_dup.sh): phase k of k_step's prologue / epilogue runs TWICE — every one
// of them is idempotent as far as the solve and the timing go — so that (time with the bit) - (time without) is what the
// phase costs in the real mix of wavefronts, without the distortion of in-kernel cycle stamps (each stamp is a memory round
// trip of its own and serialises what the hardware overlaps).  0 in the product: the loops fold away.
#ifndef OPFX_DUP
#define OPFX_DUP 0
#endif
#define OPFX_REP(k) for (int rep__ = 0; rep__ < 1 + ((OPFX_DUP >> (k)) & 1); ++rep__, dup_fence())
__device__ __forceinline__ void dup_fence() { asm volatile("" ::: "memory"); }
constexpr int MODE_SOLVE = 0;
constexpr int MODE_ENV = 1;

// ---------------------------------------------------------------------------
// device-resident plan
// ---------------------------------------------------------------------------
// s_getreg operands (size - 1) << 11 | offset << 6 | register: all 32 bits of HW_ID (wave slot [3:0], SIMD [5:4], CU [11:8],
// shader array [12], shader engine [15:13]) and of XCC_ID
constexpr int GETREG_HW_ID = (31 << 11) | 4, GETREG_XCC_ID = (31 << 11) | 20;

struct DevPlan {
  int nb, nbr, nref, nblk, nlev, nfill, npv;
  int nfull;                 // blocks [0, nfull) are stored with four values (per launch: choose_block_storage)
  int n_shared;              // > 0: some block ids are LDS slots shared over time (plan.cpp share_slots): team items carry zero stores
  int fill_lo;               // fill blocks: ids [fill_lo, fill_lo + nfill)
  double base_mva;
  const int *bus_type, *y_ptr, *y_col, *y_blk, *diag_blk, *fill_blk;
  const int *lev_tptr, *tgt_blk, *tgt_sptr, *src_ik, *src_kk, *src_kj;
  const int *lev_pptr, *piv_bus, *piv_uptr, *u_blk, *u_col;
  const int *br_f, *br_t, *br_pos, *br_island, *isl_ptr, *isl_bus, *ref_bus, *ref_ord;
  const double *vm_set, *va_set, *vr0, *vi0, *y_g, *y_b, *br_y, *br_kf, *br_kt;
  // lane programme (plan.h)
  int ra, rh, rb, rc;
  unsigned long long* stamps; // developer probe (opfx_debug_opts.stamps): per-phase cycle sums of workgroup 0
  const unsigned *lp_bc, *lp_apk, *lp_hpk;
  const unsigned* lp_bcc;                // chord stream (plan.h): [rf rounds forward substitution | rc rounds back substitution]
  int rf;                                // its forward part, padded rounds
  const int* lp_hrows;
  int n_hrows;
  const unsigned *lp_team2, *lp_team4;   // per-wave streams of the cooperative kernels (plan.h)
  int tail_m, tail_n;                    // dense tail of the elimination, solved in registers (plan.cpp; 0: none); entries of tail_ids
  const unsigned* tail_bus;              // [32] bus | diagonal block << 16 of tail pivot e
  const unsigned short* tail_ids;        // [tail_m][M] id of U-block (row e, column s) at [e * M + s], M = tail_m rounded up to 8
  int team_rounds2, team_rounds4, team_kb2, team_kb4;
  const unsigned *lp_teamc2, *lp_teamc4; // chord streams of the teams
  int team_roundsc2, team_roundsc4, team_kbc2, team_kbc4;
  const double *lp_dc, *lp_hdc;          // DC start (plan.h): B' on the Ybus pattern + constant right-hand side; nullptr: none
  const double *br_bdc, *br_pfinj;       // [nbr] DC susceptance / phase-shift injection of every branch (DC start of a solve with branches out)
  double* blk_mem;           // memory-resident kernels: [resident workgroups][blk_mem_stride] LU block values
  long long blk_mem_stride;
  double* warm;              // [resident workgroups][2*nb] base-case voltages, start of the N-1 solves
  double* pq;                // [resident workgroups][2*nbe] scheduled P/Q of the workgroup's instance (see carve)
  int* queue;                // work queue of the step kernel: instances handed out beyond the first one per workgroup
  // (members added in round 6 go to the END: the struct is a kernel argument, and the offsets of the members above decide how the
  //  scalar loads of the hot kernels group — an int put in front of DevEnv cost 0.8 % of a config-2 step, profiles/r06_ab_prev_tree.txt)
  double* theta0;            // the `warm` rows used for the base case's DC angles instead (launches whose contingencies start from a
                             // DC power flow of their own, derived from the base case's by a rank-1 update: cont_dc_*), or nullptr
};

struct DevEnv {
  int nx, na, npoly, npwl, nseg, nc, nobs, nres, ncost, ncost_pre;
  int nblk_d;                // doubles reserved for [LU blocks | result bank | staged table row beyond rhs]
  int reward_kind, diff_objective, steps_per_episode, clamp_enabled;
  int n_cont, n_inj, n_oseg, need_angle, ncel;
  int n_oseg_res;            // observation segments read from the result bank (0: the epilogue writes no observation)
  int max_mod;               // modifier records reserved per instance (env modifiers + outage + contingency)
  int n_bmod;                // branch state columns (taps, switches): see opfx_env_desc.bmod_*
  int nx_hot;                // columns [0, nx_hot) of the row are all the step kernel reads (opfx_env_create); the rest stays in HBM
                             // (the 24th int: the slot that used to be padding in front of the pointers — no other member moves)
  const int *act_kind, *bmod_branch, *bmod_src, *bmod_lo, *bmod_n, *bmod_ptr;
  const double* bmod_y;
  const int* vset_src;       // [nb] source of a per-instance |V| set-point (NOSRC: compiled value), or nullptr
  int n_qterm;               // quadratic objective terms on the result bank
  int n_xres, nres_base;     // derived result rows [nres_base, nres_base + n_xres)
  const int *xres_kind, *xres_p, *xres_q, *xres_r;
  const double* xres_scale;
  const int* qterm_idx;
  const double *qterm_target, *qterm_weight;
  double penalty_weight, clip_lo, clip_hi, objective_factor, objective_bias;
  double penalty_factor, penalty_bias, valid_reward, invalid_penalty;
  double invalid_objective_share, diff_step, clipped_action_penalty;
  double not_converged_penalty;
  const uint4* inj_pk;                             // flat list: {bus | isQ<<16, source, coefficient (2 words)}
  const double *qg_min, *qg_max;
  const int *oseg_kind, *oseg_src, *oseg_dst, *oseg_n;   // observation = list of contiguous copies
  const int *act_slot, *act_lo_slot, *act_hi_slot, *clamp_lo_slot, *clamp_hi_slot;
  const double *act_scaling, *act_lo_const, *act_hi_const, *clamp_lo_const, *clamp_hi_const;
  // cost rows in processing order: rows fed by table values/set-points first (ncost_pre), then
  // rows fed by the solve.  meta = kind | is_pwl<<4 | pwl_is_q<<5; sources: see src_val()
  const int *cost_meta, *cost_psrc, *cost_qsrc, *cost_cbase, *coef_xslot;
  const int* cost_bus;       // [ncost_pre] bus of the unit behind a pre-solve cost row (-1: none), or nullptr
  const double *cost_scale, *cost_coef;
  const int2* con_pk;                              // {result index, constraint}
  const int* con_worst;
  const double *con_min, *con_max, *con_autoscale, *con_pfac, *con_ppow, *con_cpen;
  const int *cont_branch;
  // DC start of the N-1 contingencies by a rank-1 update of the base case's DC power flow (opfx_env_create): per contingency
  // w = B'^-1 (e_f - e_t) [nb] and {b, Pfinj, 1 / (1 - b (w_f - w_t)), -}; nullptr: every contingency runs its own DC pass
  const double *cont_dc_w, *cont_dc_k;
  const double* xres_off;    // [n_xres] constant term of OPFX_XRES_AFFINE rows
  const int2* cost_res;      // [ncost] {P, Q} result-bank entries that replace the per-bus values a solve-fed cost row reads (-1: none), or nullptr
};

struct SolveIO {
  const double *p_inj, *q_inj, *qg_min, *qg_max;
  const int* outage;
  double *vm, *va, *loading, *s_ref, *q_gen, *max_mismatch;
  unsigned char* converged;
  int* iterations;
  double* min_pivot;
  int* min_pivot_bus;
  int queued;                // as StepIO::queued
};

struct StepIO {
  double* x;
  const double *action, *initial_obj;
  const int* step_in_episode;
  const int* outage;
  double *obs, *reward, *violations, *penalties, *cost, *objective, *results;
  double *mean_correction, *max_mismatch;
  unsigned char *terminated, *truncated, *valids, *converged;
  int* iterations;
  int* total_iterations;
  double* min_pivot;
  int* min_pivot_bus;
  int mode;
  int queued;                // instances beyond a workgroup's first come from the context's work queue (many per workgroup)
};

struct Opts {
  double tol;
  int max_iter;
  int enforce_q_lims;
  int contingency_start;     // opfx_solve_opts::contingency_start
  int init;                  // opfx_solve_opts::init
  double reuse_tol;          // opfx_solve_opts::jacobian_reuse_tol (kernels instantiated with CHORD)
};

// ---------------------------------------------------------------------------
// wave-level helpers (one wave == one workgroup == one instance)
// ---------------------------------------------------------------------------
__device__ __forceinline__ void wave_sync() {
  // LDS operations of one wave execute in issue order; what is needed is that
  // the compiler neither reorders across this point nor keeps LDS values in
  // registers.  (blockDim.x == 64, so this is also a full workgroup barrier.)
  __syncthreads();
}

// NaN-propagating max: a NaN mismatch (e.g. NaN set-points) must reach the convergence test
__device__ __forceinline__ double nan_max(double a, double b) {
  return (a != a) ? a : ((b != b) ? b : fmax(a, b));
}
// Same for NON-NEGATIVE operands (|x|, NaN with its sign cleared), branch-free: their bit
// patterns order like unsigned integers and every NaN pattern lies above +inf.
__device__ __forceinline__ double nn_max(double a, double b) {
  const unsigned long long ua = (unsigned long long)__double_as_longlong(a), ub = (unsigned long long)__double_as_longlong(b);
  return __longlong_as_double((long long)(ua > ub ? ua : ub));
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = nan_max(v, __shfl_xor(v, o, WAVE));
  return v;
}

// ---- DPP wave reductions (no LDS traffic): quad xor-1, xor-2, half-row mirror, row
// mirror give every lane its 16-lane row total; row_bcast:15 / row_bcast:31 fold the
// rows so that lane 63 holds the wave total, which is then read back as a scalar.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double old, double v) {
  const long long ov = __double_as_longlong(old), sv = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp((int)ov, (int)sv, CTRL, ROW_MASK, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(ov >> 32), (int)(sv >> 32), CTRL, ROW_MASK, 0xF, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double read_lane63(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)b, 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
  v += dpp_f64<0xB1, 0xF>(0.0, v);        // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E, 0xF>(0.0, v);        // quad_perm [2,3,0,1]
  v += dpp_f64<0x141, 0xF>(0.0, v);       // row_half_mirror
  v += dpp_f64<0x140, 0xF>(0.0, v);       // row_mirror
  v += dpp_f64<0x142, 0xA>(0.0, v);       // row_bcast:15 -> rows 1,3
  v += dpp_f64<0x143, 0xC>(0.0, v);       // row_bcast:31 -> rows 2,3
  return read_lane63(v);
}
__device__ __forceinline__ double wave_max_dpp(double v) {   // NaN-propagating; v >= 0 or NaN
  v = nn_max(v, dpp_f64<0xB1, 0xF>(v, v));
  v = nn_max(v, dpp_f64<0x4E, 0xF>(v, v));
  v = nn_max(v, dpp_f64<0x141, 0xF>(v, v));
  v = nn_max(v, dpp_f64<0x140, 0xF>(v, v));
  v = nn_max(v, dpp_f64<0x142, 0xA>(v, v));
  v = nn_max(v, dpp_f64<0x143, 0xC>(v, v));
  return read_lane63(v);
}
// min of NON-NEGATIVE operands by their bit patterns (a NaN pattern lies above +inf: ignored unless both are NaN)
__device__ __forceinline__ double nn_min(double a, double b) {
  const unsigned long long ua = (unsigned long long)__double_as_longlong(a), ub = (unsigned long long)__double_as_longlong(b);
  return __longlong_as_double((long long)(ua < ub ? ua : ub));
}
__device__ __forceinline__ double wave_min_dpp(double v) {   // v >= 0
  v = nn_min(v, dpp_f64<0xB1, 0xF>(v, v));
  v = nn_min(v, dpp_f64<0x4E, 0xF>(v, v));
  v = nn_min(v, dpp_f64<0x141, 0xF>(v, v));
  v = nn_min(v, dpp_f64<0x140, 0xF>(v, v));
  v = nn_min(v, dpp_f64<0x142, 0xA>(v, v));
  v = nn_min(v, dpp_f64<0x143, 0xC>(v, v));
  return read_lane63(v);
}
__device__ __forceinline__ int wave_any(int pred) { return __any(pred); }

// NW = wavefronts per instance.  Sections that only wavefront 0 executes use sec_sync
// (never a workgroup barrier); hand-overs between wavefront 0 and the team use blk_sync.
__device__ __forceinline__ void wave_fence() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
}
// With one wavefront per instance both are compiler-only fences (see wave_fence below): a
// __syncthreads() would drain every prefetched global load and every store (s_waitcnt vmcnt(0)).
template <int NW> __device__ __forceinline__ void blk_sync() { if (NW == 1) wave_fence(); else __syncthreads(); }
template <int NW> __device__ __forceinline__ void sec_sync() { wave_fence(); }

// Pointers read out of a descriptor that itself lives in memory are generic to the compiler
// (flat_load: counted on vmcnt AND lgkmcnt, serialising them with the LDS traffic); all of
// ours are hipMalloc'ed, so say so.
template <class T>
__device__ __forceinline__ const __attribute__((address_space(1))) T* as_global(const T* p) {
  return (const __attribute__((address_space(1))) T*)p;
}
template <class T>
__device__ __forceinline__ const __attribute__((address_space(4))) T* as_const(const T* p) {
  return (const __attribute__((address_space(4))) T*)p;
}
// element `idx` of an array whose base is wave-uniform: the byte offset as an unsigned 32-bit value, which is the form
// the hardware addresses as scalar base + vector offset (no 64-bit address arithmetic per lane); arrays < 4 GiB
template <class T>
__device__ __forceinline__ T ld_at(const T* base, unsigned idx) {
  typedef const __attribute__((address_space(1))) char* gbytes;
  return *reinterpret_cast<const __attribute__((address_space(1))) T*>(reinterpret_cast<gbytes>(as_global(base)) + idx * (unsigned)sizeof(T));
}
template <class T>
__device__ __forceinline__ void st_at(T* base, unsigned idx, T v) {
  typedef __attribute__((address_space(1))) char* gbytes;
  *reinterpret_cast<__attribute__((address_space(1))) T*>(reinterpret_cast<gbytes>((__attribute__((address_space(1))) T*)base) + idx * (unsigned)sizeof(T)) = v;
}

// A wave-uniform pointer made OPAQUE to the optimiser (two v_readfirstlane): the per-instance row pointer `array + b * n` of
// an output array.  Without it the compiler re-associates (array + b n) + lane into (array + lane) + b n, hoists the first
// sum — a 64-bit per-lane value — out of the instance loop and keeps it in two VGPRs through every Newton loop (the wave
// teams' DC kernels spilled exactly those, round 4); with it the store is scalar base + 32-bit lane offset (st_at).
template <class T>
__device__ __forceinline__ T* uniform_ptr(T* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}

// A loop index made opaque (an empty asm on its register): loop strength reduction otherwise turns `base[i]`, i = lane +
// 64 k, into a 64-bit POINTER induction variable whose start value base + 8 lane is invariant in the instance loop, gets
// hoisted to the kernel's entry and occupies two VGPRs through every Newton loop.  With an opaque index the access stays
// scalar base + 32-bit offset (st_at / ld_at), one shift per element in loops that run once per instance.
__device__ __forceinline__ unsigned opaque(unsigned i) { asm volatile("" : "+v"(i)); return i; }

struct Blk { double a11, a12, a21, a22; };
__device__ __forceinline__ Blk ld_blk(const double* blk, int id) {
  const double2* p = reinterpret_cast<const double2*>(blk + 4 * id);
  double2 r0 = p[0], r1 = p[1];
  return Blk{r0.x, r0.y, r1.x, r1.y};
}
__device__ __forceinline__ void st_blk(double* blk, int id, const Blk& b) {
  double2* p = reinterpret_cast<double2*>(blk + 4 * id);
  p[0] = make_double2(b.a11, b.a12);
  p[1] = make_double2(b.a21, b.a22);
}

constexpr int BT_PQ = 1, BT_PV = 2, BT_REF = 3, BT_PQ_HI = 4, BT_PQ_LO = 5;
// a bus that this instance's outage cuts off every REF bus: de-energised (pandapower's
// check_connectivity takes such buses out of service): identity rows, NaN results
constexpr int BT_DEAD = 6;

// Per-instance LDS image.  The lane-programme kernel (V2) keeps the voltage in
// rectangular form only (no |V|/angle arrays) to fit 6 instances per CU.
struct Lds {
  double *vr, *vi, *vm, *va, *psp, *qsp, *rhs, *blk, *sp, *acc;
  double* stage;             // LDS area behind rq: the LU block values live here (blk == stage) unless the grid is too large
                             // for that (memory-resident kernels: blk points into a per-workgroup row of global memory); outside
                             // the solve it holds the rest of the staged table row and the result bank
  unsigned char* bt;
  // second-generation kernels: structure-of-arrays images.  rhs = P-row values [nb], rq =
  // Q-row values [nb]; block component c of block id at blk[c * bs + id].  A wave's 64-bit
  // LDS accesses are served in two groups of 32 lanes over 32 eight-byte bank pairs: with
  // one double per id the bank depends on (id mod 32) only, whereas 32-byte block records
  // (16-byte rhs pairs) put every access of a group on 8 (16) of the 32 bank pairs.
  double* rq;
  int bs;                    // entries of the a11 / a12 arrays (all blocks)
  int nfull;                 // blocks [0, nfull) also have a21 / a22 entries (arrays at o2, o3)
  int o2, o3;
  double* mod;               // per-instance branch modifiers (MOD_DOUBLES each), see mods_*
  unsigned short* dg;        // [nb] diagonal block of every bus (copied from the plan once per workgroup)
  unsigned short* tl;        // [tail_m + 1][M] U-block ids of the dense tail (DevPlan::tail_ids; last row: none), likewise; 16-byte aligned
};
// Off-diagonal Jacobian blocks of PQ rows that no update targets keep the shape [[a, b], [-b, a]]
// (dS/dtheta = -j c, dS/dln|V| = c): the plan numbers them last and only (a, b) is stored.
// Where the four values of block `id` live in L.blk (element indices).  Default: one array per component (above).
// -DOPFX_PAIR_LAYOUT (probe): the rows of a block as 16-byte pairs — (a11, a12) at 2 id, (a21, a22) at o2 + 2 id — so
// that a row is ONE address and one two-value LDS access.
#ifdef OPFX_PAIR_LAYOUT
__device__ __forceinline__ int bx11(const Lds&, int id) { return 2 * id; }
__device__ __forceinline__ int bx12(const Lds&, int id) { return 2 * id + 1; }
__device__ __forceinline__ int bx21(const Lds& L, int id) { return L.o2 + 2 * id; }
__device__ __forceinline__ int bx22(const Lds& L, int id) { return L.o2 + 2 * id + 1; }
#else
__device__ __forceinline__ int bx11(const Lds&, int id) { return id; }
__device__ __forceinline__ int bx12(const Lds& L, int id) { return L.bs + id; }
__device__ __forceinline__ int bx21(const Lds& L, int id) { return L.o2 + id; }
__device__ __forceinline__ int bx22(const Lds& L, int id) { return L.o3 + id; }
#endif
__device__ __forceinline__ int blk_c(const Lds& L, int id, int c) { return c == 0 ? bx11(L, id) : c == 1 ? bx12(L, id) : c == 2 ? bx21(L, id) : bx22(L, id); }
template <bool PK>
__device__ __forceinline__ Blk ld_blk2(const Lds& L, int id) {
  Blk b{L.blk[bx11(L, id)], L.blk[bx12(L, id)], 0.0, 0.0};
  if (!PK || id < L.nfull) { b.a21 = L.blk[bx21(L, id)]; b.a22 = L.blk[bx22(L, id)]; }
  else { b.a21 = -b.a12; b.a22 = b.a11; }
  return b;
}
// second row of a block of a PV bus row (always a four-value block): the Q equation is replaced by d|V| = 0
__device__ __forceinline__ void blk_zero_row2(const Lds& L, int id) { L.blk[bx21(L, id)] = 0.0; L.blk[bx22(L, id)] = 0.0; }
template <bool PK>
__device__ __forceinline__ void st_blk2(const Lds& L, int id, const Blk& b) {
  L.blk[bx11(L, id)] = b.a11; L.blk[bx12(L, id)] = b.a12;
  if (!PK || id < L.nfull) { L.blk[bx21(L, id)] = b.a21; L.blk[bx22(L, id)] = b.a22; }
}
// Newton-Raphson on the instance in LDS.  Returns converged; *iters, *nrm out.
__device__ bool newton(const DevPlan& P, const Lds& L, const Opts& o, int lane,
                       int out_br, int* iters_out, double* nrm_out) {
  const int nb = P.nb;
  // outaged branch: positions of its four Ybus stamps and their values
  int op0 = -1, op1 = -1, op2 = -1, op3 = -1;
  double oy[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (out_br >= 0) {
    op0 = P.br_pos[out_br * 4 + 0]; op1 = P.br_pos[out_br * 4 + 1];
    op2 = P.br_pos[out_br * 4 + 2]; op3 = P.br_pos[out_br * 4 + 3];
#pragma unroll
    for (int q = 0; q < 8; ++q) oy[q] = P.br_y[out_br * 8 + q];
  }
  int it = 0;
  double nrm = 0.0;
  bool conv = false;
  while (true) {
    // ---- phase A: mismatch, inf-norm, Jacobian blocks (lane = bus row) -------
    for (int f = lane; f < P.nfill; f += WAVE) st_blk(L.blk, P.fill_blk[f], Blk{0.0, 0.0, 0.0, 0.0});
    double my = 0.0;
    for (int i = lane; i < nb; i += WAVE) {
      const int t = L.bt[i];
      const double vri = L.vr[i], vii = L.vi[i], vmi = L.vm[i];
      double ior = 0.0, ioi = 0.0, dr = 0.0, di = 0.0;   // off-diagonal sum, diagonal term
      const int e0 = P.y_ptr[i], e1 = P.y_ptr[i + 1];
      for (int e = e0; e < e1; ++e) {
        const int j = P.y_col[e];
        double g = P.y_g[e], b = P.y_b[e];
        if (out_br >= 0) {
          if (e == op0) { g -= oy[0]; b -= oy[1]; }
          if (e == op1) { g -= oy[2]; b -= oy[3]; }
          if (e == op2) { g -= oy[4]; b -= oy[5]; }
          if (e == op3) { g -= oy[6]; b -= oy[7]; }
        }
        const double vrj = L.vr[j], vij = L.vi[j];
        const double tr = g * vrj - b * vij, ti = g * vij + b * vrj;   // Y_ij V_j
        if (j == i) { dr = tr; di = ti; continue; }
        ior += tr; ioi += ti;
        const int bid = P.y_blk[e];
        if (bid >= 0 && t != BT_REF) {
          // c = V_i conj(Y_ij V_j);  dS_i/dth_j = -j c;  dS_i/d|V_j| = c/|V_j|
          const double cr = vri * tr + vii * ti, ci = vii * tr - vri * ti;
          const double inv = 1.0 / L.vm[j];
          Blk jb{ci, cr * inv, -cr, ci * inv};
          if (t == BT_PV) { jb.a21 = 0.0; jb.a22 = 0.0; }
          st_blk(L.blk, bid, jb);
        }
      }
      if (t != BT_REF) {
        const double ir = ior + dr, ii = ioi + di;              // I_i
        const double pc = vri * ir + vii * ii, qc = vii * ir - vri * ii;   // S_i = V_i conj(I_i)
        const double fp = pc - L.psp[i];
        const double fq = (t == BT_PV) ? 0.0 : qc - L.qsp[i];
        L.rhs[2 * i] = -fp;
        L.rhs[2 * i + 1] = -fq;
        my = nan_max(my, nan_max(fabs(fp), fabs(fq)));
        // dS_i/dth_i = j V_i conj(I_i - Y_ii V_i);  with e = V_i conj(Ioff): j e = -e.im + j e.re
        const double er = vri * ior + vii * ioi, ei = vii * ior - vri * ioi;
        // dS_i/d|V_i| = (V_i conj(Y_ii V_i) + S_i)/|V_i|
        const double yr = vri * dr + vii * di, yi = vii * dr - vri * di;
        Blk jb{-ei, (yr + pc) / vmi, er, (yi + qc) / vmi};
        if (t == BT_PV) { jb.a21 = 0.0; jb.a22 = 1.0; }
        st_blk(L.blk, P.diag_blk[i], jb);
      }
    }
    nrm = wave_max(my);
    if (!(nrm == nrm)) { conv = false; break; }          // NaN: diverged
    if (nrm < o.tol) { conv = true; break; }
    if (it >= o.max_iter) { conv = false; break; }
    ++it;
    wave_sync();
    // ---- phase B: block LU + forward substitution, level by level ------------
    for (int lev = 0; lev < P.nlev; ++lev) {
      const int t0 = P.lev_tptr[lev], t1 = P.lev_tptr[lev + 1];
      for (int t = t0 + lane; t < t1; t += WAVE) {
        const int tb = P.tgt_blk[t];
        const int s0 = P.tgt_sptr[t], s1 = P.tgt_sptr[t + 1];
        if (tb >= 0) {
          Blk a = ld_blk(L.blk, tb);
          for (int s = s0; s < s1; ++s) {
            const Blk bi = ld_blk(L.blk, P.src_ik[s]);
            const Blk bk = ld_blk(L.blk, P.src_kk[s]);
            const Blk bj = ld_blk(L.blk, P.src_kj[s]);
            const double r = 1.0 / (bk.a11 * bk.a22 - bk.a12 * bk.a21);
            const double w11 = (bi.a11 * bk.a22 - bi.a12 * bk.a21) * r;
            const double w12 = (bi.a12 * bk.a11 - bi.a11 * bk.a12) * r;
            const double w21 = (bi.a21 * bk.a22 - bi.a22 * bk.a21) * r;
            const double w22 = (bi.a22 * bk.a11 - bi.a21 * bk.a12) * r;
            a.a11 -= w11 * bj.a11 + w12 * bj.a21;
            a.a12 -= w11 * bj.a12 + w12 * bj.a22;
            a.a21 -= w21 * bj.a11 + w22 * bj.a21;
            a.a22 -= w21 * bj.a12 + w22 * bj.a22;
          }
          st_blk(L.blk, tb, a);
        } else {
          const int i = -1 - tb;
          double y1 = L.rhs[2 * i], y2 = L.rhs[2 * i + 1];
          for (int s = s0; s < s1; ++s) {
            const Blk bi = ld_blk(L.blk, P.src_ik[s]);
            const Blk bk = ld_blk(L.blk, P.src_kk[s]);
            const int k = P.src_kj[s];
            const double r1 = L.rhs[2 * k], r2 = L.rhs[2 * k + 1];
            const double r = 1.0 / (bk.a11 * bk.a22 - bk.a12 * bk.a21);
            const double z1 = (bk.a22 * r1 - bk.a12 * r2) * r;
            const double z2 = (bk.a11 * r2 - bk.a21 * r1) * r;
            y1 -= bi.a11 * z1 + bi.a12 * z2;
            y2 -= bi.a21 * z1 + bi.a22 * z2;
          }
          L.rhs[2 * i] = y1;
          L.rhs[2 * i + 1] = y2;
        }
      }
      wave_sync();
    }
    // ---- phase C: back substitution, levels in reverse ------------------------
    for (int lev = P.nlev - 1; lev >= 0; --lev) {
      const int p0 = P.lev_pptr[lev], p1 = P.lev_pptr[lev + 1];
      for (int q = p0 + lane; q < p1; q += WAVE) {
        const int k = P.piv_bus[q];
        double y1 = L.rhs[2 * k], y2 = L.rhs[2 * k + 1];
        const int u0 = P.piv_uptr[q], u1 = P.piv_uptr[q + 1];
        for (int u = u0; u < u1; ++u) {
          const Blk a = ld_blk(L.blk, P.u_blk[u]);
          const int j = P.u_col[u];
          const double x1 = L.rhs[2 * j], x2 = L.rhs[2 * j + 1];
          y1 -= a.a11 * x1 + a.a12 * x2;
          y2 -= a.a21 * x1 + a.a22 * x2;
        }
        const Blk bk = ld_blk(L.blk, P.diag_blk[k]);
        const double r = 1.0 / (bk.a11 * bk.a22 - bk.a12 * bk.a21);
        L.rhs[2 * k] = (bk.a22 * y1 - bk.a12 * y2) * r;
        L.rhs[2 * k + 1] = (bk.a11 * y2 - bk.a21 * y1) * r;
      }
      wave_sync();
    }
    // ---- phase D: update V (polar), lane = bus --------------------------------
    for (int i = lane; i < nb; i += WAVE) {
      if (L.bt[i] == BT_REF) continue;
      double va = L.va[i] + L.rhs[2 * i];
      double vm = L.vm[i] + L.rhs[2 * i + 1];
      if (vm < 0.0) { vm = -vm; va += M_PI; }     // V = Vm e^{jVa}; Vm = |V| (newtonpf)
      double s, c;
      sincos(va, &s, &c);
      L.va[i] = va; L.vm[i] = vm; L.vr[i] = vm * c; L.vi[i] = vm * s;
    }
    wave_sync();
  }
  *iters_out = it;
  *nrm_out = nrm;
  return conv;
}

// ---------------------------------------------------------------------------
// Newton-Raphson, second generation: driven by the plan's LANE PROGRAMME.
// All structure a lane needs comes as fixed-size descriptors laid out
// [round][lane] (coalesced, identical for every wave -> L1 resident) and is
// prefetched one round ahead, so no phase chases index chains through memory;
// update terms that share a target accumulate with LDS atomics (one wave, fixed
// lane order -> deterministic); the unknown for |V| is the relative step
// d|V|/|V|, which makes the Jacobian division-free.
// ---------------------------------------------------------------------------
// One wave per workgroup and the LDS executes a wave's operations in issue order, so
// making one lane's LDS writes visible to the other lanes needs no hardware wait at all:
// only the COMPILER must not move or cache LDS accesses across this point.  (A
// __syncthreads()/fence here would also drain the prefetched global loads: s_waitcnt vmcnt(0).)
// 1/x by hardware estimate + two Newton steps (relative error ~1e-16; no denormal/overflow
// special-casing: Jacobian pivots are O(1..1e4) in per-unit)
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}
// one Newton step (relative error ~ 1e-14): enough for the elimination multipliers — every item that uses a
// pivot block derives the same value from it, so the factorisation is that of a matrix perturbed by 1e-14
// and Newton's convergence does not notice; the solution itself (solve_pivot) takes two steps
__device__ __forceinline__ double fast_rcp1(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ void lds_sub(double* p, double v) {
  __hip_atomic_fetch_add(p, -v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_add(double* p, double v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

#ifdef OPFX_ENABLE_STAMPS
#define OPFX_STAMP_INIT() unsigned long long t_last__ = __builtin_readcyclecounter()
#define OPFX_STAMP_RESET() t_last__ = __builtin_readcyclecounter()
#define OPFX_STAMP(slot)                                                                   \
  do {                                                                                      \
    if (P.stamps && blockIdx.x == 0) {                                                       \
      const unsigned long long now__ = __builtin_readcyclecounter();                          \
      if (threadIdx.x == 0) P.stamps[slot] += now__ - t_last__;                                \
      t_last__ = __builtin_readcyclecounter();                                                \
    }                                                                                       \
  } while (0)
#else
// Product build: no probe code at all (a conditional store inside the Newton loops would
// make the compiler's wait-count insertion conservative).  Diagnostic build: -DOPFX_ENABLE_STAMPS.
#define OPFX_STAMP_INIT() do { } while (0)
#define OPFX_STAMP_RESET() do { } while (0)
#define OPFX_STAMP(slot) do { } while (0)
#endif

// All LDS reads of an item — the two factor blocks AND the third operand (the block A_kj or the right-hand
// side y_k) — are issued before anything is computed, branch-free: the reciprocal of the pivot determinant (a
// chain of ~15 dependent FP64 operations) then overlaps the return of the third operand instead of being
// followed by a second LDS round trip.  A two-value block reads its first row twice (clamped second-row
// addresses) and selects.  With one wavefront busy per instance (the dense tail of a meshed grid) nothing
// else hides that latency.
// Workgroup barrier that orders LDS traffic only: __syncthreads() would also drain the
// descriptor loads in flight (s_waitcnt vmcnt(0)) at every group end.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// MEM: the memory-resident form of the wave-team kernels (grids whose LU blocks do not fit the LDS: they live in a
// per-workgroup row of global memory, L2-resident).  Group ends then also wait for the vector-memory operations, and a
// wavefront that carries on alone waits for its own block updates before it reads them back.
template <bool MEM>
__device__ __forceinline__ void team_sync() { if (MEM) __syncthreads(); else lds_barrier(); }
template <bool MEM>
__device__ __forceinline__ void mem_fence() {
  if (MEM) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  wave_fence();
}
template <bool PK>
__device__ __forceinline__ void ld_blk_raw(const Lds& L, unsigned id, double& a11, double& a12, double& x21, double& x22) {
  const unsigned idc = (!PK || id < (unsigned)L.nfull) ? id : 0u;
  a11 = L.blk[bx11(L, id)]; a12 = L.blk[bx12(L, id)]; x21 = L.blk[bx21(L, idc)]; x22 = L.blk[bx22(L, idc)];
}
// The twelve values an item reads.  Loading (item_load) and using them (item_apply) are separate steps so
// that the reads of the NEXT round can be in flight while this round computes, whenever the plan marks the two
// rounds as independent (ITEM_NEXT_INDEPENDENT: same elimination level / same back-substitution group).
// (round 4: an item may carry a SECOND column of the same multiplier — word 2: target2 | A_kj2 << 16 — whose block is `e`;
// word 3 holds the round's flags in bits 0-1 and the rider's buses i / k in bits 2-16 / 17-31, 0x7FFF: none)
struct ItemRegs { double i11, i12, i21, i22, k11, k12, k21, k22, c11, c12, c21, c22, e11, e12, e21, e22, y1, y2; };
__device__ __forceinline__ unsigned rider_i(const uint4 d) { return (d.w >> 2) & 0x7FFFu; }
__device__ __forceinline__ unsigned rider_k(const uint4 d) { return d.w >> 17; }
constexpr unsigned ITEM_BARRIER = 1u, ITEM_NEXT_INDEPENDENT = 2u;   // flags in word 3 of a round's items (plan.cpp)
// SEC: how the stream's second columns are handled — SEC_NONE: it has none (back substitution: no test, no registers);
// SEC_EARLY: the block is read with the item's other blocks; SEC_LATE: it is read in item_apply, after the first column's
// updates have been issued, into the registers the first column has left (the wave-team kernels, which have no eight
// registers to spare: scratch otherwise).
constexpr int SEC_NONE = 0, SEC_EARLY = 1, SEC_LATE = 2;
template <bool PK, bool RIDERS, int SEC = SEC_EARLY>
__device__ __forceinline__ ItemRegs item_load(const Lds& L, const uint4 d) {
  constexpr unsigned NONE = 0xFFFFu;
  const unsigned tb = d.x & 0xFFFF;
  const bool live = tb != NONE;                     // (an empty item reads block 0 / bus 0 and adds nothing)
  const unsigned ik = live ? d.x >> 16 : 0u, kk = live ? d.y & 0xFFFF : 0u, kj = live ? d.y >> 16 : 0u;
  const bool rhs_t = live && (tb & 0x8000u) != 0;
  ItemRegs r;
  ld_blk_raw<PK>(L, ik, r.i11, r.i12, r.i21, r.i22);
  ld_blk_raw<false>(L, kk, r.k11, r.k12, r.k21, r.k22);        // (a pivot's diagonal block always holds four values, plan.cpp)
  // third operand C: A_kj, or the column (y_k ; .) of the right-hand side
  const unsigned kjb = rhs_t ? 0u : kj;                                  // (any valid block for the unused reads)
  const unsigned kjc = (!PK || kjb < (unsigned)L.nfull) ? kjb : 0u;
  const double* c1p = rhs_t ? (L.rhs + kj) : (L.blk + bx11(L, kj));
  const double* c3p = rhs_t ? (L.rq + kj) : (L.blk + bx21(L, kjc));
  r.c11 = *c1p; r.c21 = *c3p;
  r.c12 = L.blk[bx12(L, kjb)]; r.c22 = L.blk[bx22(L, kjc)];
  // the second column: read only where a lane of the wavefront has one (most rounds of a wave team have none, plan.cpp level_for)
  const bool two = SEC != SEC_NONE && live && (d.z & 0xFFFF) != NONE;
  r.e11 = 0.0; r.e12 = 0.0; r.e21 = 0.0; r.e22 = 0.0;
  if (SEC == SEC_EARLY && two) ld_blk_raw<PK>(L, d.z >> 16, r.e11, r.e12, r.e21, r.e22);
  // rider (plan.cpp): the item's multiplier also takes y_k to y_i — the forward substitution of the pair (i, k)
  r.y1 = 0.0; r.y2 = 0.0;
  if (RIDERS && rider_i(d) != 0x7FFFu) { const unsigned k = rider_k(d); r.y1 = L.rhs[k]; r.y2 = L.rq[k]; }
  return r;
}
template <bool PK, bool RIDERS, int SEC = SEC_EARLY>
__device__ __forceinline__ void item_apply(const Lds& L, const uint4 d, const ItemRegs& r) {
  constexpr unsigned NONE = 0xFFFFu;
  const unsigned tb = d.x & 0xFFFF;
  if (tb == NONE) return;
  const unsigned ik = d.x >> 16, kk = d.y & 0xFFFF, kj = d.y >> 16;
  const bool rhs_t = (tb & 0x8000u) != 0;
  Blk bi{r.i11, r.i12, r.i21, r.i22}, bk{r.k11, r.k12, r.k21, r.k22};
  if (PK) {
    const bool fi = ik < (unsigned)L.nfull;
    bi.a21 = fi ? r.i21 : -r.i12; bi.a22 = fi ? r.i22 : r.i11;
  }
  const bool fj = !PK || rhs_t || kj < (unsigned)L.nfull;
  const double c11 = r.c11, c12 = r.c12;
  const double c21 = fj ? r.c21 : -c12, c22 = fj ? r.c22 : c11;
  // m = -A_ik A_kk^-1 (the sign folded into the reciprocal: the update is an atomic ADD of m C)
  const double nrdet = fast_rcp(bk.a12 * bk.a21 - bk.a11 * bk.a22);
  const double m11 = (bi.a11 * bk.a22 - bi.a12 * bk.a21) * nrdet;
  const double m12 = (bi.a12 * bk.a11 - bi.a11 * bk.a12) * nrdet;
  const double m21 = (bi.a21 * bk.a22 - bi.a22 * bk.a21) * nrdet;
  const double m22 = (bi.a22 * bk.a11 - bi.a21 * bk.a12) * nrdet;
  const unsigned ti = tb & 0x7FFF;
  double* t1p = rhs_t ? (L.rhs + ti) : (L.blk + bx11(L, tb));
  double* t3p = rhs_t ? (L.rq + ti) : (L.blk + bx21(L, tb));
  lds_add(t1p, m11 * c11 + m12 * c21);
  lds_add(t3p, m21 * c11 + m22 * c21);
  if (!rhs_t) {
    lds_add(L.blk + bx12(L, tb), m11 * c12 + m12 * c22);
    lds_add(L.blk + bx22(L, tb), m21 * c12 + m22 * c22);
  }
  const unsigned tb2 = d.z & 0xFFFF;
  if (SEC != SEC_NONE && tb2 != NONE) {                      // the second column of the same multiplier: A_ij2 += m A_kj2 (a block target always)
    const unsigned ej = d.z >> 16;
    const bool fe = !PK || ej < (unsigned)L.nfull;
    double e11 = r.e11, e12 = r.e12, x21 = r.e21, x22 = r.e22;
    if (SEC == SEC_LATE) ld_blk_raw<PK>(L, ej, e11, e12, x21, x22);
    const double e21 = fe ? x21 : -e12, e22 = fe ? x22 : e11;
    lds_add(L.blk + bx11(L, tb2), m11 * e11 + m12 * e21);
    lds_add(L.blk + bx21(L, tb2), m21 * e11 + m22 * e21);
    lds_add(L.blk + bx12(L, tb2), m11 * e12 + m12 * e22);
    lds_add(L.blk + bx22(L, tb2), m21 * e12 + m22 * e22);
  }
  if (RIDERS && rider_i(d) != 0x7FFFu) {
    const unsigned i = rider_i(d);
    lds_add(L.rhs + i, m11 * r.y1 + m12 * r.y2);
    lds_add(L.rq + i, m21 * r.y1 + m22 * r.y2);
  }
}
// RIDERS: the stream may carry forward-substitution riders (plan.cpp).  Only the single-wave kernels' stream does:
// there a rider saves whole rounds (144-bus grid: 11 -> 8 rounds of factorisation, 0.286 -> 0.273 ms), whereas the
// wave teams walk one round per wavefront through most levels either way and the two tests per item cost more
// than the saved rounds give back (config 3: 1.967 -> 1.976 ms with riders, 2.06 ms with the tests but no riders).
template <bool PK, bool RIDERS, int SEC = SEC_EARLY>
__device__ __forceinline__ void item_factor(const Lds& L, const uint4 d) {
  if ((d.x & 0xFFFF) == 0xFFFFu) return;           // empty item: nothing read (idle waves of a team stay off the LDS)
  const ItemRegs r = item_load<PK, RIDERS, SEC>(L, d);
  item_apply<PK, RIDERS, SEC>(L, d, r);
}
// Two consecutive rounds of one wavefront.  When the plan marks the second as independent of the first (same group)
// all LDS reads of both are requested first.
template <bool PK, bool RIDERS>
__device__ __forceinline__ void item_pair(const Lds& L, const uint4 da, const uint4 db) {
  const unsigned fl = __builtin_amdgcn_readfirstlane(da.w);
  if (fl & ITEM_NEXT_INDEPENDENT) {
    const ItemRegs ra = item_load<PK, RIDERS>(L, da);
    const ItemRegs rb = item_load<PK, RIDERS>(L, db);
    item_apply<PK, RIDERS>(L, da, ra);
    item_apply<PK, RIDERS>(L, db, rb);
    wave_fence();
  } else {
    item_factor<PK, RIDERS>(L, da); wave_fence();
    item_factor<PK, RIDERS>(L, db); wave_fence();
  }
}
// Back substitution through the dense tail of the elimination (plan.cpp: the last m levels hold one pivot each
// and their U-rows are full): a strictly serial chain.  As LDS groups it costs one round trip + one 2x2 inverse
// per level with a handful of live lanes (13.7 % of the 306-bus step).  Here wavefront 0 runs it in registers
// between the two parts of the team stream: lane e owns tail pivot e and keeps y_e and A_ee^-1; step s sends
// x_s = A_ss^-1 y_s to the other lanes with v_readlane and the lanes e < s subtract U_es x_s.  The other
// wavefronts wait at the barrier meanwhile, so what counts is the NUMBER of instructions wavefront 0 issues
// (a lone wave issues in order), not the depth of the chain: per step four LDS reads (the lane's U-block ids
// came in one read; tail blocks always hold four values, plan.cpp), x on every lane (only lane s's is used),
// four readlanes, four FMAs, and a select with a compile-time lane mask instead of a branch.  Windows of
// OPFX_TAIL_W steps, double buffered; compiler barriers keep the scheduler from hoisting ALL reads of the
// unrolled chain (it spills).
__device__ __forceinline__ double readlane_f64(double v, int src) {
  const long long b = __double_as_longlong(v);
  const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)b, src), hi = __builtin_amdgcn_readlane((int)(unsigned)(b >> 32), src);
  return __longlong_as_double(((long long)hi << 32) | lo);
}
// FWD (kernels compiled with chord steps): the same chain run FORWARD first — the forward substitution through the tail,
// y_e -= L_es A_ss^-1 y_s for e > s, steps s = 0 .. m-2 — when `fwd` says so (a chord iteration: the factorisation items
// that carry the tail's forward substitution in an ordinary iteration are not walked).  The L-blocks' ids are the LOWER
// triangle of the same table (plan.cpp: entry [e][s], e > s, = block (row e, column s)); the backward steps mask the
// rows e >= s out before they form an address, so what they read is what they read without the lower triangle.
struct TailBlk { double u11, u12, u21, u22; bool live; };
template <int M, bool FWD>
__device__ __forceinline__ void tail_chain(const Lds& L, int row, int lane, double& y0, double& y1,
                                           double i11, double i12, double i21, double i22, bool fwd) {
  // the lane's row of U-block ids: M 16-bit entries, entry s = block (row e, column s)
  unsigned w[M / 2];
  {
    const uint4* tr = reinterpret_cast<const uint4*>(L.tl + row * M);
#pragma unroll
    for (int q = 0; q < M / 8; ++q) { const uint4 v = tr[q]; w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w; }
  }
#ifndef OPFX_TAIL_W
#define OPFX_TAIL_W 2
#endif
  constexpr int W = OPFX_TAIL_W;
  if (FWD && fwd) {
    auto request_f = [&](int s) {
      TailBlk t{0.0, 0.0, 0.0, 0.0, false};
      if (s > M - 2) return t;                             // (compile-time after unrolling)
      const unsigned id = (w[s >> 1] >> (16 * (s & 1))) & 0xFFFFu;
      t.live = lane > s && id != 0xFFFFu;
      const int ib = t.live ? (int)id : 0;
      t.u11 = L.blk[bx11(L, ib)]; t.u12 = L.blk[bx12(L, ib)]; t.u21 = L.blk[bx21(L, ib)]; t.u22 = L.blk[bx22(L, ib)];
      return t;
    };
    auto apply_f = [&](const TailBlk& t, int s) {
      if (s > M - 2) return;
      const double x0 = readlane_f64(i11 * y0 + i12 * y1, s), x1 = readlane_f64(i21 * y0 + i22 * y1, s);
      const double n0 = fma(-t.u11, x0, fma(-t.u12, x1, y0)), n1 = fma(-t.u21, x0, fma(-t.u22, x1, y1));
      y0 = t.live ? n0 : y0;
      y1 = t.live ? n1 : y1;
    };
    // (one step at a time, no double buffering: this direction runs in chord iterations only, and the wave-team step
    //  kernels have no registers to spare for a second window)
#pragma unroll
    for (int s = 0; s < M - 1; ++s) {
      const TailBlk t = request_f(s);
      asm volatile("" ::: "memory");
      apply_f(t, s);
      asm volatile("" ::: "memory");
    }
  }
  auto request = [&](int s) {
    TailBlk t{0.0, 0.0, 0.0, 0.0, false};
    if (s < 1) return t;                                   // (compile-time after unrolling)
    const unsigned id = (w[s >> 1] >> (16 * (s & 1))) & 0xFFFFu;
    t.live = lane < s && id != 0xFFFFu;                    // (rows s.. are final; a tail that is not completely filled in)
    const int ib = t.live ? (int)id : 0;
    t.u11 = L.blk[bx11(L, ib)]; t.u12 = L.blk[bx12(L, ib)]; t.u21 = L.blk[bx21(L, ib)]; t.u22 = L.blk[bx22(L, ib)];
    return t;
  };
  auto apply = [&](const TailBlk& t, int s) {
    if (s < 1) return;
    const double x0 = readlane_f64(i11 * y0 + i12 * y1, s), x1 = readlane_f64(i21 * y0 + i22 * y1, s);
    const double n0 = fma(-t.u11, x0, fma(-t.u12, x1, y0)), n1 = fma(-t.u21, x0, fma(-t.u22, x1, y1));
    y0 = t.live ? n0 : y0;
    y1 = t.live ? n1 : y1;
  };
  constexpr int NWIN = (M - 1 + W - 1) / W;                // steps M-1 .. 1 in windows of W
  TailBlk b[2][W];
#pragma unroll
  for (int q = 0; q < W; ++q) b[0][q] = request(M - 1 - q);
#pragma unroll
  for (int win = 0; win < NWIN; ++win) {
    const int s0 = M - 1 - W * win;
    if (win + 1 < NWIN) {
#pragma unroll
      for (int q = 0; q < W; ++q) b[(win + 1) & 1][q] = request(s0 - W - q);
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int q = 0; q < W; ++q) apply(b[win & 1][q], s0 - q);
    asm volatile("" ::: "memory");
  }
}
// `tail`: bus | diagonal block << 16 of the lane's tail pivot (DevPlan::tail_bus, read once per solve).
// The chain is instantiated for M = the next multiple of 8 >= tail_m; lanes >= tail_m carry y = 0 and an
// id row of "none" (row tail_m of the table).
template <bool FWD = false>
__device__ __forceinline__ void tail_solve(const Lds& L, int m, int lane, unsigned tail, bool fwd = false) {
  const bool mine = lane < m;
  const int bus = mine ? (int)(tail & 0xFFFFu) : 0;
  const int ib = mine ? (int)(tail >> 16) : 0;         // (diagonal blocks always hold four values)
  const double a11 = L.blk[bx11(L, ib)], a12 = L.blk[bx12(L, ib)], a21 = L.blk[bx21(L, ib)], a22 = L.blk[bx22(L, ib)];
  double y0 = L.rhs[bus], y1 = L.rq[bus];
  const double rdet = mine ? fast_rcp(a11 * a22 - a12 * a21) : 0.0;
  const double i11 = a22 * rdet, i12 = -a12 * rdet, i21 = -a21 * rdet, i22 = a11 * rdet;
  const int row = mine ? lane : m;                       // (row m of the table: all "none")
  if (m <= 8) tail_chain<8, FWD>(L, row, lane, y0, y1, i11, i12, i21, i22, fwd);
  else if (m <= 16) tail_chain<16, FWD>(L, row, lane, y0, y1, i11, i12, i21, i22, fwd);
  else if (m <= 24) tail_chain<24, FWD>(L, row, lane, y0, y1, i11, i12, i21, i22, fwd);
  else tail_chain<32, FWD>(L, row, lane, y0, y1, i11, i12, i21, i22, fwd);
  if (mine) { L.rhs[bus] = y0; L.rq[bus] = y1; }
}
// One round of a wave team's B/C stream.  (Issuing the NEXT round's reads before this round computes —
// the plan marks independent rounds, ITEM_NEXT_INDEPENDENT — was tried and is slower: the compiler's wait-count
// insertion treats loads that are pending across the loop's back edge conservatively and drains the LDS queue,
// lgkmcnt(0), at the first use, so the early reads only lengthen that wait: config 3 2.27 -> 2.66 ms.)
// zops (wave-uniform: the plan shares slots, plan.cpp share_slots): bits 2-16 of an item's word 3 name a four-value block id
// to be set to zero during this round — the slot of a block that is dead, for the fill block that is born into it at a later
// level (team items carry no rider there; 0x7FFF: none).
template <bool PK, bool MEM = false>
__device__ __forceinline__ void team_step(const Lds& L, const uint4 d, bool zops) {
  const unsigned fl = __builtin_amdgcn_readfirstlane(d.w);        // same for every item of a round
  item_factor<PK, false, SEC_LATE>(L, d);
  if (!MEM && zops) {        // (the memory-resident form needs no shared slots: its block values are not in LDS; refused at launch)
    const unsigned z = (d.w >> 2) & 0x7FFFu;
    // (stored through an explicit LDS pointer: with the generic one this ROCm's AMDGPU backend dies in two instantiations
    //  with "Illegal instruction detected: V_CMP_NE_U32_e32 0, $src_shared_base" — the LDS-to-flat cast of a uniform pointer)
    auto* const lb = (__attribute__((address_space(3))) double*)L.blk;
    if (z != 0x7FFFu) { lb[bx11(L, z)] = 0.0; lb[bx12(L, z)] = 0.0; lb[bx21(L, z)] = 0.0; lb[bx22(L, z)] = 0.0; }
  }
  if (fl & ITEM_BARRIER) team_sync<MEM>(); else mem_fence<MEM>();  // (no barrier: the same wavefront carries on)
}

#ifdef OPFX_PAIR_ROUNDS
template <bool PK>
__device__ __forceinline__ void team_pair(const Lds& L, const uint4 da, const uint4 db) {
  const unsigned fa = __builtin_amdgcn_readfirstlane(da.w), fb = __builtin_amdgcn_readfirstlane(db.w);
  if ((fa & ITEM_NEXT_INDEPENDENT) && !(fa & ITEM_BARRIER)) {
    const ItemRegs ra = item_load<PK, false>(L, da);
    const ItemRegs rb = item_load<PK, false>(L, db);
    item_apply<PK, false>(L, da, ra);
    item_apply<PK, false>(L, db, rb);
    if (fb & ITEM_BARRIER) lds_barrier(); else wave_fence();
  } else {
    team_step<PK>(L, da, false);
    team_step<PK>(L, db, false);
  }
}
#endif
// POLAR SHADOW (round 5).  The solver keeps V in rectangular form only, so the result bank used to take |V| = sqrt(vr^2 + vi^2)
// and the angle = atan2(vi, vr) of every bus after the solve: 38 + 121 vector instructions per bus round, 3.2 % of all the
// instructions of a 144-bus step, in a kernel whose time is its instruction count.  But phase D already holds both in
// polar form — V <- V (1 + d|V|/|V|) e^{j dth} — and the bus -> lane map of phase D (bus lane + 64 r) is that of the result
// pass: the lane keeps theta and |V| of its buses in REGISTERS (2 x POLAR_R doubles; the single-wave kernels have 80 VGPRs to
// spare at two wavefronts per SIMD), theta += dth and |V| *= 1 + d|V|/|V| per iteration, and the result pass reads them.
// Same values to rounding (the product / sum of the steps against sqrt / atan2 of the rotated vector: ~1e-16 per
// iteration).  Used where every solve of the launch starts from init_voltage and nothing rescales V behind the solver's
// back: single-wave step kernels specialised SPEC_NO_MOD (no modifiers, contingencies, per-instance |V| set-points), grids
// of at most 64 POLAR_R buses; everything else takes sqrt / atan2 as before.
constexpr int POLAR_R = 4;
// PQREG (round 6): in the same kernels, when the plan has no PV bus either (SPEC_NO_PV: nothing ever moves a bus's scheduled Q), the
// scheduled P / Q of the lane's buses stay in REGISTERS as well instead of in the workgroup's row of global memory: they are read-only
// after the prologue, and the 2 304 B row per 144-bus instance was written through to HBM for every instance (20 MB of the 92 MB a
// config-2 launch writes, EXPERIMENTS #33).  A register array wants a compile-time index: the bus rounds of phase A are unrolled
// over the POLAR_R rounds such a grid can have (round r of a lane is bus lane + 64 r, the map of the polar shadow).
#ifndef OPFX_POLAR_DC           // (the polar shadow also in the DC-start kernels: round 6; 0 = as before, for A/B builds)
#define OPFX_POLAR_DC 1
#endif
#ifndef OPFX_PQREG_DC           // (... also in the DC-start kernels: 256 VGPRs + 124 B of scratch — developer probe, EXPERIMENTS #36)
#define OPFX_PQREG_DC 0
#endif
#ifndef OPFX_PQREG
#define OPFX_PQREG 1
#endif
#ifndef OPFX_TPQ              // (the same for the wave teams, see TEAM_PQ_R; 0: probe builds)
#define OPFX_TPQ 1
#endif
struct Polar { double th[POLAR_R], vm[POLAR_R], p[POLAR_R], q[POLAR_R]; };
constexpr int TEAM_PQ_R = 2;       // bus rounds per wavefront of a team whose P / Q stay in registers: grids of up to 64 NW TEAM_PQ_R buses (do_step)
// Phase D of the lane-programme kernels: after the back-substitution items y_i of bus i holds its
// right-hand side with every U-term removed; x_i = A_ii^-1 y_i, then V_i <- V_i (1 + d|V|/|V|) e^{j dth}.
// `piv` keeps the smallest relative pivot seen by this lane: |det| / (|a11 a22| + |a12 a21|) of the 2x2
// diagonal block the bus is solved with (1 = no cancellation, -> 0 = the block is numerically singular:
// static pivoting inside the blocks has broken down, SURVEY §7 hard part 2).
// `pbus`: the bus that holds this lane's smallest pivot (where a breakdown sits: opfx_step_io.min_pivot_bus).
__device__ __forceinline__ void solve_pivot(const Lds& L, int i, double& dth, double& dvm, double& piv, int& pbus) {
  const int db = L.dg[i];
  // (diagonal blocks always hold four values)
  const double a11 = L.blk[bx11(L, db)], a12 = L.blk[bx12(L, db)], a21 = L.blk[bx21(L, db)], a22 = L.blk[bx22(L, db)];
  const double y1 = L.rhs[i], y2 = L.rq[i];
  const double p1 = a11 * a22, p2 = a12 * a21;
  const double det = p1 - p2;
  const double rdet = fast_rcp(det);
  // (an exactly singular block with vanishing products is 0 / 0: that is a pivot of zero, not "no information")
  const double den = fabs(p1) + fabs(p2);
  const double ratio = den == 0.0 ? 0.0 : fabs(det) * __builtin_amdgcn_rcp(den);
  pbus = ratio < piv ? i : pbus;
  piv = nn_min(piv, ratio);
  dth = (a22 * y1 - a12 * y2) * rdet;
  dvm = (a11 * y2 - a21 * y1) * rdet;
}

// One bus round of phase A for one lane (plan.h lp_apk): KA off-diagonal entries of the lane's row, the row's
// diagonal entry and its diagonal block.
constexpr int KA = opfx_plan::KA;
struct ARound { unsigned ent[KA]; double2 y[KA]; double2 yd; unsigned dw; };

__device__ __forceinline__ ARound load_around(const DevPlan& P, int r, int lane) {
  const uint4* q = reinterpret_cast<const uint4*>(P.lp_apk) + (size_t)r * opfx_plan::APK_VECS * WAVE + lane;
  ARound a;
  const uint4 e = q[0];
  const unsigned ew[4] = {e.x, e.y, e.z, e.w};
#pragma unroll
  for (int k = 0; k < KA; ++k) {
    a.ent[k] = ew[k];
    const uint4 w = q[(1 + k) * WAVE];
    a.y[k] = make_double2(__longlong_as_double(((long long)w.y << 32) | w.x),
                          __longlong_as_double(((long long)w.w << 32) | w.z));
  }
  a.dw = e.w;
  const uint4 w = q[(1 + KA) * WAVE];
  a.yd = make_double2(__longlong_as_double(((long long)w.y << 32) | w.x),
                      __longlong_as_double(((long long)w.w << 32) | w.z));
  return a;
}

// ---- per-instance branch modifiers ------------------------------------------------------
// An instance may differ from the shared Ybus in a few branches: one out of service (outage
// axis, N-1 contingency, an open switch), a transformer on another tap.  Each such branch is a
// MODIFIER: the change dY of its four stamps (ff, ft, tf, tt; 0 - Y for a removed branch) plus
// the ids the correction needs.  Phase A runs on the shared Ybus; afterwards two lanes per
// modifier (one per end bus) add V conj(dY V) to the mismatch, the (f,t)/(t,f) Jacobian blocks
// and the two diagonal blocks with LDS atomics (several modifiers may meet on one bus), and
// the mismatch norm is recomputed.  (Comparing every Ybus entry against the modifiers would
// cost ~25 instructions per entry in every solve of every instance.)
constexpr int MOD_DOUBLES = 12;     // dY[8] | ints: f, t, blk_ft, blk_tf, dblk_f, dblk_t, branch, state
// state of the modified branch: 0 coupled (another tap position), MOD_REMOVED out of service (no stamps at
// all), MOD_OPEN_ENDED behind one open switch (a shunt at its connected end, see case.py:open_ended_stamps):
// both of the latter connect nothing any more, only the first carries no current
constexpr int MOD_REMOVED = 1, MOD_OPEN_ENDED = 2;
__device__ __forceinline__ double* mod_dy(const Lds& L, int m) { return L.mod + m * MOD_DOUBLES; }
__device__ __forceinline__ int* mod_ids(const Lds& L, int m) { return reinterpret_cast<int*>(L.mod + m * MOD_DOUBLES + 8); }

// Writes modifier m for branch br.  Lanes 0..7 hold dY[lane] in `dy_lane` (ignored when
// `removed`: dY = -Y); lanes 8..12 fetch the ids.  Ends with a wave fence.
// A removal also cancels what the first n_prev modifiers changed on the same branch (a
// contingency on a transformer whose tap position differs from the compiled one).
__device__ __forceinline__ void mod_set(const DevPlan& P, const Lds& L, int lane, int m, int br, double dy_lane, bool removed, int n_prev,
                                        bool open_ended = false) {
  if (lane < 8) {
    double v = dy_lane;
    if (removed) {
      v = -P.br_y[br * 8 + lane];
      for (int j = 0; j < n_prev; ++j) if (mod_ids(L, j)[6] == br) v -= mod_dy(L, j)[lane];
    }
    mod_dy(L, m)[lane] = v;
  }
  int* id = mod_ids(L, m);
  if (lane == 8) { const int f = P.br_f[br]; id[0] = f; id[4] = P.diag_blk[f]; }
  if (lane == 9) { const int t = P.br_t[br]; id[1] = t; id[5] = P.diag_blk[t]; }
  // (a branch compiled without coupling — open-ended in the net itself — has no off-diagonal stamps: -1)
  if (lane == 10) { const int e = P.br_pos[br * 4 + 1]; id[2] = e >= 0 ? P.y_blk[e] : -1; }
  if (lane == 11) { const int e = P.br_pos[br * 4 + 2]; id[3] = e >= 0 ? P.y_blk[e] : -1; }
  if (lane == 12) { id[6] = br; id[7] = removed ? MOD_REMOVED : (open_ended ? MOD_OPEN_ENDED : 0); }
  wave_fence();
}

// Modifier m for a BUS SHUNT in steps (an ('shunt', 'step') actuator, opf_env.py:476-481): a branch modifier whose two ends
// are the same bus, without off-diagonal stamps and without a branch; lanes 6 / 7 hold the change (dG, dB) of the bus's shunt
// admittance against the compiled case — the "to" end's self admittance, which is the end mods_inline picks when both are
// the bus.  Ends with a wave fence.
__device__ __forceinline__ void mod_set_shunt(const DevPlan& P, const Lds& L, int lane, int m, int bus, double dy_lane) {
  if (lane < 8) mod_dy(L, m)[lane] = lane >= 6 ? dy_lane : 0.0;
  int* id = mod_ids(L, m);
  if (lane == 8) { id[0] = bus; id[1] = bus; }
  if (lane == 9) { const int d = P.diag_blk[bus]; id[4] = d; id[5] = d; }
  if (lane == 10) { id[2] = -1; id[3] = -1; }
  if (lane == 12) { id[6] = -1; id[7] = 0; }
  wave_fence();
}

// De-energised buses (mark_island): their rows become identity rows — off-diagonal blocks 0,
// diagonal block I, right-hand side 0.  Runs after phase A in the modifier path only (an island
// always comes with the modifier of the branch that cut it off), re-reading the row descriptors.
// jac = false (a chord iteration: the blocks hold the factorisation of an earlier iteration): right-hand sides only.
template <bool PK>
__device__ void dead_rows_patch(const DevPlan& P, const Lds& L, int lane, bool jac = true) {
  constexpr unsigned NONE = 0xFFFFu;
  bool any = false;
  for (int i = lane; i < P.nb; i += WAVE) any = any || L.bt[i] == BT_DEAD;
  if (!__any(any)) return;
  const uint4* hpk = reinterpret_cast<const uint4*>(P.lp_hpk) + lane;
  for (int h = 0; h < P.rh; ++h) {
    const uint4 he = hpk[(size_t)(h * 2 + 1) * WAVE];
    const unsigned bid = he.x >> 16;
    if (jac && (he.x & 0xFFFF) != NONE && bid != NONE && L.bt[he.y] == BT_DEAD) st_blk2<PK>(L, bid, Blk{0.0, 0.0, 0.0, 0.0});
  }
  for (int r = 0; r < P.ra; ++r) {
    const ARound a = load_around(P, r, lane);
    const int i = lane + WAVE * r;
    if (i >= P.nb || L.bt[i] != BT_DEAD) continue;
    const unsigned (&ent)[KA] = a.ent;
    if (jac) {
#pragma unroll
      for (int k = 0; k < KA; ++k) if ((ent[k] >> 16) != NONE) st_blk2<PK>(L, ent[k] >> 16, Blk{0.0, 0.0, 0.0, 0.0});
      st_blk2<PK>(L, a.dw & 0xFFFF, Blk{1.0, 0.0, 0.0, 1.0});
    }
    L.rhs[i] = 0.0; L.rq[i] = 0.0;
  }
}

// lanes 0 .. 2*n_mod-1: end e = lane & 1 of modifier lane >> 1
__device__ __forceinline__ void mods_apply(const Lds& L, int lane, int n_mod, bool jac = true) {
  if (lane >= 2 * n_mod) return;
  const int m = lane >> 1, e = lane & 1;
  const double* dy = mod_dy(L, m);
  const int* id = mod_ids(L, m);
  const int i = id[e], j = id[1 - e], ob = id[2 + e], db = id[4 + e];
  const double yii_g = dy[e ? 6 : 0], yii_b = dy[e ? 7 : 1], yij_g = dy[e ? 4 : 2], yij_b = dy[e ? 5 : 3];
  const double vri = L.vr[i], vii = L.vi[i], vrj = L.vr[j], vij = L.vi[j];
  const double tr = yij_g * vrj - yij_b * vij, ti = yij_g * vij + yij_b * vrj;
  const double dcr = vri * tr + vii * ti, dci = vii * tr - vri * ti;      // V_i conj(dY_ij V_j)
  const double v2 = vri * vri + vii * vii;
  const double dyr = yii_g * v2, dyi = -yii_b * v2;                         // conj(dY_ii)|V_i|^2
  const int t = L.bt[i];
  if (t == BT_DEAD) return;                // de-energised end: identity row
  if (t == BT_REF) {                       // parked injection S_i
    lds_add(&L.rhs[i], dcr + dyr);
    lds_add(&L.rq[i], dci + dyi);
    return;
  }
  const bool pv = t == BT_PV;
  lds_add(&L.rhs[i], -(dcr + dyr));        // rhs = -F
  if (!pv) lds_add(&L.rq[i], -(dci + dyi));
  if (!jac) return;                        // (chord iteration: the mismatch only)
  if (ob >= 0) {                           // dS_i/dth_j = -j c, dS_i/dln|V_j| = c
    lds_add(L.blk + blk_c(L, ob, 0), dci); lds_add(L.blk + blk_c(L, ob, 1), dcr);
    if (!pv && ob < L.nfull) { lds_add(L.blk + blk_c(L, ob, 2), -dcr); lds_add(L.blk + blk_c(L, ob, 3), dci); }   // (two-value blocks: implied)
  }
  // {-S_off.im, Y|V|^2.re + P, S_off.re, Y|V|^2.im + Q}
  lds_add(L.blk + blk_c(L, db, 0), -dci); lds_add(L.blk + blk_c(L, db, 1), 2.0 * dyr + dcr);
  if (!pv) { lds_add(L.blk + blk_c(L, db, 2), dcr); lds_add(L.blk + blk_c(L, db, 3), 2.0 * dyi + dci); }
}

// The same correction from inside the bus round of phase A, by the lane that owns bus i (its row sums sr/si, its
// diagonal term dyr/dyi, its own off-diagonal block): no pass of its own, no barriers, no second norm
// reduction — 3.8 k of the 12 k cycles of a wave-team phase A on the N-1 workload.  Only without de-energised
// buses (their identity rows are patched after the phase, dead_rows_patch); `t` = bus type of i.
__device__ __forceinline__ void mods_inline(const Lds& L, int n_mod, int i, int t, double vri, double vii,
                                            double& sr, double& si, double& dyr, double& dyi, bool jac = true) {
  for (int m = 0; m < n_mod; ++m) {
    const int* id = mod_ids(L, m);
    const int f = id[0], tt = id[1];
    if (i != f && i != tt) continue;
    const int e = i == tt ? 1 : 0;
    const double* dy = mod_dy(L, m);
    const int j = id[1 - e], ob = id[2 + e];
    const double yii_g = dy[e ? 6 : 0], yii_b = dy[e ? 7 : 1], yij_g = dy[e ? 4 : 2], yij_b = dy[e ? 5 : 3];
    const double vrj = L.vr[j], vij = L.vi[j];
    const double tr = yij_g * vrj - yij_b * vij, ti = yij_g * vij + yij_b * vrj;
    const double dcr = vri * tr + vii * ti, dci = vii * tr - vri * ti;      // V_i conj(dY_ij V_j)
    const double v2 = vri * vri + vii * vii;
    dyr += yii_g * v2; dyi -= yii_b * v2;                                    // conj(dY_ii)|V_i|^2
    sr += dcr; si += dci;
    if (jac && ob >= 0 && t != BT_REF) {   // this lane stored the block earlier in this phase: plain read-modify-write
      L.blk[bx11(L, ob)] += dci; L.blk[bx12(L, ob)] += dcr;
      if (t != BT_PV && ob < L.nfull) { L.blk[bx21(L, ob)] -= dcr; L.blk[bx22(L, ob)] += dci; }
    }
  }
}

// current injected at bus i by the modifiers: I_i += dY_ii V_i + dY_ij V_j
__device__ __forceinline__ void mods_row_current(const Lds& L, int n_mod, int i, double& ir, double& ii) {
  for (int m = 0; m < n_mod; ++m) {
    const int* id = mod_ids(L, m);
    const double* dy = mod_dy(L, m);
    const int f = id[0], t = id[1];
    if (i == f) {
      ir += dy[0] * L.vr[f] - dy[1] * L.vi[f] + dy[2] * L.vr[t] - dy[3] * L.vi[t];
      ii += dy[0] * L.vi[f] + dy[1] * L.vr[f] + dy[2] * L.vi[t] + dy[3] * L.vr[t];
    }
    if (i == t) {
      ir += dy[4] * L.vr[f] - dy[5] * L.vi[f] + dy[6] * L.vr[t] - dy[7] * L.vi[t];
      ii += dy[4] * L.vi[f] + dy[5] * L.vr[f] + dy[6] * L.vi[t] + dy[7] * L.vr[t];
    }
  }
}

// Smallest pivot of the wavefront and the bus it belongs to: the pivots are in [0, 1], so their bit patterns order like
// integers; the low 15 bits of the mantissa make room for the bus number and one min-reduction carries both.
__device__ __forceinline__ void piv_argmin(double piv, int pbus, double* piv_out, int* pbus_out) {
  const unsigned long long key = ((unsigned long long)__double_as_longlong(piv) & ~0x7FFFull) | (unsigned long long)(pbus < 0 ? 0x7FFF : pbus);
  const unsigned long long best = (unsigned long long)__double_as_longlong(wave_min_dpp(__longlong_as_double((long long)key)));
  *piv_out = wave_min_dpp(piv);
  const int b = (int)(best & 0x7FFFull);
  *pbus_out = b == 0x7FFF ? -1 : b;
}

// ---- DC start (opfx_solve_opts.init = OPFX_INIT_DC; pandapower init='dc', pypower dcpf) -----------------------------
// theta of the free buses from B' theta = P - (phase-shift injections + Gs + B'_ref theta_ref), |V| as in the flat start.
// B' has the Ybus pattern, so the solve runs through the Newton schedule itself: every block becomes [[B'_ij, 0], [0,
// B'_ij]] (which the two-value storage (a, b) -> [[a, b], [-b, a]] represents exactly), the right-hand side (P_i - c_i, 0);
// phases B and C as in an iteration; then V_i = |V_i| e^{j x_i}.  One linear solve before the first iteration: a PASS OF
// THE NEWTON LOOP ITSELF (`dc_pass`, kernels instantiated with DC) whose phase A writes B' instead of the Jacobian and
// whose phase D sets the angles — phases B / C and the dense tail's register chain are the loop's own code, not a second
// inlined copy (round 3 had one: the wave-team kernels went to 255-256 VGPRs + 36 B of scratch, k_solve to 716 B).
// PQ: the scheduled P of the lane's buses comes from registers (single-wave PQREG kernels, `pol`: rounds at compile-time positions)
template <bool PK, bool PQ = false>
__device__ __forceinline__ void dc_rows(const DevPlan& P, const Lds& L, int first, int stride, int lane, const Polar* pol = nullptr) {
  constexpr unsigned NONE = 0xFFFFu;
  const int nb = P.nb;
  auto row_round = [&](const int r, const bool from_regs, const double p_reg) __attribute__((always_inline)) {
    const ARound a = load_around(P, r, lane);
    const double* dc = P.lp_dc + (size_t)r * (KA + 2) * WAVE + lane;
    double bij[KA];
#pragma unroll
    for (int k = 0; k < KA; ++k) bij[k] = dc[k * WAVE];
    const double bii = dc[KA * WAVE], cst = dc[(KA + 1) * WAVE];
    const int i = lane + WAVE * r;
    if (i >= nb) return;
    const double p_sched = from_regs ? p_reg : L.psp[i];
    if (L.bt[i] == BT_REF) return;
#pragma unroll
    for (int k = 0; k < KA; ++k) {
      const unsigned bid = a.ent[k] >> 16;
      if (bid != NONE) st_blk2<PK>(L, bid, Blk{bij[k], 0.0, 0.0, bij[k]});
    }
    st_blk2<PK>(L, a.dw & 0xFFFF, Blk{bii, 0.0, 0.0, bii});
    L.rhs[i] = p_sched - cst;
    L.rq[i] = 0.0;
  };
  if (PQ) {
#pragma unroll
    for (int r = 0; r < POLAR_R; ++r) if (r < P.ra) row_round(r, true, pol->p[r]);
  } else {
    for (int r = first; r < P.ra; r += stride) row_round(r, false, 0.0);
  }
}
template <bool PK>
__device__ __forceinline__ void dc_overflow(const DevPlan& P, const Lds& L, int first, int stride, int lane) {
  constexpr unsigned NONE = 0xFFFFu;
  const uint4* hpk = reinterpret_cast<const uint4*>(P.lp_hpk);
  for (int h = first; h < P.rh; h += stride) {
    const uint4 he = hpk[(size_t)(h * 2 + 1) * WAVE + lane];
    const double b = P.lp_hdc[(size_t)h * WAVE + lane];
    const unsigned bid = he.x >> 16;
    if ((he.x & 0xFFFF) != NONE && bid != NONE) st_blk2<PK>(L, bid, Blk{b, 0.0, 0.0, b});
  }
}
// The DC start of a solve with branches OUT OF SERVICE (an outage, an N-1 contingency, an open line switch: modifiers of kind
// MOD_REMOVED) — pandapower runs its DC power flow on the net as it is, i.e. without them: B' and the constant part of the
// right-hand side (plan.cpp) less the share of every removed branch k = (f, t): B'_ff -= b, B'_tt -= b, B'_ft += b, B'_tf += b;
// c_f -= Pfinj, c_t += Pfinj; a REF end contributes B'_ir theta_r to c_i.  Two lanes per modifier, one per end, after the
// pass has written the compiled values (blocks [[B', 0], [0, B']]: the first and, where stored, the fourth component).
// Solves with other modifiers (a tap position, an open-ended branch, a shunt step change B' in ways the per-branch arrays
// do not describe) start flat: dc_start_possible.
__device__ __forceinline__ void dc_mods(const DevPlan& P, const Lds& L, int lane, int n_mod) {
  if (lane >= 2 * n_mod) return;
  const int m = lane >> 1, e = lane & 1;
  const int* id = mod_ids(L, m);
  const int i = id[e], j = id[1 - e], ob = id[2 + e], db = id[4 + e], br = id[6];
  if (L.bt[i] == BT_REF) return;
  const double b = P.br_bdc[br], pf = P.br_pfinj[br];
  if (ob >= 0) {
    lds_add(L.blk + blk_c(L, ob, 0), b);
    if (ob < L.nfull) lds_add(L.blk + blk_c(L, ob, 3), b);
  }
  lds_add(L.blk + blk_c(L, db, 0), -b);
  lds_add(L.blk + blk_c(L, db, 3), -b);
  double dr = e == 0 ? pf : -pf;                          // rhs = P - c
  if (L.bt[j] == BT_REF) dr -= b * P.va_set[j];           // (c_i held B'_ij theta_j = -b theta_j)
  lds_add(&L.rhs[i], dr);
}
// every modifier of the solve takes a branch out of service (wave-uniform)
__device__ __forceinline__ bool dc_start_possible(const DevPlan& P, const Lds& L, int n_mod) {
  if (n_mod == 0) return true;
  if (P.br_bdc == nullptr) return false;
  bool ok = true;
  for (int m = 0; m < n_mod; ++m) ok = ok && mod_ids(L, m)[7] == MOD_REMOVED && mod_ids(L, m)[6] >= 0;
  return ok;
}
// CHORD: compiled with chord steps (opfx_solve_opts.jacobian_reuse_tol > 0; kernels of their own like DC): an iteration may
// keep the factorisation of an earlier one — phase A then computes the mismatch only (`jac` false: no block is written) and
// the wavefront walks the CHORD stream, forward substitution alone + the same back substitution (plan.h lp_bcc), instead
// of factorisation + forward substitution.  Which stream the NEXT iteration walks is known once this iteration's norm is,
// i.e. before its own rounds run out, so the four descriptors in flight across the loop's back edge come from the right one.
template <bool PK, bool DC = false, bool CHORD = false, int SPEC = 0, bool POLAR = false>
__device__ bool newton2(const DevPlan& P, const Lds& L, const Opts& o, int lane, int n_mod,
                        int* iters_out, double* nrm_out, double* piv_out, int* pbus_out, bool dc_pass = false, Polar* pol = nullptr) {
  double piv = 1.0;
  int pbus = -1;
  constexpr bool NOPV = (SPEC & SPEC_NO_PV) != 0, NOMOD = (SPEC & SPEC_NO_MOD) != 0;
  constexpr bool PQREG = OPFX_PQREG && POLAR && NOPV && !(DC && !OPFX_PQREG_DC);      // scheduled P / Q in pol->p / pol->q (see Polar)
  constexpr unsigned NONE = 0xFFFFu;
  const int nb = P.nb;
  // Descriptor streams: every load below is UNCONDITIONAL and sits in straight-line code, so
  // that the compiler's wait-count insertion sees a fixed number of loads in flight and
  // waits for the oldest only (a conditional load anywhere in these loops degrades every
  // wait to vmcnt(0), i.e. one exposed L2 round trip per round).
  const uint4* stream = reinterpret_cast<const uint4*>(P.lp_bc) + lane;
  const uint4* hpk = reinterpret_cast<const uint4*>(P.lp_hpk) + lane;
  const int RB_main = P.rb, R_main = P.rb + P.rc;        // padded round counts (multiples of 4, R >= 4)
  const uint4* const stream_c = CHORD ? reinterpret_cast<const uint4*>(P.lp_bcc) + lane : stream;
  const uint4 *st_cur = stream, *st_nxt = stream;        // stream of this iteration / of the next one
  int RB = RB_main, R = R_main;                          // rounds of this iteration's stream
  bool chord_now = false;                                // this iteration keeps the blocks as they are
  double e_prev = 0.0;
  auto ld_desc = [&](int r) { return r < R ? st_cur[(size_t)r * WAVE] : st_nxt[(size_t)(r - R) * WAVE]; };
  const int hrow0 = lane < P.n_hrows ? P.lp_hrows[lane] : -1;
  const int fill_lo = P.fill_lo;                           // fill blocks: ids [fill_lo, fill_lo + nfill) (plan.cpp)

  int it = 0;
  double nrm = 0.0;
  bool conv = false;
  OPFX_STAMP_INIT();
  ARound cur = load_around(P, 0, lane);
  // scheduled P/Q of this lane's row of the next round, fetched with the descriptors (global row, see carve)
  const double* psp_g = L.psp; const double* qsp_g = L.qsp;
  double pcur = 0.0, qcur = 0.0;
  if (!PQREG) { pcur = psp_g[lane < nb ? lane : nb - 1]; qcur = qsp_g[lane < nb ? lane : nb - 1]; }
  uint4 hy = make_uint4(0, 0, 0, 0), he = make_uint4(NONE | (NONE << 16), 0, 0, 0);
  if (P.rh > 0) { hy = hpk[0]; he = hpk[WAVE]; }
  // rounds 0..3 of phases B/C; re-loaded by the tail of phase C for the next iteration
  uint4 q0 = ld_desc(0), q1 = ld_desc(1), q2 = ld_desc(2), q3 = ld_desc(3);
  while (true) {
    const bool jac = !(CHORD && chord_now);
    // ---- phase A -----------------------------------------------------------------
    if (jac) for (int f = fill_lo + lane; f < fill_lo + P.nfill; f += WAVE) st_blk2<PK>(L, f, Blk{0.0, 0.0, 0.0, 0.0});
    OPFX_STAMP(10);
    double my = 0.0;
    if (DC && dc_pass) {
      // the DC start: B' in the blocks, P - c on the right-hand side (see dc_rows); phases B / C below solve it
      dc_overflow<PK>(P, L, 0, 1, lane);
      dc_rows<PK, PQREG>(P, L, 0, 1, lane, pol);
      wave_fence();
      if (!NOMOD && n_mod > 0) { dc_mods(P, L, lane, n_mod); wave_fence(); }
    } else {
    // overflow entries of rows longer than the ELL width: any row per lane, row sums accumulated in the
    // rhs slots of those rows (zeroed first) with LDS atomics
    if (P.rh > 0) {
      if (hrow0 >= 0) { L.rhs[hrow0] = 0.0; L.rq[hrow0] = 0.0; }
      for (int h = lane + WAVE; h < P.n_hrows; h += WAVE) { const int i = P.lp_hrows[h]; L.rhs[i] = 0.0; L.rq[i] = 0.0; }
      wave_fence();
      for (int h = 0; h < P.rh; ++h) {
        const uint4 cy = hy, ce = he;
        const int hn = h + 1 < P.rh ? h + 1 : 0;           // next round (or round 0 of the next iteration)
        hy = hpk[(size_t)(hn * 2) * WAVE]; he = hpk[(size_t)(hn * 2 + 1) * WAVE];
        const unsigned ent = ce.x;
        const unsigned j = ent & 0xFFFF;
        if (j != NONE) {
          const int i = ce.y;
          const double g = __longlong_as_double(((long long)cy.y << 32) | cy.x);
          const double b = __longlong_as_double(((long long)cy.w << 32) | cy.z);
          const double vrj = L.vr[j], vij = L.vi[j], vri = L.vr[i], vii = L.vi[i];
          const double tr = g * vrj - b * vij, ti = g * vij + b * vrj;
          const double cr = vri * tr + vii * ti, ci = vii * tr - vri * ti;
          const unsigned bid = ent >> 16;
          const int t = L.bt[i];
          if (bid != NONE && jac) { st_blk2<PK>(L, bid, Blk{ci, cr, -cr, ci}); if (!NOPV && t == BT_PV) blk_zero_row2(L, bid); }
          lds_add(&L.rhs[i], cr);
          lds_add(&L.rq[i], ci);
        }
      }
      wave_fence();
    }
    OPFX_STAMP(11);
    // one bus round: rows lane + 64 r (the descriptors of the NEXT round are requested first)
    auto bus_round = [&](const int r, const double p_sched, const double q_sched) __attribute__((always_inline)) {
      const ARound a = cur;
      {
        const int rn = r + 1 < P.ra ? r + 1 : 0;                 // next round (or round 0 of the next iteration)
        cur = load_around(P, rn, lane);
        if (!PQREG) {
          const int in_ = lane + WAVE * rn < nb ? lane + WAVE * rn : nb - 1;
          pcur = psp_g[in_]; qcur = qsp_g[in_];
        }
      }
      const int i = lane + WAVE * r;
      if (i < nb) {
        const int t = L.bt[i];
        const double vri = L.vr[i], vii = L.vi[i];
        double sr = 0.0, si = 0.0;                       // S_off = V_i conj(sum_{j!=i} Y_ij V_j)
        if (a.dw >> 16) { sr = L.rhs[i]; si = L.rq[i]; }
        const unsigned (&ent)[KA] = a.ent;
        // branch-free over the ELL slots (padding slots carry Y = 0 and read V_i): the four
        // dependency chains interleave instead of being serialised by exec-mask branches
#pragma unroll
        for (int k = 0; k < KA; ++k) {
          const unsigned j = ent[k] & 0xFFFF;               // (padding slots: own row, Y = 0)
          const double g = a.y[k].x, b = a.y[k].y;
          const double vrj = L.vr[j], vij = L.vi[j];
          const double tr = g * vrj - b * vij, ti = g * vij + b * vrj;
          const double cr = vri * tr + vii * ti, ci = vii * tr - vri * ti;
          sr += cr; si += ci;
          const unsigned bid = ent[k] >> 16;
          if (bid != NONE && jac) {                          // (rows and columns of REF buses have no blocks)
            // dS_i/dth_j = -j c ; dS_i/dln|V_j| = c ; PV rows are patched after the loop
            st_blk2<PK>(L, bid, Blk{ci, cr, -cr, ci});
          }
        }
        if (!NOPV && t == BT_PV && jac) {                      // rare: skipped as a whole when the wave has no PV row
#pragma unroll
          for (int k = 0; k < KA; ++k) if ((ent[k] >> 16) != NONE) blk_zero_row2(L, ent[k] >> 16);
        }
        if (t != BT_REF) {
          const double g = a.yd.x, b = a.yd.y;
          const double v2 = vri * vri + vii * vii;
          const double yr = g * v2, yi = -b * v2;          // V_i conj(Y_ii V_i) = conj(Y_ii)|V_i|^2
          const double pc = sr + yr, qc = si + yi;
          const double fp = pc - p_sched;
          const double fq = (!NOPV && t == BT_PV) ? 0.0 : qc - q_sched;
          L.rhs[i] = -fp;
          L.rq[i] = -fq;
          my = nn_max(my, nn_max(fabs(fp), fabs(fq)));
          // dS_i/dth_i = j S_off ; dS_i/dln|V_i| = V_i conj(Y_ii V_i) + S_i
          Blk jb{-si, yr + pc, sr, yi + qc};
          if (!NOPV && t == BT_PV) { jb.a21 = 0.0; jb.a22 = 1.0; }
          if (jac) st_blk2<PK>(L, a.dw & 0xFFFF, jb);
        } else {
          // REF row: no equation; park the calculated injection S_i = S_off + conj(Y_ii)|V_i|^2
          // in its rhs slots so that the result pass needs no second walk over the row
          const double g = a.yd.x, b = a.yd.y;
          const double v2 = vri * vri + vii * vii;
          L.rhs[i] = sr + g * v2;
          L.rq[i] = si - b * v2;
        }
      }
    };
    if (PQREG) {
#pragma unroll
      for (int r = 0; r < POLAR_R; ++r) if (r < P.ra) bus_round(r, pol->p[r], pol->q[r]);
    } else {
      for (int r = 0; r < P.ra; ++r) { const double p_sched = pcur, q_sched = qcur; bus_round(r, p_sched, q_sched); }
    }
    // (the wave teams fold the modifiers into the bus rounds, mods_inline; here, where the kernel's common case has
    //  none, even the test for it in the bus round costs 1.5 % — measured — so they keep their own pass)
    if (!NOMOD && n_mod > 0) {             // rare: outage / contingency / switch / tap (see mods_apply)
      wave_fence();
      dead_rows_patch<PK>(P, L, lane, jac);
      wave_fence();
      mods_apply(L, lane, n_mod, jac);
      wave_fence();
      my = 0.0;
      for (int i = lane; i < nb; i += WAVE)
        if (L.bt[i] != BT_REF) my = nn_max(my, nn_max(fabs(L.rhs[i]), fabs(L.rq[i])));
    }
    OPFX_STAMP(12);
    // the wave-uniform decisions are votes (two instructions each); the max reduction (~45
    // instructions) runs once, on the way out.  Same outcome as testing the reduced norm:
    // NaN anywhere -> not converged; every row below tol -> converged; else iterate to max_iter.
    const bool below = !wave_any(!(my < o.tol));               // (false if any lane holds a NaN)
    if (below || wave_any(my != my) || it >= o.max_iter) {
      nrm = wave_max_dpp(my);
      conv = below;
      OPFX_STAMP(1);
      break;
    }
    ++it;
    if (CHORD && o.reuse_tol > 0.0) {
      // keep this iteration's factorisation for the next one?  Only while the step before this iteration cut the mismatch
      // at least tenfold (Newton is past its slow start / the chord step did its job; never after the first iteration:
      // e_prev starts at 0) and, for an iteration that factorised, only once its own mismatch is below the threshold.
      const double e = wave_max_dpp(my);
      const bool chord_next = e < 0.1 * e_prev && (!jac || e < o.reuse_tol);
      e_prev = e;
      st_nxt = chord_next ? stream_c : stream;
    }
    wave_fence();
    }
    // ---- phase B: block LU + forward substitution; phase C: back substitution ---------------
    // Rounds of one level are independent; ordering is needed at level ends only, but on a
    // single wave the fence is free (the LDS executes a wave's operations in order).
#ifdef OPFX_PAIR_ROUNDS
    // Two rounds of one elimination level (the plan's ITEM_NEXT_INDEPENDENT flag) as one step: the LDS reads of both
    // are requested before either computes, so the second round's round trip hides behind the first round's arithmetic.
    for (int r = 0; r < RB; r += 4) {
      item_pair<PK, true>(L, q0, q1); q0 = ld_desc(r + 4); q1 = ld_desc(r + 5);
      item_pair<PK, true>(L, q2, q3); q2 = ld_desc(r + 6); q3 = ld_desc(r + 7);
    }
    OPFX_STAMP(2);
    for (int r = RB; r < R; r += 4) {
      item_pair<PK, false>(L, q0, q1); q0 = ld_desc(r + 4); q1 = ld_desc(r + 5);
      item_pair<PK, false>(L, q2, q3); q2 = ld_desc(r + 6); q3 = ld_desc(r + 7);
    }
#else
    for (int r = 0; r < RB; r += 4) {
      item_factor<PK, true>(L, q0); wave_fence(); q0 = ld_desc(r + 4);
      item_factor<PK, true>(L, q1); wave_fence(); q1 = ld_desc(r + 5);
      item_factor<PK, true>(L, q2); wave_fence(); q2 = ld_desc(r + 6);
      item_factor<PK, true>(L, q3); wave_fence(); q3 = ld_desc(r + 7);
    }
    OPFX_STAMP(2);
    for (int r = RB; r < R; r += 4) {
      item_factor<PK, false, SEC_NONE>(L, q0); wave_fence(); q0 = ld_desc(r + 4);
      item_factor<PK, false, SEC_NONE>(L, q1); wave_fence(); q1 = ld_desc(r + 5);
      item_factor<PK, false, SEC_NONE>(L, q2); wave_fence(); q2 = ld_desc(r + 6);
      item_factor<PK, false, SEC_NONE>(L, q3); wave_fence(); q3 = ld_desc(r + 7);
    }
#endif
    OPFX_STAMP(3);
    // ---- phase D: x_i = A_ii^-1 y_i, V <- V (1 + d|V|/|V|) e^{j dth}  (rectangular update, no |V|/angle arrays) ----
    auto phase_d = [&](int i, int r) {
      if (L.bt[i] == BT_REF) return;                   // (rhs of a REF row holds its parked injection)
      double dth, dvm;
      solve_pivot(L, i, dth, dvm, piv, pbus);
      // (the DC pass solved for the ANGLE itself: turn the start voltage by the difference to its start angle, |V| stays)
      if (DC && dc_pass) { if (P.theta0) st_at(P.theta0 + (size_t)blockIdx.x * 2 * P.nb, (unsigned)i, dth); dth -= P.va_set[i]; dvm = 0.0; }
      const double sc = 1.0 + dvm;
      // (the polar shadow of this lane's buses, see Polar; a step past |V| = 0 — 1 + d|V|/|V| < 0 — turns V by pi)
      if (POLAR) { pol->th[r] += sc < 0.0 ? dth + M_PI : dth; pol->vm[r] *= fabs(sc); }
      double sn, cs;
      if (fabs(dth) <= 0.25) {
        // Taylor series to x^15 / x^14, truncation error < 1e-21 (the other branch is skipped as a whole
        // while no lane needs it: after the first iteration the steps are small)
        const double z = dth * dth;
        sn = dth * (1.0 + z * (-1.0 / 6 + z * (1.0 / 120 + z * (-1.0 / 5040 + z * (1.0 / 362880
             + z * (-1.0 / 39916800 + z * (1.0 / 6227020800.0 + z * (-1.0 / 1307674368000.0))))))));
        cs = 1.0 + z * (-0.5 + z * (1.0 / 24 + z * (-1.0 / 720 + z * (1.0 / 40320 + z * (-1.0 / 3628800
             + z * (1.0 / 479001600 + z * (-1.0 / 87178291200.0)))))));
      } else {
        sincos(dth, &sn, &cs);                         // (a NaN step ends here and stays NaN)
      }
      const double vr = L.vr[i], vi = L.vi[i];
      L.vr[i] = (vr * cs - vi * sn) * sc;
      L.vi[i] = (vr * sn + vi * cs) * sc;
    };
    if (POLAR) {
#pragma unroll
      for (int r = 0; r < POLAR_R; ++r) { const int i = lane + WAVE * r; if (i < nb) phase_d(i, r); }
    } else {
      for (int i = lane; i < nb; i += WAVE) phase_d(i, 0);
    }
    wave_fence();
    OPFX_STAMP(4);
    if (DC) dc_pass = false;
    if (CHORD) {                      // the next iteration's stream (its first four rounds are in flight already)
      chord_now = st_nxt != stream;
      st_cur = st_nxt;
      RB = chord_now ? P.rf : RB_main;
      R = RB + P.rc;
    }
  }
  *iters_out = it;
  *nrm_out = nrm;
  piv_argmin(piv, pbus, piv_out, pbus_out);
  return conv;
}

// ---------------------------------------------------------------------------
// Cooperative variant for LARGE grids: NW wavefronts (one workgroup) share ONE
// instance.  Meshed HV grids need 70-90 KB of LDS per instance, i.e. only one
// or two instances fit a CU; with one wave each, three of the four SIMDs would
// idle.  Here the independent rounds of a group (an elimination level, its U
// pre-items, its solves), the bus rows and the voltage update are dealt round-
// robin to the NW waves and groups are separated by workgroup barriers.  Update
// terms of different waves meet in LDS atomics, so the summation order — and the
// last bits of the result — may differ between runs (the single-wave kernel is
// bit-reproducible).
// ---------------------------------------------------------------------------
// Workgroup barrier that orders LDS traffic only: __syncthreads() would also drain the
// descriptor loads in flight (s_waitcnt vmcnt(0)) at every group end.
template <int NW, bool PK, bool MEM = false, bool DC = false, bool CHORD = false, int SPEC = 0>
__device__ bool newton2_coop(const DevPlan& P, const Lds& L, const Opts& o, int n_mod,
                             int* iters_out, double* nrm_out, double* piv_out, int* pbus_out, bool inline_mods, bool dc_pass = false,
                             const Polar* pol = nullptr) {
  double piv = 1.0;
  int pbus = -1;
  constexpr bool NOPV = (SPEC & SPEC_NO_PV) != 0, NOMOD = (SPEC & SPEC_NO_MOD) != 0;
  // TPQ: the scheduled P / Q of the thread's buses in registers (pol->p / q [TEAM_PQ_R]) — team kernels of plans without a PV bus,
  // where nothing changes them after the prologue (see Polar / PQREG; bus tid + NT k is round wave + NW k of this wavefront)
  constexpr bool TPQ = OPFX_PQREG && OPFX_TPQ && NOPV && !MEM && NW > 1;
  constexpr unsigned NONE = 0xFFFFu;
  constexpr int NT = WAVE * NW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nb = P.nb;
  const uint4* stream = reinterpret_cast<const uint4*>(NW == 2 ? P.lp_team2 : P.lp_team4) + (size_t)wave * WAVE + lane;
  const int K_main = NW == 2 ? P.team_rounds2 : P.team_rounds4;   // multiple of 4, >= 4
  const int Kb_main = NW == 2 ? P.team_kb2 : P.team_kb4;          // rounds before the tail chain (= K without a tail)
  // chord steps (see newton2): the team's chord stream — forward substitution | (tail chain) | back substitution
  const uint4* const stream_c = CHORD ? reinterpret_cast<const uint4*>(NW == 2 ? P.lp_teamc2 : P.lp_teamc4) + (size_t)wave * WAVE + lane : stream;
  const uint4 *st_cur = stream, *st_nxt = stream;
  int K = K_main, Kb = Kb_main;
  bool chord_now = false;
  double e_prev = 0.0;
  auto ld_desc = [&](int k) { return k < K ? st_cur[(size_t)k * (NW * WAVE)] : st_nxt[(size_t)(k - K) * (NW * WAVE)]; };   // unconditional (see newton2)
  const uint4* hpk = reinterpret_cast<const uint4*>(P.lp_hpk);
  double* xw = L.acc;                     // [NW] cross-wave scratch (reuses the constraint accumulators)
  const bool zops = P.n_shared > 0;
  int it = 0;
  double nrm = 0.0;
  bool conv = false;
  OPFX_STAMP_INIT();
  uint4 q0 = ld_desc(0), q1 = ld_desc(1), q2 = ld_desc(2), q3 = ld_desc(3);
  const unsigned tail = P.tail_bus[lane & 31];     // the lane's pivot of the dense tail (tail_solve)
  // overflow entries of rows longer than the ELL width (phase A): the row this thread zeroes and this wavefront's first round
  const int hrow0 = tid < P.n_hrows ? P.lp_hrows[tid] : -1;
  uint4 hy0 = make_uint4(0, 0, 0, 0), he0 = make_uint4(NONE | (NONE << 16), 0, 0, 0);
  if (wave < P.rh) { hy0 = hpk[(size_t)(wave * 2) * WAVE + lane]; he0 = hpk[(size_t)(wave * 2 + 1) * WAVE + lane]; }
  // this wave's next bus round (descriptors + scheduled P/Q of the row), one round ahead
  const double* psp_g = L.psp; const double* qsp_g = L.qsp;
  const int r_first = wave < P.ra ? wave : 0;
  ARound a_next = load_around(P, r_first, lane);
  double p_next = 0.0, q_next = 0.0;
  if (!TPQ) { const int in_ = lane + WAVE * r_first < nb ? lane + WAVE * r_first : nb - 1; p_next = psp_g[in_]; q_next = qsp_g[in_]; }
  while (true) {
    // ---- phase A ------------------------------------------------------------------
    // (nothing in this prologue of the phase waits for global memory: fill blocks are one contiguous id range,
    // the overflow rows and this wavefront's first overflow round were fetched once per solve)
    const bool jac = !(CHORD && chord_now);
    if (jac) for (int f = P.fill_lo + tid; f < P.fill_lo + P.nfill; f += NT) st_blk2<PK>(L, f, Blk{0.0, 0.0, 0.0, 0.0});
    // the DC start (see dc_rows) is a pass of this loop: B' in the blocks, P - c on the right-hand side, through the SAME
    // overflow / bus-round loops (their descriptors are in registers already), then phases B / C as in an iteration
    const bool dcp = DC && dc_pass;
    if (hrow0 >= 0) { L.rhs[hrow0] = 0.0; L.rq[hrow0] = 0.0; }
    for (int h = tid + NT; h < P.n_hrows; h += NT) { const int i = P.lp_hrows[h]; L.rhs[i] = 0.0; L.rq[i] = 0.0; }
    team_sync<MEM>();
    OPFX_STAMP(10);
    for (int h = wave; h < P.rh; h += NW) {
      const uint4 hy = h == wave ? hy0 : hpk[(size_t)(h * 2) * WAVE + lane];
      const uint4 he = h == wave ? he0 : hpk[(size_t)(h * 2 + 1) * WAVE + lane];
      const unsigned ent = he.x;
      const unsigned j = ent & 0xFFFF;
      if (DC && dcp) {
        const double b = P.lp_hdc[(size_t)h * WAVE + lane];
        const unsigned bid = ent >> 16;
        if (j != NONE && bid != NONE) st_blk2<PK>(L, bid, Blk{b, 0.0, 0.0, b});
      } else if (j != NONE) {
        const int i = he.y;
        double g = __longlong_as_double(((long long)hy.y << 32) | hy.x);
        double b = __longlong_as_double(((long long)hy.w << 32) | hy.z);
        const double vrj = L.vr[j], vij = L.vi[j], vri = L.vr[i], vii = L.vi[i];
        const double tr = g * vrj - b * vij, ti = g * vij + b * vrj;
        const double cr = vri * tr + vii * ti, ci = vii * tr - vri * ti;
        const unsigned bid = ent >> 16;
        const int t = L.bt[i];
        if (bid != NONE && jac) {
          Blk jb{ci, cr, -cr, ci};
          if (!NOPV && t == BT_PV) { jb.a21 = 0.0; jb.a22 = 0.0; }
          st_blk2<PK>(L, bid, jb);
        }
        lds_add(&L.rhs[i], cr);
        lds_add(&L.rq[i], ci);
      }
    }
    team_sync<MEM>();
    OPFX_STAMP(11);
    double my = 0.0;
    auto bus_round = [&](const int r, const double p_sched, const double q_sched) __attribute__((always_inline)) {
      const ARound a = a_next;
      {
        const int rn = r + NW < P.ra ? r + NW : r_first;       // (last round: first round of the next iteration)
        a_next = load_around(P, rn, lane);
        if (!TPQ) {
          const int in_ = lane + WAVE * rn < nb ? lane + WAVE * rn : nb - 1;
          p_next = psp_g[in_]; q_next = qsp_g[in_];
        }
      }
      const int i = lane + WAVE * r;
      if (DC && dcp) {
        // (one value at a time, each used before the next is requested: this pass runs once per solve, its latency does
        //  not matter, the registers of a batched load would — the kernel sits at the top of the register file)
        const double* dc = P.lp_dc + (size_t)r * (KA + 2) * WAVE + lane;      // ([ra][KA + 2][64]: padded, any lane is valid)
        if (i < nb && L.bt[i] != BT_REF) {
#pragma unroll
          for (int k = 0; k < KA; ++k) {
            const unsigned bid = a.ent[k] >> 16;
            const double bij = dc[k * WAVE];
            if (bid != NONE) st_blk2<PK>(L, bid, Blk{bij, 0.0, 0.0, bij});
            asm volatile("" ::: "memory");
          }
          const double bii = dc[KA * WAVE];
          st_blk2<PK>(L, a.dw & 0xFFFF, Blk{bii, 0.0, 0.0, bii});
          asm volatile("" ::: "memory");
          L.rhs[i] = p_sched - dc[(KA + 1) * WAVE];
          L.rq[i] = 0.0;
        }
      } else if (i < nb) {
        const int t = L.bt[i];
        const double vri = L.vr[i], vii = L.vi[i];
        double sr = 0.0, si = 0.0;
        if (a.dw >> 16) { sr = L.rhs[i]; si = L.rq[i]; }
        const unsigned (&ent)[KA] = a.ent;
#pragma unroll
        for (int k = 0; k < KA; ++k) {
          const unsigned j = ent[k] & 0xFFFF;               // (padding slots: own row, Y = 0)
          double g = a.y[k].x, b = a.y[k].y;
          const double vrj = L.vr[j], vij = L.vi[j];
          const double tr = g * vrj - b * vij, ti = g * vij + b * vrj;
          const double cr = vri * tr + vii * ti, ci = vii * tr - vri * ti;
          sr += cr; si += ci;
          const unsigned bid = ent[k] >> 16;
          if (bid != NONE && jac) st_blk2<PK>(L, bid, Blk{ci, cr, -cr, ci});   // (PV rows are patched after the loop)
        }
        if (!NOPV && t == BT_PV && jac) {
#pragma unroll
          for (int k = 0; k < KA; ++k) if ((ent[k] >> 16) != NONE) blk_zero_row2(L, ent[k] >> 16);
        }
        double dyr = 0.0, dyi = 0.0;
        if (!NOMOD && inline_mods && n_mod > 0) mods_inline(L, n_mod, i, t, vri, vii, sr, si, dyr, dyi, jac);
        double g = a.yd.x, b = a.yd.y;
        const double v2 = vri * vri + vii * vii;
        const double yr = g * v2 + dyr, yi = -b * v2 + dyi;
        if (t != BT_REF) {
          const double pc = sr + yr, qc = si + yi;
          const double fp = pc - p_sched;
          const double fq = (!NOPV && t == BT_PV) ? 0.0 : qc - q_sched;
          L.rhs[i] = -fp;
          L.rq[i] = -fq;
          my = nn_max(my, nn_max(fabs(fp), fabs(fq)));
          Blk jb{-si, yr + pc, sr, yi + qc};
          if (!NOPV && t == BT_PV) { jb.a21 = 0.0; jb.a22 = 1.0; }
          if (jac) st_blk2<PK>(L, a.dw & 0xFFFF, jb);
        } else {
          L.rhs[i] = sr + yr;
          L.rq[i] = si + yi;
        }
      }
    };
    if (TPQ) {
#pragma unroll
      for (int k = 0; k < TEAM_PQ_R; ++k) { const int r = wave + NW * k; if (r < P.ra) bus_round(r, pol->p[k], pol->q[k]); }
    } else {
      for (int r = wave; r < P.ra; r += NW) { const double p_sched = p_next, q_sched = q_next; bus_round(r, p_sched, q_sched); }
    }
    OPFX_STAMP(12);
    if (!(DC && dcp)) {
    if (!NOMOD && n_mod > 0 && !inline_mods) {
      team_sync<MEM>();
      if (wave == 0) { dead_rows_patch<PK>(P, L, lane, jac); mem_fence<MEM>(); mods_apply(L, lane, n_mod, jac); }
      team_sync<MEM>();
      my = 0.0;
      for (int i = tid; i < nb; i += NT)
        if (L.bt[i] != BT_REF) my = nn_max(my, nn_max(fabs(L.rhs[i]), fabs(L.rq[i])));
    }
    my = wave_max_dpp(my);
    if (lane == 0) xw[wave] = my;
    team_sync<MEM>();
    nrm = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) nrm = nn_max(nrm, xw[w]);
    OPFX_STAMP(1);
    if (!(nrm == nrm)) { conv = false; break; }
    if (nrm < o.tol) { conv = true; break; }
    if (it >= o.max_iter) { conv = false; break; }
    ++it;
    if (CHORD && o.reuse_tol > 0.0) {          // (see newton2; every wavefront holds the same norm)
      const bool chord_next = nrm < 0.1 * e_prev && (!jac || nrm < o.reuse_tol);
      e_prev = nrm;
      st_nxt = chord_next ? stream_c : stream;
    }
    } else {
      team_sync<MEM>();        // (the DC pass has no norm exchange: its blocks and right-hand side are complete here)
      if (!NOMOD && n_mod > 0) { if (wave == 0) dc_mods(P, L, lane, n_mod); team_sync<MEM>(); }
    }
    // ---- phases B and C: this wave's rounds in its own stream (4 in flight), a barrier where a
    // group of mutually independent rounds ends ------------------------------------------------
#ifdef OPFX_PAIR_ROUNDS
    for (int k = 0; k < Kb; k += 4) {
      team_pair<PK>(L, q0, q1); q0 = ld_desc(k + 4); q1 = ld_desc(k + 5);
      team_pair<PK>(L, q2, q3); q2 = ld_desc(k + 6); q3 = ld_desc(k + 7);
    }
#else
    for (int k = 0; k < Kb; k += 4) {
      team_step<PK, MEM>(L, q0, zops); q0 = ld_desc(k + 4);
      team_step<PK, MEM>(L, q1, zops); q1 = ld_desc(k + 5);
      team_step<PK, MEM>(L, q2, zops); q2 = ld_desc(k + 6);
      team_step<PK, MEM>(L, q3, zops); q3 = ld_desc(k + 7);
    }
#endif
    OPFX_STAMP(2);
    if (P.tail_m > 0) {          // the dense tail's back substitution: a register chain on wavefront 0
      if (wave == 0) tail_solve<CHORD>(L, P.tail_m, lane, tail, CHORD && chord_now);
      team_sync<MEM>();
      OPFX_STAMP(20);
#ifdef OPFX_PAIR_ROUNDS
      for (int k = Kb; k < K; k += 4) {
        team_pair<PK>(L, q0, q1); q0 = ld_desc(k + 4); q1 = ld_desc(k + 5);
        team_pair<PK>(L, q2, q3); q2 = ld_desc(k + 6); q3 = ld_desc(k + 7);
      }
#else
      for (int k = Kb; k < K; k += 4) {
        team_step<PK, MEM>(L, q0, zops); q0 = ld_desc(k + 4);
        team_step<PK, MEM>(L, q1, zops); q1 = ld_desc(k + 5);
        team_step<PK, MEM>(L, q2, zops); q2 = ld_desc(k + 6);
        team_step<PK, MEM>(L, q3, zops); q3 = ld_desc(k + 7);
      }
#endif
    }
    OPFX_STAMP(3);
    // ---- phase D ---------------------------------------------------------------------------------
    for (int i = tid; i < nb; i += NT) {
      if (L.bt[i] == BT_REF) continue;
      double dth, dvm;
      solve_pivot(L, i, dth, dvm, piv, pbus);
      // (the DC pass solved for the ANGLE itself: turn the start voltage by the difference to its start angle, |V| stays)
      if (DC && dcp) { if (P.theta0) st_at(P.theta0 + (size_t)blockIdx.x * 2 * nb, (unsigned)i, dth); dth -= P.va_set[i]; dvm = 0.0; }
      const double sc = 1.0 + dvm;
      double sn, cs;
      if (fabs(dth) <= 0.25) {                         // (as in newton2: Taylor series, truncation error < 1e-21)
        const double z = dth * dth;
        sn = dth * (1.0 + z * (-1.0 / 6 + z * (1.0 / 120 + z * (-1.0 / 5040 + z * (1.0 / 362880
             + z * (-1.0 / 39916800 + z * (1.0 / 6227020800.0 + z * (-1.0 / 1307674368000.0))))))));
        cs = 1.0 + z * (-0.5 + z * (1.0 / 24 + z * (-1.0 / 720 + z * (1.0 / 40320 + z * (-1.0 / 3628800
             + z * (1.0 / 479001600 + z * (-1.0 / 87178291200.0)))))));
      } else {
        sincos(dth, &sn, &cs);
      }
      const double vr = L.vr[i], vi = L.vi[i];
      L.vr[i] = (vr * cs - vi * sn) * sc;
      L.vi[i] = (vr * sn + vi * cs) * sc;
    }
    team_sync<MEM>();
    OPFX_STAMP(4);
    if (DC) dc_pass = false;
    if (CHORD) {
      chord_now = st_nxt != stream;
      st_cur = st_nxt;
      K = chord_now ? (NW == 2 ? P.team_roundsc2 : P.team_roundsc4) : K_main;
      Kb = chord_now ? (NW == 2 ? P.team_kbc2 : P.team_kbc4) : Kb_main;
    }
  }
  team_sync<MEM>();          // xw (aliases the constraint accumulators) is free again
  {
    int wbus;
    piv_argmin(piv, pbus, &piv, &wbus);
    if (lane == 0) { xw[wave] = piv; xw[NW + wave] = (double)wbus; }
    team_sync<MEM>();
    piv = xw[0];
    pbus = (int)xw[NW];
#pragma unroll
    for (int w = 1; w < NW; ++w) { const double pw = xw[w]; pbus = pw < piv ? (int)xw[NW + w] : pbus; piv = nn_min(piv, pw); }
    team_sync<MEM>();
  }
  *iters_out = it;
  *nrm_out = nrm;
  *piv_out = piv;
  *pbus_out = pbus;
  return conv;
}

// (Re)start an instance: flat/shift-aware start voltages and the grid's bus types.
// A bus that an earlier solve of this instance pinned at a reactive limit gets
// its generator share removed from q_sp again (L.bt must hold valid codes).
template <int V2, int SPEC = 0>
__device__ void init_voltage(const DevPlan& P, const Lds& L, int lane, const double* qg_min,
                             const double* qg_max, bool pin_point_ranges, int stride = WAVE) {
  constexpr bool NOPV = (SPEC & SPEC_NO_PV) != 0;           // (no PV bus: nothing is ever pinned at a reactive limit)
  for (int i = lane; i < P.nb; i += stride) {
    const int t = NOPV ? BT_PQ : L.bt[i];
    if (t == BT_PQ_HI) L.qsp[i] -= qg_max[i];
    if (t == BT_PQ_LO) L.qsp[i] -= qg_min[i];
    if (!V2) { L.vm[i] = P.vm_set[i]; L.va[i] = P.va_set[i]; }
    L.vr[i] = P.vr0[i]; L.vi[i] = P.vi0[i];
    int bt0 = P.bus_type[i];
    // A generator whose reactive range is a single point (eco_dispatch.py:86-88 sets
    // min_q = max_q = 0) always ends at that limit after the first enforce_q_lims pass
    // (unless its free Q happens to equal it exactly): start it there, one solve saved
    // (opfx_solve_opts.enforce_q_lims = 1; = 2 walks pypower's path: every generator starts as PV).
    if (!NOPV && bt0 == BT_PV && pin_point_ranges && qg_min != nullptr && qg_min[i] == qg_max[i]) {
      bt0 = BT_PQ_HI;
      L.qsp[i] += qg_max[i];
    }
    // A NaN reactive injection at a PV bus: the bus's Q mismatch is no equation of the power flow, so the solve would converge
    // around it — pypower's does not (makeSbus builds P + 1j * Q, and 1j * NaN is NaN + NaN j: the ACTIVE mismatch of the bus is NaN
    // as well).  Reachable: voltage_control.py:123-125 takes sqrt(max_s^2 - p^2) for the reactive range of a controllable unit,
    // NaN where p exceeds max_s by rounding, and the action's set-point with it (found by the fuzzer once such a unit sat on a
    // regulated bus, round 6).
    if (!NOPV && bt0 == BT_PV && L.qsp[i] != L.qsp[i]) L.psp[i] = L.qsp[i];
    L.bt[i] = (unsigned char)bt0;
  }
}

// Outer loop: NR + enforce_q_lims PV->PQ switching.
// 0: no islanding outage among the n_rem removed branches; 1: exactly one removed branch and it
// islands (its cut-off set is precomputed); 2: islanding with several branches out (unknown set)
__device__ __forceinline__ int island_state(bool v2, int n_rem, bool any_island) {
  return !any_island ? 0 : ((v2 && n_rem == 1) ? 1 : 2);
}

// De-energise the buses that outage `br` cuts off (single outage only: the island sets are
// precomputed per branch, plan.cpp).  Call after init_voltage.
__device__ __forceinline__ void mark_island(const DevPlan& P, const Lds& L, int lane, int br,
                                            const double* qg_min, const double* qg_max) {
  for (int q = P.isl_ptr[br] + lane; q < P.isl_ptr[br + 1]; q += WAVE) {
    const int i = P.isl_bus[q];
    const int t = L.bt[i];
    if (t == BT_PQ_HI) L.qsp[i] -= qg_max[i];       // undo the start pin of init_voltage
    if (t == BT_PQ_LO) L.qsp[i] -= qg_min[i];
    L.bt[i] = (unsigned char)BT_DEAD;
  }
}

// Several branches out at once (open switches and/or an outage and/or a contingency): whether and what
// they cut off is decided per instance by label propagation from the REF buses over the branches that
// remain (a rare path: O(diameter) sweeps over the branch list; flags in the not-yet-used rhs area).
// Unreached buses are de-energised exactly as mark_island does.  Call after init_voltage and mod_set.
__device__ void mark_islands_multi(const DevPlan& P, const Lds& L, int lane, int n_mod,
                                   const double* qg_min, const double* qg_max) {
  unsigned char* reach = reinterpret_cast<unsigned char*>(L.rhs);
  for (int i = lane; i < P.nb; i += WAVE) reach[i] = P.bus_type[i] == BT_REF ? 1 : 0;
  wave_fence();
  for (int sweep = 0; sweep < P.nb; ++sweep) {
    int changed = 0;
    for (int k = lane; k < P.nbr; k += WAVE) {
      bool removed = false;
      for (int m = 0; m < n_mod; ++m) { const int* id = mod_ids(L, m); removed = removed || (id[6] == k && id[7] != 0); }
      if (removed) continue;
      const int f = P.br_f[k], t = P.br_t[k];
      if (reach[f] != reach[t]) { reach[f] = 1; reach[t] = 1; changed = 1; }
    }
    wave_fence();
    if (!wave_any(changed)) break;
  }
  for (int i = lane; i < P.nb; i += WAVE) {
    if (reach[i]) continue;
    const int t = L.bt[i];
    if (t == BT_PQ_HI) L.qsp[i] -= qg_max[i];       // undo the start pin of init_voltage
    if (t == BT_PQ_LO) L.qsp[i] -= qg_min[i];
    L.bt[i] = (unsigned char)BT_DEAD;
  }
  wave_fence();
}

// DC: compiled with the DC start (opfx_solve_opts.init).  A template parameter, i.e. kernels of their own: with the DC
// code inlined next to them the Newton loops of the plain kernels lose registers (216 -> 224 VGPRs single-wave, spills
// in the wave teams) although the region runs once per solve.
template <int V2, int NW, bool DC = false, bool MEM = false, bool CHORD = false, int SPEC = 0, bool POLAR = false>
__device__ __forceinline__ bool solve_instance(const DevPlan& P, const Lds& L, const Opts& o, int lane, int out_br, int n_mod,
                               const double* qg_min, const double* qg_max, int* iters, double* nrm, double* min_piv,
                               int* min_piv_bus, int isl_state = 0, Polar* pol = nullptr, bool warm_started = false) {
  const int wave = threadIdx.x >> 6;
  // isl_state (islanding outages, see island_state): 1 = the caller has de-energised the island
  // (mark_island) and the solve proceeds on the rest; 2 = the cut-off set is not known exactly
  // (several branches out at once, or the first-generation kernel): the Newton matrix would be
  // singular, reported as not converged at once.
  if (isl_state == 2) { *iters = 1; *nrm = __builtin_nan(""); return false; }
  int total = 0;
  bool conv = false;
  // DC start: on the compiled topology only (a modifier changes B' as well; such solves start flat), V2 kernels; it is
  // the first pass of the first Newton loop of the solve
  // (a solve whose modifiers all take a branch out of service — outage, contingency, open line switch — starts from the DC
  //  power flow of the net without them, as pandapower does, dc_mods; other modifiers or a de-energised island: flat)
  // (warm_started: the caller has loaded a solution to start from — a contingency from the base case's voltages,
  //  contingency_start = 0: no DC pass, which would turn that start by theta_dc - va_set)
  const bool dc_first = DC && V2 && o.init == OPFX_INIT_DC && P.lp_dc != nullptr && isl_state == 0 && !warm_started && dc_start_possible(P, L, n_mod);
  for (int outer = 0; outer <= P.npv; ++outer) {
    int it;
    double pv_ = __builtin_nan("");          // (the first-generation kernel does not monitor its pivots)
    int pb_ = -1;
    // (modifiers are folded into the bus rounds of phase A unless an island has been de-energised)
    if (NW > 1) conv = newton2_coop<NW, V2 == 2, MEM, DC && !MEM, CHORD && !MEM, SPEC>(P, L, o, n_mod, &it, nrm, &pv_, &pb_, isl_state == 0, dc_first && outer == 0, pol);
    else conv = V2 ? newton2<V2 == 2, DC && V2 != 0, CHORD && V2 != 0, SPEC, POLAR && NW == 1 && V2 != 0>(P, L, o, lane, n_mod, &it, nrm, &pv_, &pb_, dc_first && outer == 0, pol) : newton(P, L, o, lane, out_br, &it, nrm);
    total += it;
    if (pv_ == pv_ && pv_ < *min_piv) { *min_piv = pv_; *min_piv_bus = pb_; }
    if ((SPEC & SPEC_NO_PV) || !conv || !o.enforce_q_lims || P.npv == 0 || qg_min == nullptr) break;
    // generator reactive output at PV buses: Qg = Qcalc - q_inj(non-generator)
    int changed = 0;
    sec_sync<NW>();
    if (wave == 0) {
      for (int i = lane; i < P.nb; i += WAVE) {
        if (L.bt[i] != BT_PV) continue;
        double ir = 0.0, ii = 0.0;
        for (int e = P.y_ptr[i]; e < P.y_ptr[i + 1]; ++e) {
          const int j = P.y_col[e];
          double g = P.y_g[e], b = P.y_b[e];
          if (!V2 && out_br >= 0) {          // first-generation kernel: single outage by stamp position
            if (e == P.br_pos[out_br * 4 + 0]) { g -= P.br_y[out_br * 8 + 0]; b -= P.br_y[out_br * 8 + 1]; }
            if (e == P.br_pos[out_br * 4 + 1]) { g -= P.br_y[out_br * 8 + 2]; b -= P.br_y[out_br * 8 + 3]; }
            if (e == P.br_pos[out_br * 4 + 2]) { g -= P.br_y[out_br * 8 + 4]; b -= P.br_y[out_br * 8 + 5]; }
            if (e == P.br_pos[out_br * 4 + 3]) { g -= P.br_y[out_br * 8 + 6]; b -= P.br_y[out_br * 8 + 7]; }
          }
          ir += g * L.vr[j] - b * L.vi[j];
          ii += g * L.vi[j] + b * L.vr[j];
        }
        if (V2) mods_row_current(L, n_mod, i, ir, ii);
        const double qc = L.vi[i] * ir - L.vr[i] * ii;
        const double qg = qc - L.qsp[i];
        const double lo = qg_min[i], hi = qg_max[i];
        if (qg > hi) { L.bt[i] = BT_PQ_HI; L.qsp[i] += hi; changed = 1; }
        else if (qg < lo) { L.bt[i] = BT_PQ_LO; L.qsp[i] += lo; changed = 1; }
      }
      changed = wave_any(changed);
      if (NW > 1 && lane == 0) L.acc[0] = changed ? 1.0 : 0.0;
    }
    if (NW > 1) {
      __syncthreads();
      changed = L.acc[0] != 0.0;
      __syncthreads();
    } else {
      sec_sync<NW>();
    }
    if (!changed) break;
  }
  *iters = total;
  return conv;
}

// after convergence: result bank in LDS region R (reuses the LU block storage)
//   [vm nb | va_deg nb | loading nbr | p_ext nref | q_ext nref | q_gen nb]
// `lane` / `stride`: the calling thread's index and the number of threads that share the work (a wavefront,
// or the whole wave team)
template <int V2, int SPEC = 0, bool POLAR = false, bool TEAMPQ = false, bool LANEPQ = POLAR>
__device__ void compute_results(const DevPlan& P, const Lds& L, int lane, int out_br, int n_mod,
                                const double* qg_min, const double* qg_max, double* R, bool physical,
                                bool want_angle, int stride = WAVE, const Polar* pol = nullptr) {
  constexpr bool NOPV = (SPEC & SPEC_NO_PV) != 0, NOMOD = (SPEC & SPEC_NO_MOD) != 0;
  // scheduled P / Q in pol->p / pol->q (see Polar): single-wave PQREG kernels (bus lane + 64 r) and the teams' TPQ kernels
  // (bus tid + NT k; `stride` = NT there)
  constexpr bool PQREG = OPFX_PQREG && NOPV && V2 != 0 && ((POLAR && LANEPQ) || TEAMPQ);      // (LANEPQ: the caller's PQREG — not in the DC-start kernels)
  const int nb = P.nb, nbr = P.nbr, nref = P.nref;
  double* r_vm = R;
  double* r_va = R + nb;
  double* r_ld = R + 2 * nb;
  double* r_pe = r_ld + nbr;
  double* r_qe = r_pe + nref;
  double* r_qg = r_qe + nref;
  const double base = physical ? P.base_mva : 1.0;
  if (POLAR) {
    // |V| and the angle from the polar shadow (see Polar; stride == WAVE here: the bus -> lane map of phase D); the angle
    // brought back into (-pi, pi] as atan2 reports it
#pragma unroll
    for (int r = 0; r < POLAR_R; ++r) {
      const int i = lane + WAVE * r;
      if (i >= nb) continue;
      r_vm[i] = pol->vm[r];
      if (want_angle) {
        const double th = pol->th[r];
        const double ang = fabs(th) <= M_PI ? th : th - (2.0 * M_PI) * rint(th * (0.5 / M_PI));
        r_va[i] = physical ? ang * (180.0 / M_PI) : ang;
      }
      if (PQREG && L.bt[i] == BT_REF) {           // slack power: the parked injection less what was scheduled at the bus
        const int ro = P.ref_ord[i];
        (R + 2 * nb + nbr)[ro] = (L.rhs[i] - pol->p[r]) * base;
        (R + 2 * nb + nbr + nref)[ro] = (L.rq[i] - pol->q[r]) * base;
      }
    }
  }
  if (TEAMPQ && PQREG) {
#pragma unroll
    for (int k = 0; k < TEAM_PQ_R; ++k) {
      const int i = lane + stride * k;
      if (i < nb && L.bt[i] == BT_REF) {
        const int ro = P.ref_ord[i];
        (R + 2 * nb + nbr)[ro] = (L.rhs[i] - pol->p[k]) * base;
        (R + 2 * nb + nbr + nref)[ro] = (L.rq[i] - pol->q[k]) * base;
      }
    }
  }
  for (int i = lane; i < nb; i += stride) {
    if (!POLAR) {
      r_vm[i] = sqrt(L.vr[i] * L.vr[i] + L.vi[i] * L.vi[i]);
      if (want_angle) {
        const double ang = atan2(L.vi[i], L.vr[i]);
        r_va[i] = physical ? ang * (180.0 / M_PI) : ang;
      }
    }
    const int t = L.bt[i];
    double qgen = 0.0;
    if (!NOMOD && t == BT_DEAD) {        // de-energised: no voltage (NaN as in pandapower's res_bus)
      r_vm[i] = __builtin_nan(""); r_va[i] = __builtin_nan("");
    } else if (V2 && t == BT_REF) {
      if (!PQREG) {
        const int ro = P.ref_ord[i];
        r_pe[ro] = (L.rhs[i] - L.psp[i]) * base;          // (V2: rhs/rq are separate arrays)
        r_qe[ro] = (L.rq[i] - L.qsp[i]) * base;
      }
    } else if ((!V2 && t == BT_REF) || (!NOPV && t == BT_PV)) {
      double ir = 0.0, ii = 0.0;
      for (int e = P.y_ptr[i]; e < P.y_ptr[i + 1]; ++e) {
        const int j = P.y_col[e];
        double g = P.y_g[e], b = P.y_b[e];
        if (!V2 && out_br >= 0) {
          if (e == P.br_pos[out_br * 4 + 0]) { g -= P.br_y[out_br * 8 + 0]; b -= P.br_y[out_br * 8 + 1]; }
          if (e == P.br_pos[out_br * 4 + 1]) { g -= P.br_y[out_br * 8 + 2]; b -= P.br_y[out_br * 8 + 3]; }
          if (e == P.br_pos[out_br * 4 + 2]) { g -= P.br_y[out_br * 8 + 4]; b -= P.br_y[out_br * 8 + 5]; }
          if (e == P.br_pos[out_br * 4 + 3]) { g -= P.br_y[out_br * 8 + 6]; b -= P.br_y[out_br * 8 + 7]; }
        }
        ir += g * L.vr[j] - b * L.vi[j];
        ii += g * L.vi[j] + b * L.vr[j];
      }
      if (V2 && !NOMOD) mods_row_current(L, n_mod, i, ir, ii);
      const double pc = L.vr[i] * ir + L.vi[i] * ii, qc = L.vi[i] * ir - L.vr[i] * ii;
      if (t == BT_REF) {
        const int ro = P.ref_ord[i];
        r_pe[ro] = (pc - L.psp[i]) * base;
        r_qe[ro] = (qc - L.qsp[i]) * base;
      } else {
        qgen = (qc - L.qsp[i]) * base;
      }
    } else if (!NOPV && t == BT_PQ_HI) {
      qgen = qg_max[i] * base;
    } else if (!NOPV && t == BT_PQ_LO) {
      qgen = qg_min[i] * base;
    }
    r_qg[i] = qgen;
  }
  // (the stamps and factors of the NEXT round of branches are requested before this round is worked: the rounds were a chain
  //  of memory round trips — 5 % of a 144-bus step, found with -DOPFX_DUP — and are one round trip plus arithmetic now)
  struct BrRow { double y[8]; int f, t; double kf, kt; };
  auto ld_row = [&](int k) {
    const unsigned kc = (unsigned)(k < nbr ? k : nbr - 1);
    BrRow r;
#pragma unroll
    for (int q = 0; q < 8; ++q) r.y[q] = ld_at(P.br_y, 8 * kc + q);
    r.f = ld_at(P.br_f, kc); r.t = ld_at(P.br_t, kc); r.kf = ld_at(P.br_kf, kc); r.kt = ld_at(P.br_kt, kc);
    return r;
  };
  BrRow nxt_row{};
  if (nbr > 0) nxt_row = ld_row(lane);
  for (int k = lane; k < nbr; k += stride) {
    double ld = 0.0;
    const BrRow row = nxt_row;
    nxt_row = ld_row(k + stride);
    double y[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) y[q] = row.y[q];
    bool removed = !V2 && k == out_br;
    if (V2 && !NOMOD) for (int m = 0; m < n_mod; ++m) {
      const int* id = mod_ids(L, m);
      if (id[6] != k) continue;
      const double* dy = mod_dy(L, m);
#pragma unroll
      for (int q = 0; q < 8; ++q) y[q] += dy[q];
      removed = removed || id[7] == MOD_REMOVED;
    }
    // |V| of a de-energised bus is NaN and so is every current computed with it (pandapower divides the
    // branch's apparent power by it), in service or not
    const int f = row.f, t = row.t;
    if (!NOMOD && (L.bt[f] == BT_DEAD || L.bt[t] == BT_DEAD)) { r_ld[k] = __builtin_nan(""); continue; }
    if (!removed) {
      const double vfr = L.vr[f], vfi = L.vi[f], vtr = L.vr[t], vti = L.vi[t];
      const double ifr = y[0] * vfr - y[1] * vfi + y[2] * vtr - y[3] * vti;
      const double ifi = y[0] * vfi + y[1] * vfr + y[2] * vti + y[3] * vtr;
      const double itr = y[4] * vfr - y[5] * vfi + y[6] * vtr - y[7] * vti;
      const double iti = y[4] * vfi + y[5] * vfr + y[6] * vti + y[7] * vtr;
      // (one square root for the two ends: kf, kt >= 0, so max(|If| kf, |It| kt) = sqrt(max(|If|^2 kf^2, |It|^2 kt^2)))
      ld = sqrt(fmax((ifr * ifr + ifi * ifi) * (row.kf * row.kf), (itr * itr + iti * iti) * (row.kt * row.kt)));
    }
    r_ld[k] = ld;
  }
}

// accumulator doubles of the env kernels: five per constraint group (at least 8) + two for the work queue's hand-over
__host__ __device__ constexpr int env_nacc(int nc) { return (5 * nc > 8 ? 5 * nc : 8) + 2; }
template <int V2, bool MEM = false>
__device__ __forceinline__ Lds carve(const DevPlan& P, int na, int nres, double* base, int nacc, int nmod) {
  Lds L;
  const int nb = P.nb;
  const int nbe = (nb + 1) & ~1;               // even count keeps every array 16-byte aligned
  L.vr = base; L.vi = L.vr + nbe;
  double* nxt = L.vi + nbe;
  if (V2) { L.vm = nullptr; L.va = nullptr; }
  else { L.vm = nxt; L.va = L.vm + nbe; nxt = L.va + nbe; }
  if (V2) {
    // scheduled P/Q are touched by the lane that owns the bus row only (phase A, q-limits,
    // results): they live in a per-workgroup row of global memory (L2-resident, read one round
    // ahead with the descriptors) and free 16*nb bytes of LDS — the resource that sets how many
    // instances a CU holds
    L.psp = P.pq + (size_t)blockIdx.x * 2 * nbe; L.qsp = L.psp + nbe; L.rhs = nxt;
  } else { L.psp = nxt; L.qsp = L.psp + nbe; L.rhs = L.qsp + nbe; }
  L.rq = L.rhs + nbe;
  L.stage = L.rhs + 2 * nbe;
  L.blk = MEM ? P.blk_mem + (size_t)blockIdx.x * P.blk_mem_stride : L.stage;
  L.bs = (P.nblk + 1) & ~1;
  L.nfull = V2 ? P.nfull : P.nblk;
  const int nfs = (L.nfull + 1) & ~1;
  L.o2 = 2 * L.bs; L.o3 = 2 * L.bs + nfs;
  const int nval = V2 ? 2 * L.bs + 2 * nfs : 4 * L.bs;      // (first-generation kernel: 32-byte records)
  const int nblk_d = (!MEM && nval > nres) ? nval : nres;
  L.sp = L.stage + ((nblk_d + 1) & ~1);
  L.acc = L.sp + na;
  L.mod = L.acc + nacc;
  L.bt = reinterpret_cast<unsigned char*>(L.mod + MOD_DOUBLES * nmod);
  L.dg = reinterpret_cast<unsigned short*>(L.bt + ((nb + 1) & ~1));
  {   // the tail table is read with 16-byte LDS loads
    const size_t off = (size_t)(reinterpret_cast<char*>(L.dg + ((nb + 1) & ~1)) - reinterpret_cast<char*>(base));
    L.tl = reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(base) + ((off + 15) & ~(size_t)15));
  }
  return L;
}

// The next instance of a workgroup of the solve / step kernels: gridDim.x + the queue value thread 0 fetched after the solve
// (handed to the other wavefronts of a team through LDS), or the fixed stride; doubles as the end-of-instance barrier.
template <int NW>
__device__ __forceinline__ long long next_instance(double* slot, long long b, bool queued, int nxt_v) {
  if (!queued) { blk_sync<NW>(); return b + gridDim.x; }
  int* const queue_slot = reinterpret_cast<int*>(slot);
  if (threadIdx.x == 0) *queue_slot = nxt_v;
  blk_sync<NW>();                                     // (all wavefronts are through with the instance; the value is there)
  const long long n = (long long)gridDim.x + __builtin_amdgcn_readfirstlane(*queue_slot);
  blk_sync<NW>();                                     // (read by everyone before thread 0 writes the next one)
  return n;
}

// ---------------------------------------------------------------------------
// pure power flow kernel (opfx_solve)
// ---------------------------------------------------------------------------
// Two wavefronts per SIMD (256 VGPRs each) is what LDS capacity allows anyway: 8 single-wave instances or 2 teams of
// four per CU.  -DOPFX_MIN_WAVES_PER_SIMD=3 caps the kernels at 168 VGPRs: the build of the negative result in
// profiles/r02_ab_three_waves_per_simd.txt (it spills, and the third wave slot stays empty).
#ifndef OPFX_MIN_WAVES_PER_SIMD
#define OPFX_MIN_WAVES_PER_SIMD 2
#endif
template <int V2, int NW, bool DC = false, bool MEM = false, bool CHORD = false>
__global__ __launch_bounds__(WAVE * NW, OPFX_MIN_WAVES_PER_SIMD) void k_solve(const DevPlan P, SolveIO io, Opts o, long long B) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nres_ = 3 * P.nb + P.nbr + 2 * P.nref;
  const Lds L = carve<V2, MEM>(P, 0, nres_, smem, 8, 1);
  if (V2) {
    for (int i = threadIdx.x; i < P.nb; i += blockDim.x) L.dg[i] = (unsigned short)P.diag_blk[i];
    for (int i = threadIdx.x; i < P.tail_n; i += blockDim.x) L.tl[i] = P.tail_ids[i];
    blk_sync<NW>();
  }
  int turn = NW == 1 ? (int)(__builtin_amdgcn_s_getreg(GETREG_HW_ID) & 1u) : 0;     // (HW_ID: the wave slot on its SIMD; see k_step)
  for (long long b = blockIdx.x; b < B;) {
    if (NW == 1) { if (turn & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); ++turn; }
    if (wave == 0) {
      for (int i = lane; i < P.nb; i += WAVE) {
        L.psp[i] = io.p_inj[b * P.nb + i];
        L.qsp[i] = io.q_inj[b * P.nb + i];
        L.bt[i] = BT_PQ;
      }
      init_voltage<V2>(P, L, lane, io.qg_min, io.qg_max, o.enforce_q_lims == 1);
    }
    const int out_br = io.outage ? io.outage[b] : -1;
    const int n_mod = (V2 && out_br >= 0) ? 1 : 0;
    if (n_mod && wave == 0) mod_set(P, L, lane, 0, out_br, 0.0, true, 0);
    const int isl = island_state(V2 != 0, out_br >= 0 ? 1 : 0, out_br >= 0 && P.br_island[out_br]);
    if (isl == 1 && wave == 0) mark_island(P, L, lane, out_br, io.qg_min, io.qg_max);
    blk_sync<NW>();
    int iters; double nrm;
    double min_piv = V2 ? 1.0 : __builtin_nan("");
    int min_piv_bus = -1;
    const bool conv = solve_instance<V2, NW, DC, MEM, CHORD>(P, L, o, lane, out_br, n_mod, io.qg_min, io.qg_max, &iters, &nrm, &min_piv, &min_piv_bus, isl);
    blk_sync<NW>();
    int nxt_v = 0;                            // (work queue, see k_step)
    if (io.queued && threadIdx.x == 0) nxt_v = __hip_atomic_fetch_add(P.queue, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (wave == 0) {
      double* R = L.stage;
      compute_results<V2>(P, L, lane, out_br, n_mod, io.qg_min, io.qg_max, R, false, io.va != nullptr);
      sec_sync<NW>();
      const int nb = P.nb, nbr = P.nbr, nref = P.nref;
      if (io.vm) for (int i = lane; i < nb; i += WAVE) io.vm[b * nb + i] = R[i];
      if (io.va) for (int i = lane; i < nb; i += WAVE) io.va[b * nb + i] = R[nb + i];
      if (io.loading) for (int k = lane; k < nbr; k += WAVE) io.loading[b * nbr + k] = R[2 * nb + k];
      if (io.s_ref) for (int r = lane; r < nref; r += WAVE) {
        io.s_ref[(b * nref + r) * 2] = R[2 * nb + nbr + r];
        io.s_ref[(b * nref + r) * 2 + 1] = R[2 * nb + nbr + nref + r];
      }
      if (io.q_gen) for (int i = lane; i < nb; i += WAVE) io.q_gen[b * nb + i] = R[2 * nb + nbr + 2 * nref + i];
      if (lane == 0) {
        if (io.converged) io.converged[b] = conv ? 1 : 0;
        if (io.iterations) io.iterations[b] = iters;
        if (io.max_mismatch) io.max_mismatch[b] = nrm;
        if (io.min_pivot) io.min_pivot[b] = min_piv;
        if (io.min_pivot_bus) io.min_pivot_bus[b] = min_piv_bus;
      }
    }
    b = next_instance<NW>(L.acc + 6, b, io.queued != 0, nxt_v);
  }
}

// ---------------------------------------------------------------------------
// fused env.step kernel (opfx_step)
// ---------------------------------------------------------------------------

// value of a table cell during a step: src >= 0 -> staged table row, src < 0 -> set-point of
// action ~src (a column written by an action is read from the set-point), NOSRC -> absent (0)
constexpr int NOSRC = 0x7FFFFFFF;
__device__ __forceinline__ double src_val(const double* xs, const double* sp, int src) {
  return src == NOSRC ? 0.0 : (src >= 0 ? xs[src] : sp[~src]);
}

// the same with the table row in GLOBAL memory (after the solve the staged copy is gone) and the set-points in LDS: explicit
// address spaces, so that no generic pointer is selected between the two (a flat load, and in some instantiations this
// ROCm's backend dies on the LDS-to-flat cast of the uniform pointer: "Illegal instruction detected: V_CMP_NE_U32_e32 0,
// $src_shared_base" — the stamps build did, round 5)
__device__ __forceinline__ double src_val_g(const double* xr, const double* sp, int src) {
  double v = 0.0;
  if (src != NOSRC) {
    if (src >= 0) v = ld_at(xr, (unsigned)src);
    else v = ((const __attribute__((address_space(3))) double*)sp)[~src];
  }
  return v;
}

__device__ __forceinline__ double sgn(double v) { return (v > 0.0) - (v < 0.0); }

__device__ __forceinline__ double u2d(unsigned lo, unsigned hi) {
  return __longlong_as_double(((long long)hi << 32) | lo);
}

// ---------------------------------------------------------------------------
// reset kernel: SimBench state sampling (opf_env.py:317-372) + `_sampling` tails
// ---------------------------------------------------------------------------
// What a reset executes, compiled at opfx_env_set_reset time into CHUNKS of up to 64 consecutive elements whose
// structure is the same for every lane — a run of columns of one profile table, or 64 elements of one vector op —
// so that everything but the per-element data is WAVE-UNIFORM: a chunk's descriptor (32 bytes) comes through the scalar
// cache into SGPRs, the op code is a scalar branch (only the code that runs is executed, and once), destination /
// source / constant addresses are a scalar base + the lane, and the per-element data (types, peaks, limits, constants)
// are coalesced loads.  The element-list form of this (round 3, first half: one descriptor record per element and lane)
// spent 1 687 vector instructions and 217 vector loads on a 1 082-column row, 100 per 64 outputs, nearly all of it
// decoding per-lane descriptors that are equal across the rows; the kernel was bound by exactly that.
//   * profile chunks: columns [e0, e0 + n) of the list of all profile columns, all of one table;
//   * op chunks by STAGE: an op is cut into chunks; ops that do not depend on one another share a stage (one pass, one
//     LDS fence), e.g. VoltageControl's tail {max_p, min_p, q := 0} | {max_q} | {min_q}; within a stage the chunks whose
//     code needs a long function (inverse normal CDF, truncated normal, in-kernel normal draws) come last and run in a
//     loop of their own, so that the common loop stays small and is unrolled over several chunks;
//   * the observation as one element list (source kind | index).
constexpr int MAX_TABLES = 8, MAX_STAGES = 12;
typedef int i32x8 __attribute__((ext_vector_type(8)));
// profile chunk words: 0-1 address of the table's relative profiles ([n_steps, n_types]), 2 first column number e0,
//   3 columns n (1..64), 4 types per step, 5 steps of the table, 6 table number
// op chunk words: 0 code | mode mask << 8 | OCH_* flags << 16, 1 elements n (1..64), 2 first destination slot,
//   3 first source slot or first draw number, 4-6 offsets of the element constants k0 / k1 / k2 in `consts` (-1: none, 0.0)
constexpr int OCH_READS_ROW = 1 << 16, OCH_LONG = 2 << 16;
struct DevReset {
  int n_tables, n_uniform, n_normal, n_noise, nx, init_off, has_mode;
  int skip_template;                       // every slot of the row is written by a profile column or an op (and no per-instance modes)
  // profile columns, all tables: per-column data by column number; chunks of columns of one table
  int n_tcols, n_pch;
  const i32x8* pch;
  const int *tc_typ, *tc_slot;             // type, destination slot
  const double *tc_peak, *tc_lo, *tc_hi;
  // op chunks by stage
  int n_stages;
  unsigned st_barrier;                     // bit s: stage s starts with a workgroup barrier (teams of wavefronts)
  int st_ptr[2 * MAX_STAGES + 1];          // stage s = chunks [st_ptr[2s], st_ptr[2s+2]), the long ones from st_ptr[2s+1] on
  const i32x8* och;
  const double* consts;                    // row template at init_off; element constants of the ops
  // observation elements (environments whose observation needs no power flow)
  int n_oel;
  const int* oe_src;                       // position in the LDS image of the instance (table row, then the action set-points), -1: NaN (result entry)
};

struct ResetIO {
  const int* step_idx;
  const double *noise, *interp, *uniform, *normal;
  double normal_noise_factor;
  double* x;
  const int* mode;
  const double* action;      // optional: initial action [B,na] (reset without power flow)
  double* obs;               // optional: table observation [B,nobs]
  int keep_state;            // start from the instance's current row instead of the template
  const int* step_pool;      // optional: draw the step in the kernel (opfx_reset_io::step_pool)
  int n_step_pool;
  unsigned long long rng_seed;
  int* step_out;
  unsigned long long* stamps;   // developer probe (stamps build): per-phase cycle sums of wavefront 0 of workgroup 0
};

// Standard normal truncated to [a, b], by inverse CDF of a uniform draw u — what scipy.stats.truncnorm.ppf(u, a, b)
// computes, in log space so that bounds far out in a tail work (opf_env.py:306-309 hands scipy the raw MW bounds as
// STANDARDISED ones, defect D14: a unit between 50 and 300 MW is "50 to 300 sigma"; Phi(50) == 1.0 in double).
//   log Phi(t): erfcx form in the lower tail;  a > 0: mirrored problem (-b, -a), 1 - u, result negated;
//   log P = log Phi(b) + log(r + u (1 - r)),  r = Phi(a) / Phi(b);  z = (log Phi)^-1(log P): normcdfinv where P is
//   representable (through 1 - P next to 1), Newton on log Phi below that (d/dz log Phi = phi / Phi = sqrt(2/pi) / erfcx(-z / sqrt 2)).
__device__ __forceinline__ double log_ndtr(double t) {
  const double s = t * 0.70710678118654752440;
  return t < 0.0 ? log(0.5 * erfcx(-s)) - s * s : log1p(-0.5 * erfc(s));
}
__device__ __forceinline__ double ndtri_log(double lp) {
  if (lp > -0.69314718055994531) return -normcdfinv(-expm1(lp));
  if (lp > -600.0) return normcdfinv(exp(lp));
  double z = -sqrt(-2.0 * lp);
  for (int k = 0; k < 6; ++k) {
    const double f = log_ndtr(z) - lp;
    const double d = 0.79788456080286535588 / erfcx(-z * 0.70710678118654752440);
    z -= f / d;
  }
  return z;
}
__device__ __forceinline__ double truncnorm_ppf(double u, double a, double b) {
  const bool flip = a > 0.0;
  const double lo = flip ? -b : a, hi = flip ? -a : b;
  const double q = flip ? 1.0 - u : u;
  const double lhi = log_ndtr(hi);
  const double r = exp(log_ndtr(lo) - lhi);
  double z = ndtri_log(lhi + log(r + q * (1.0 - r)));
  z = fmin(fmax(z, lo), hi);
  return flip ? -z : z;
}

// Counter-based random numbers for draws made inside the kernels (splitmix64 finaliser): value j of instance b under a
// per-reset seed; nothing to store, reproducible from the seed the caller drew from ITS generator.
__device__ __forceinline__ unsigned long long mix64(unsigned long long h) {
  h = (h ^ (h >> 30)) * 0xBF58476D1CE4E5B9ull;
  h = (h ^ (h >> 27)) * 0x94D049BB133111EBull;
  return h ^ (h >> 31);
}
__device__ __forceinline__ unsigned long long draw_bits(unsigned long long seed, long long b, unsigned stream) {
  return mix64(mix64(seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(b + 1)) + 0xD6E8FEB86659FD93ull * (unsigned long long)(stream + 1));
}
__device__ __forceinline__ double draw_uniform(unsigned long long seed, long long b, unsigned stream) {     // [0, 1)
  return (double)(draw_bits(seed, b, stream) >> 11) * (1.0 / 9007199254740992.0);
}
__device__ __forceinline__ double draw_normal(unsigned long long seed, long long b, unsigned stream) {      // Box-Muller
  const double u1 = 1.0 - draw_uniform(seed, b, 2u * stream + 0x40000000u), u2 = draw_uniform(seed, b, 2u * stream + 0x40000001u);
  return sqrt(-2.0 * log(u1)) * cospi(2.0 * u2);
}

#ifdef OPFX_ENABLE_STAMPS
#define RSTAMP(slot)                                                                          \
  do {                                                                                        \
    if (io.stamps && blockIdx.x == 0 && threadIdx.x < 64) {                                   \
      const unsigned long long now__ = __builtin_readcyclecounter();                          \
      if (threadIdx.x == 0) io.stamps[slot] += now__ - rt_last__;                              \
      rt_last__ = __builtin_readcyclecounter();                                               \
    }                                                                                         \
  } while (0)
#define RSTAMPD(slot) do { __builtin_amdgcn_s_waitcnt(0); RSTAMP(slot); } while (0)
#else
#define RSTAMP(slot) do { } while (0)
#define RSTAMPD(slot) do { } while (0)
#endif

// One vector-op element (include/opfx.h OPFX_OP_*): rv = the source value of the row (or 0), dr = its draw (or 0).
// `code` is wave-uniform (a chunk holds elements of one op).  LONG: with the codes that need a long function.
template <bool LONG>
__device__ __forceinline__ double op_value(int code, double rv, double dr, double k0, double k1, double k2) {
  switch (code) {
    case OPFX_OP_SET_CONST: return k0;
    case OPFX_OP_AFFINE: return rv * k0 + k1;
    case OPFX_OP_SQRT_DIFF: return sqrt(k0 * k0 - rv * rv);
    case OPFX_OP_NEG: return -rv;
    case OPFX_OP_UNIFORM: return (k0 + dr * (k1 - k0)) / k2;
    case OPFX_OP_NORMAL: return k0 + k1 * dr;
    case OPFX_OP_CLIP: return fmin(fmax(rv, k0), k1);
    case OPFX_OP_NORMINV: return LONG ? k0 + k1 * normcdfinv(rv) : 0.0;
    case OPFX_OP_TRUNCNORM: return LONG ? truncnorm_ppf(rv, k0, k1) : 0.0;
    default: return rv / k0;                 // OPFX_OP_DIV
  }
}

// a chunk descriptor through the scalar cache (`c` is wave-uniform)
__device__ __forceinline__ i32x8 ld_chunk(const i32x8* base, int c) { return as_const(base)[c]; }

// rows move between memory and the LDS two columns per lane where the row allows it (even column count: every row
// of the batch starts on a 16-byte boundary), UC chunks per round trip
typedef double f64x2 __attribute__((ext_vector_type(2)));
template <int UC>
__device__ __forceinline__ void row_copy_in(double* row, const double* src, int nx, int lane) {
  if ((nx & 1) == 0 && (reinterpret_cast<size_t>(src) & 15) == 0) {
    const int n2 = nx >> 1;
    const f64x2* s2 = reinterpret_cast<const f64x2*>(src);
    f64x2* r2 = reinterpret_cast<f64x2*>(row);
    for (int j0 = lane; j0 < n2; j0 += 64 * UC) {
      f64x2 t[UC];
#pragma unroll
      for (int u = 0; u < UC; ++u) { const int j = j0 + 64 * u; t[u] = ld_at(s2, (unsigned)(j < n2 ? j : n2 - 1)); }
#pragma unroll
      for (int u = 0; u < UC; ++u) { const int j = j0 + 64 * u; if (j < n2) r2[j] = t[u]; }
    }
  } else {
    for (int j0 = lane; j0 < nx; j0 += 64 * UC) {
      double t[UC];
#pragma unroll
      for (int u = 0; u < UC; ++u) { const int j = j0 + 64 * u; t[u] = ld_at(src, (unsigned)(j < nx ? j : nx - 1)); }
#pragma unroll
      for (int u = 0; u < UC; ++u) { const int j = j0 + 64 * u; if (j < nx) row[j] = t[u]; }
    }
  }
}
template <int UC>
__device__ __forceinline__ void row_copy_out(double* dst, const double* row, int nx, int lane) {
  if ((nx & 1) == 0 && (reinterpret_cast<size_t>(dst) & 15) == 0) {
    const int n2 = nx >> 1;
    f64x2* d2 = reinterpret_cast<f64x2*>(dst);
    const f64x2* r2 = reinterpret_cast<const f64x2*>(row);
    for (int j0 = lane; j0 < n2; j0 += 64 * UC) {
      f64x2 t[UC];
#pragma unroll
      for (int u = 0; u < UC; ++u) { const int j = j0 + 64 * u; t[u] = r2[j < n2 ? j : n2 - 1]; }
#pragma unroll
      for (int u = 0; u < UC; ++u) { const int j = j0 + 64 * u; if (j < n2) st_at(d2, (unsigned)j, t[u]); }
    }
  } else {
    for (int j0 = lane; j0 < nx; j0 += 64 * UC) {
      double t[UC];
#pragma unroll
      for (int u = 0; u < UC; ++u) { const int j = j0 + 64 * u; t[u] = row[j < nx ? j : nx - 1]; }
#pragma unroll
      for (int u = 0; u < UC; ++u) { const int j = j0 + 64 * u; if (j < nx) st_at(dst, (unsigned)j, t[u]); }
    }
  }
}

// The reset of NR instances by one workgroup of NR wavefronts.  The rows are built in LDS (`rows`: NR x (nx + na)
// doubles) — template (unless every column is written anyway) -> profile values -> vector-op stages -> optionally the
// initial action and the table observation (opf_env.py:201-207,218) — and leave with one coalesced store each: the
// intermediate values never make a round trip through memory.
// The CHUNKS of a pass are dealt over the wavefronts and a wavefront applies its chunk to ALL NR rows: the chunk's
// descriptor and per-element data (types, peaks, limits, op constants — equal for every row) are fetched once per NR
// rows, and a row's chain of dependent round trips shrinks with it (one wavefront per row walked 9 profile chunks and 15
// op chunks of the 144-bus VoltageControl row in 12 round trips; a team of four walks them in 7).  Passes are separated
// by LDS-only workgroup barriers.  Template, initial action, observation and the final store are per row: wavefront w
// owns row w.
// (Round 3 also ran the reset in the epilogue of the step kernel — bit-identical rows, 404.6 us per launch against
// 270.3 + 56.8 us for two launches: the step kernel's throughput is waves / latency per instance and the reset's chain of
// round trips adds its whole latency to every instance — and with its descriptors in an LDS image per persistent
// workgroup — 64.9 vs 56.8 us: 13 rows per CU instead of 16.  profiles/r03_reset_experiments.txt.  Both removed.)
// U: chunks of one wavefront per round trip; UC: 64-element pieces per round trip of the plain copies.
template <int NR, int U, int UC, bool FULL>
__device__ __forceinline__ void reset_rows(const DevReset& R, const DevEnv* __restrict__ Ep, const ResetIO& io, long long b0, long long B,
                                           int lane, int wib, double* const rows, int row_doubles) {
  // Wave-uniform decisions below are kept out of the per-(chunk, row) code where they would be scalar BRANCHES (a taken
  // branch costs a wavefront ~20 cycles, and a first version with a decision tree per chunk and row spent two thirds
  // of its cycles in them): dead chunks / rows / ops read valid addresses and store under an empty lane mask.
  const double NaN = __builtin_nan("");
#ifdef OPFX_ENABLE_STAMPS
  unsigned long long rt_last__ = __builtin_readcyclecounter();
#endif
  const int nxe = (R.nx + 1) & ~1;
  double* const row = rows + (size_t)wib * row_doubles;                  // this wavefront's own row
  double* const sp = row + nxe;
  int* const shared = reinterpret_cast<int*>(rows + (size_t)NR * row_doubles);   // [NR] time steps, [NR] data sources
  const long long bw = b0 + wib;
  const bool mine = bw < B;
  const long long bwc = mine ? bw : B - 1;
  double* xr = io.x + bwc * R.nx;
  int step[NR], mode[NR];
  bool live[NR];
  // What does not depend on the rows is requested AHEAD of where it is used: the descriptors and per-column data of this
  // wavefront's first profile chunks before the time steps are known, and those of the first chunks of op stage s + 1
  // while stage s computes — a row's chain of dependent round trips is what the kernel's time consists of.
  struct PfRegs { i32x8 ch[U]; int typ[U], slot[U]; double peak[U], lo[U], hi[U]; };
  auto pf_load = [&](int c0, PfRegs& q) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + NR * u;
      q.ch[u] = ld_chunk(R.pch, c < R.n_pch ? c : (R.n_pch > 0 ? R.n_pch - 1 : 0));
      q.ch[u][3] = c < R.n_pch ? q.ch[u][3] : 0;                            // (past the end: no live lane)
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int n = q.ch[u][3], e = q.ch[u][2] + (lane < n ? lane : (n > 0 ? n - 1 : 0));
      q.typ[u] = 0; q.slot[u] = 0; q.peak[u] = 0.0; q.lo[u] = 0.0; q.hi[u] = 0.0;
      if (u > 0 && n == 0) continue;                       // (a slot past the end costs one scalar branch, not its instructions)
      q.typ[u] = ld_at(R.tc_typ, (unsigned)e); q.slot[u] = ld_at(R.tc_slot, (unsigned)e);
      q.peak[u] = ld_at(R.tc_peak, (unsigned)e); q.lo[u] = ld_at(R.tc_lo, (unsigned)e); q.hi[u] = ld_at(R.tc_hi, (unsigned)e);
    }
  };
  PfRegs pf;
  if (R.n_pch > 0) pf_load(wib, pf);
  {
    int my_step;
    if (io.step_pool) {
      // counter-based draw: entry floor(h n / 2^64) of the pool (a multiply-high instead of a 64-bit division), uniform
      // up to a bias of n / 2^64
      const unsigned long long h = mix64(io.rng_seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(bwc + 1));
      my_step = as_const(io.step_pool)[(int)__umul64hi(h, (unsigned long long)io.n_step_pool)];
      if (lane == 0 && mine && io.step_out) io.step_out[bw] = my_step;
    } else {
      my_step = as_const(io.step_idx)[bwc];
    }
    const int my_mode = (FULL && io.mode && R.has_mode) ? as_const(io.mode)[bwc] : -1;    // data source of this instance ('mixed'), -1: none
    if (NR == 1) { step[0] = my_step; mode[0] = my_mode; live[0] = mine; }
    else if (lane == 0) { shared[wib] = my_step; shared[NR + wib] = my_mode; }
  }
  RSTAMP(0);
  // ---- the row template (or the instance's own row: keep_state) ---------------------------------------------------
  if ((!R.skip_template || io.keep_state) && mine) {
    const double* src = (R.init_off >= 0 && !io.keep_state) ? R.consts + R.init_off : xr;
    row_copy_in<UC>(row, src, R.nx, lane);
  }
  if (NR > 1) {
    lds_barrier();
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      step[r] = __builtin_amdgcn_readfirstlane(shared[r]);
      mode[r] = __builtin_amdgcn_readfirstlane(shared[NR + r]);
      live[r] = b0 + r < B;
    }
  }
  RSTAMP(1);
  // ---- profile values of every table (opf_env.py:339-372) -------------------------------------------------------
  // FULL: the kernel with interpolation between time steps, noise on the profile values or per-instance data sources (all
  // rare: kernels of their own, so that the plain one neither carries their registers nor evaluates them under a mask)
  const bool any_itp = FULL && io.interp != nullptr, any_noise = FULL && io.noise != nullptr;
  for (int c0 = wib; c0 < R.n_pch; c0 += NR * U) {
    if (c0 != wib) pf_load(c0, pf);
    const i32x8 (&ch)[U] = pf.ch;
    const int (&typ)[U] = pf.typ; const int (&slot)[U] = pf.slot;
    const double (&peak)[U] = pf.peak; const double (&lo)[U] = pf.lo; const double (&hi)[U] = pf.hi;
    double r0[U][NR], r1[FULL ? U : 1][FULL ? NR : 1], rr[FULL ? U : 1][FULL ? NR : 1];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const double* rel = reinterpret_cast<const double*>(((unsigned long long)(unsigned)ch[u][1] << 32) | (unsigned)ch[u][0]);
#pragma unroll
      for (int r = 0; r < NR; ++r) r0[u][r] = 0.0;
      if (u > 0 && ch[u][3] == 0) continue;
#pragma unroll
      for (int r = 0; r < NR; ++r) r0[u][r] = ld_at(rel + (long long)step[r] * ch[u][4], (unsigned)typ[u]);
    }
    if (FULL && any_itp) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const double* rel = reinterpret_cast<const double*>(((unsigned long long)(unsigned)ch[u][1] << 32) | (unsigned)ch[u][0]);
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const bool itp = step[r] < ch[u][5] - 1;                                   // :345 (the last step has no successor)
          r1[FULL ? u : 0][FULL ? r : 0] = ld_at(rel + (long long)step[r] * ch[u][4], (unsigned)((itp ? ch[u][4] : 0) + typ[u]));
          rr[FULL ? u : 0][FULL ? r : 0] = as_const(io.interp)[(live[r] ? b0 + r : B - 1) * R.n_tables + ch[u][6]];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int n = ch[u][3], e = ch[u][2] + (lane < n ? lane : (n > 0 ? n - 1 : 0));
      if (u > 0 && n == 0) continue;
      double v[NR];
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        v[r] = r0[u][r] * peak[u];                                                 // :343
        if (FULL && any_itp) {                                                     // :347-349
          const double rr_ = rr[FULL ? u : 0][FULL ? r : 0];
          const double w = v[r] * rr_ + (r1[FULL ? u : 0][FULL ? r : 0] * peak[u]) * (1.0 - rr_);
          v[r] = step[r] < ch[u][5] - 1 ? w : v[r];
        }
        if (FULL && any_noise) {
          const double nz = ld_at(io.noise + (live[r] ? b0 + r : B - 1) * R.n_noise, (unsigned)e);   // (noise columns are numbered like the list)
          if (io.normal_noise_factor > 0.0) v[r] = v[r] + fabs(v[r]) * io.normal_noise_factor * nz;   // :359-360
          else v[r] = v[r] * nz;                                                   // :354-356
        }
        v[r] = fmin(fmax(v[r], lo[u]), hi[u]);                                     // :364-369
      }
      // (rows past the end of the batch are computed like the last one and never leave the LDS)
      if (FULL) {
#pragma unroll
        for (int r = 0; r < NR; ++r) if (lane < (mode[r] <= 0 ? n : 0)) rows[(size_t)r * row_doubles + slot[u]] = v[r];
      } else if (lane < n) {
#pragma unroll
        for (int r = 0; r < NR; ++r) rows[(size_t)r * row_doubles + slot[u]] = v[r];           // :371-372
      }
    }
  }
  RSTAMP(2);
  // ---- the `_sampling` tail: stages of mutually independent op chunks ----------------------------------------------
  struct OpRegs { i32x8 ch[U]; double k0[U], k1[U], k2[U]; };
  auto op_load = [&](int c0, int sl, OpRegs& q) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + NR * u;
      q.ch[u] = ld_chunk(R.och, c < sl ? c : (sl > 0 ? sl - 1 : 0));
      q.ch[u][1] = c < sl ? q.ch[u][1] : 0;                                 // (past the end: no live lane)
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int n = q.ch[u][1], lc = lane < n ? lane : (n > 0 ? n - 1 : 0);
      q.k0[u] = 0.0; q.k1[u] = 0.0; q.k2[u] = 0.0;
      if (u > 0 && n == 0) continue;                       // (a slot past the end costs one scalar branch, not its instructions)
      // (an op without a constant points at a block of zeros, opfx_env_set_reset)
      q.k0[u] = ld_at(R.consts, (unsigned)(q.ch[u][4] + lc));
      q.k1[u] = ld_at(R.consts, (unsigned)(q.ch[u][5] + lc));
      q.k2[u] = ld_at(R.consts, (unsigned)(q.ch[u][6] + lc));
    }
  };
  auto op_apply = [&](const OpRegs& q) {
    double rv[U][NR];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int n = q.ch[u][1], lc = lane < n ? lane : (n > 0 ? n - 1 : 0);
      const int src = (q.ch[u][0] & OCH_READS_ROW) ? q.ch[u][3] + lc : 0;
#pragma unroll
      for (int r = 0; r < NR; ++r) rv[u][r] = 0.0;
      if (u > 0 && n == 0) continue;
#pragma unroll
      for (int r = 0; r < NR; ++r) rv[u][r] = rows[(size_t)r * row_doubles + src];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int n = q.ch[u][1], code = q.ch[u][0] & 0xFF, lc = lane < n ? lane : (n > 0 ? n - 1 : 0);
      if (u > 0 && n == 0) continue;
      const double k0 = q.k0[u], k1 = q.k1[u], k2 = q.k2[u];
      double v[NR];
      // ONE dispatch on the op code per chunk, the rows inside it (OP_CASE: an empty volatile asm keeps each case a
      // branch target whatever the optimiser thinks of the cost of the other cases' division and square root)
#define OP_CASE() asm volatile("" ::: "memory")
      switch (code) {
        case OPFX_OP_AFFINE:
          OP_CASE();
#pragma unroll
          for (int r = 0; r < NR; ++r) v[r] = rv[u][r] * k0 + k1;
          break;
        case OPFX_OP_SET_CONST:
          OP_CASE();
#pragma unroll
          for (int r = 0; r < NR; ++r) v[r] = k0;
          break;
        case OPFX_OP_UNIFORM:
          OP_CASE();
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            // the element's draw: from the caller's array, or made here from the per-reset seed
            const long long b = live[r] ? b0 + r : B - 1;
            const double dr = io.uniform ? ld_at(io.uniform + b * R.n_uniform, (unsigned)(q.ch[u][3] + lc))
                                         : draw_uniform(io.rng_seed, b, (unsigned)(q.ch[u][3] + lc));
            v[r] = op_value<false>(OPFX_OP_UNIFORM, 0.0, dr, k0, k1, k2);
          }
          break;
        case OPFX_OP_SQRT_DIFF:
          OP_CASE();
#pragma unroll
          for (int r = 0; r < NR; ++r) v[r] = op_value<false>(OPFX_OP_SQRT_DIFF, rv[u][r], 0.0, k0, k1, k2);
          break;
        case OPFX_OP_NEG:
          OP_CASE();
#pragma unroll
          for (int r = 0; r < NR; ++r) v[r] = -rv[u][r];
          break;
        case OPFX_OP_CLIP:
          OP_CASE();
#pragma unroll
          for (int r = 0; r < NR; ++r) v[r] = op_value<false>(OPFX_OP_CLIP, rv[u][r], 0.0, k0, k1, k2);
          break;
        default:                                           // OPFX_OP_DIV (the codes with long functions are not in this loop)
          OP_CASE();
#pragma unroll
          for (int r = 0; r < NR; ++r) v[r] = op_value<false>(OPFX_OP_DIV, rv[u][r], 0.0, k0, k1, k2);
      }
#undef OP_CASE
      if (FULL) {
        // (an op this instance's data source does not run stores nothing)
#pragma unroll
        for (int r = 0; r < NR; ++r)
          if (lane < ((mode[r] < 0 || ((q.ch[u][0] >> (8 + mode[r])) & 1)) ? n : 0)) rows[(size_t)r * row_doubles + q.ch[u][2] + lane] = v[r];
      } else if (lane < n) {
#pragma unroll
        for (int r = 0; r < NR; ++r) rows[(size_t)r * row_doubles + q.ch[u][2] + lane] = v[r];
      }
    }
  };
  OpRegs pre;
  if (R.n_stages > 0) op_load(R.st_ptr[0] + wib, R.st_ptr[1], pre);
  for (int sgi = 0; sgi < R.n_stages; ++sgi) {
    const int s0 = R.st_ptr[2 * sgi], sl = R.st_ptr[2 * sgi + 1], s1 = R.st_ptr[2 * sgi + 2];
    RSTAMPD(8);
    // (a stage whose ops depend on earlier ones chunk by chunk only — chunk i on chunk i, which the same wavefront ran,
    //  the stage lists being padded to whole rounds of the wavefronts — needs no barrier: the LDS keeps a wavefront's order)
    if (NR > 1 && ((R.st_barrier >> sgi) & 1)) lds_barrier(); else wave_fence();
    RSTAMPD(9);
    OpRegs cur = pre;
    if (sgi + 1 < R.n_stages) op_load(R.st_ptr[2 * sgi + 2] + wib, R.st_ptr[2 * sgi + 3], pre);
    RSTAMPD(10);
    op_apply(cur);
    RSTAMPD(11);
    for (int c0 = s0 + wib + NR * U; c0 < sl; c0 += NR * U) {
      op_load(c0, sl, cur);
      op_apply(cur);
    }
    for (int c = sl + wib; c < s1; c += NR) {
      const i32x8 ch = ld_chunk(R.och, c);
      const int n = ch[1], code = ch[0] & 0xFF, lc = lane < n ? lane : n - 1;
      const double k0 = ld_at(R.consts, (unsigned)(ch[4] + lc)), k1 = ld_at(R.consts, (unsigned)(ch[5] + lc)),
                   k2 = ld_at(R.consts, (unsigned)(ch[6] + lc));
      for (int r = 0; r < NR; ++r) {
        const long long b = b0 + r;
        if (b >= B) break;
        int md = -1;
#pragma unroll
        for (int q = 0; q < NR; ++q) if (q == r) md = mode[q];
        if (md >= 0 && !((ch[0] >> (8 + md)) & 1)) continue;
        const double rv = (ch[0] & OCH_READS_ROW) ? rows[(size_t)r * row_doubles + ch[3] + lc] : 0.0;
        double dr = 0.0;
        if (code == OPFX_OP_UNIFORM)
          dr = io.uniform ? ld_at(io.uniform + b * R.n_uniform, (unsigned)(ch[3] + lc)) : draw_uniform(io.rng_seed, b, (unsigned)(ch[3] + lc));
        if (code == OPFX_OP_NORMAL)
          dr = io.normal ? ld_at(io.normal + b * R.n_normal, (unsigned)(ch[3] + lc)) : draw_normal(io.rng_seed, b, (unsigned)(ch[3] + lc));
        const double v = op_value<true>(code, rv, dr, k0, k1, k2);
        if (lane < n) rows[(size_t)r * row_doubles + ch[2] + lane] = v;
      }
    }
  }
  // the first observation descriptors are on their way while the last stage finishes
  int ow[UC];
  const bool with_obs = io.obs && Ep;
  if (with_obs) {
#pragma unroll
    for (int u = 0; u < UC; ++u) { const int e = lane + 64 * u; ow[u] = ld_at(R.oe_src, (unsigned)(e < R.n_oel ? e : R.n_oel - 1)); }
  }
  if (NR > 1) lds_barrier(); else wave_fence();
  RSTAMP(3);
  if (!mine) return;
  if (with_obs) {
    // reset without power flow: initial action as ABSOLUTE set-points (opf_env.py:207), then the
    // table part of the observation (:218); result entries are NaN
    const DevEnv& E = *Ep;
    for (int k = lane; k < E.na; k += 64) {
      // (descriptors and the action first, unconditionally, then the arithmetic: one memory round trip)
      const unsigned ku = (unsigned)k;
      const int slot = ld_at(E.act_slot, ku), ls = ld_at(E.act_lo_slot, ku), hs = ld_at(E.act_hi_slot, ku);
      const double loc = ld_at(E.act_lo_const, ku), hic = ld_at(E.act_hi_const, ku), scal = ld_at(E.act_scaling, ku);
      const int kind = ld_at(E.act_kind, ku);
      const bool clampa = (E.clamp_enabled & 2) != 0;
      const int ch = clampa ? ld_at(E.clamp_hi_slot, ku) : -2, cl = clampa ? ld_at(E.clamp_lo_slot, ku) : -2;
      const double chc = clampa ? ld_at(E.clamp_hi_const, ku) : 0.0, clc = clampa ? ld_at(E.clamp_lo_const, ku) : 0.0;
      double a = io.action ? ld_at(io.action + bw * E.na, ku) : 0.0;
      double xv = row[slot];
      if (io.action) {
        a = (a != a) ? a : fmin(fmax(a, 0.0), 1.0);                                  // :429
        const double lo = ls >= 0 ? row[ls] : loc;
        const double hi = hs >= 0 ? row[hs] : hic;
        double spt = a * (hi - lo) + lo;                                              // :461
        if (clampa) {                                                                 // :464-470 (autoscale off)
          if (ch > -2) { const double m = ch >= 0 ? row[ch] : chc; if (spt > m) spt = m; }
          if (cl > -2) { const double m = cl >= 0 ? row[cl] : clc; if (spt < m) spt = m; }
        }
        xv = spt / scal;                                                              // :472-474
        if (kind != OPFX_ACT_CONTINUOUS) { xv = rint(xv); if (kind == OPFX_ACT_BOOLEAN) xv = xv != 0.0 ? 1.0 : 0.0; }
      }
      sp[k] = xv;
    }
    // (limits are read before any set-point is written: all set-points first, then their slots)
    wave_fence();
    for (int k = lane; k < E.na; k += 64) row[ld_at(E.act_slot, (unsigned)k)] = sp[k];
    wave_fence();
    RSTAMP(4);
    double* const out = io.obs + bw * E.nobs;
    for (int e0 = lane; e0 < R.n_oel; e0 += 64 * UC) {
      int w[UC];
      double v[UC];
#pragma unroll
      for (int u = 0; u < UC; ++u) { const int e = e0 + 64 * u; w[u] = e0 == lane ? ow[u] : ld_at(R.oe_src, (unsigned)(e < R.n_oel ? e : R.n_oel - 1)); }
#pragma unroll
      for (int u = 0; u < UC; ++u) v[u] = w[u] < 0 ? NaN : row[w[u]];        // (set-points live behind the row: sp = row + nxe)
#pragma unroll
      for (int u = 0; u < UC; ++u) { const int e = e0 + 64 * u; if (e < R.n_oel) out[e] = v[u]; }
    }
  }
  RSTAMP(5);
  row_copy_out<UC>(xr, row, R.nx, lane);
  wave_fence();
#ifdef OPFX_ENABLE_STAMPS
  __builtin_amdgcn_s_waitcnt(0);
#endif
  RSTAMP(6);
}

// NR: wavefronts per workgroup = rows per workgroup (as many as the LDS takes next to three more workgroups)
// The reset programme's descriptor (about 70 dwords) is passed BY POINTER and read where it is needed, through the scalar
// cache: by value all of it was loaded at kernel entry and parked in VGPR lanes (132 v_writelane in the prologue, a
// v_readlane at every use — a tenth of the kernel's vector instructions, round 3).
template <int NR, bool FULL>
__global__ __launch_bounds__(64 * NR, NR == 1 ? 2 : 4) void k_reset(const DevReset* __restrict__ Rp, const DevEnv* __restrict__ Ep, ResetIO io, long long B, int row_doubles) {
  const DevReset& R = *Rp;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x & 63;
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);          // (wave-uniform: scalar addressing)
  // (the four wavefronts that share a SIMD — of four workgroups — rotate through the user priority levels, one step per
  //  set of rows: the arbiter's oldest-first rule otherwise lets the workgroups finish in the order of their age and the
  //  launch end with the youngest; 16 384 rows 57.9 -> 54.4 us, 8 192: 33.3 -> 32.9; see k_step)
  int turn = (int)(__builtin_amdgcn_s_getreg(GETREG_HW_ID) & 3u);     // HW_ID: the wave slot on its SIMD
  for (long long b0 = (long long)blockIdx.x * NR; b0 < B; b0 += (long long)gridDim.x * NR) {
    switch (turn++ & 3) { case 0: __builtin_amdgcn_s_setprio(0); break; case 1: __builtin_amdgcn_s_setprio(1); break;
                          case 2: __builtin_amdgcn_s_setprio(2); break; default: __builtin_amdgcn_s_setprio(3); }
    reset_rows<NR, NR == 2 ? 4 : (NR == 4 && FULL ? 2 : 3), (NR == 1 || FULL) ? 6 : 9, FULL>(R, Ep, io, b0, B, lane, wib, smem, row_doubles);
    if (NR > 1) lds_barrier();            // (the next rows are written by all wavefronts)
  }
}

// cost of one cost row (objective.py:34-77) given its active/reactive power; coefficients are
// constants or sampled prices living in the instance's table row `xc`
__device__ __forceinline__ double cost_row(const DevEnv& E, const double* xc, int meta, int cbase, double pw_, double qv_) {
  if (!(meta & 16)) {
    double cf[6];
    int xsl[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) { xsl[q] = as_global(E.coef_xslot)[cbase + q]; cf[q] = as_global(E.cost_coef)[cbase + q]; }
#pragma unroll
    for (int q = 0; q < 6; ++q) if (xsl[q] >= 0) cf[q] = xc[xsl[q]];
    double pc = cf[0]; pc += cf[1] * pw_; pc += cf[2] * (pw_ * pw_);          // :38-40
    double qc = cf[3]; qc += cf[4] * qv_; qc += cf[5] * (qv_ * qv_);          // :41-43
    return pc + qc;
  }
  const double pwr = (meta & 32) ? qv_ : pw_;
  const double s = sgn(pwr), pa = fabs(pwr);
  double cst = 0.0;
  for (int sg = 0; sg < E.nseg; ++sg) {                                     // :60-75
    const int ci = cbase + sg * 3;
    const double lo = as_global(E.cost_coef)[ci], hi = as_global(E.cost_coef)[ci + 1];
    const int xsl = as_global(E.coef_xslot)[ci + 2];
    double price = as_global(E.cost_coef)[ci + 2];
    if (xsl >= 0) price = xc[xsl];
    const double la = fabs(lo), ha = fabs(hi);
    const double inside = fmin(la, ha);
    const bool same = (s == sgn(lo + hi));
    const bool in_f = (pa > inside) && same;
    const bool out_f = pa > fmax(la, ha);
    if (out_f) cst += s * (hi - lo) * price;
    if (in_f && !out_f) cst += s * (pa - inside) * price;
  }
  return cst;
}

// One launch = one env.step() for B instances.  Per instance: (1) the table row is staged in
// LDS with coalesced loads (all HBM traffic of the prologue is in flight at once; every later
// gather hits LDS), actions -> set-points, bus injections, table observations and the cost
// rows that do not depend on the solve; (2) Newton; (3) results, constraints, remaining
// costs, reward, result observations.  Descriptor loads are batched (fixed unroll, clamped
// indices) so that each phase pays one L2 round trip, not one per 64 items.
// MINW: wavefronts per SIMD the kernel is compiled for (launch bounds: 2 -> up to 256 VGPRs, 3 -> 168).  Three is for ONE case,
// measured in round 5: a grid whose instance takes just under a third of the LDS (the 306-bus grid on a plan with shared slots)
// runs 13.7 % faster as three teams of FOUR per CU — twelve wavefronts, 168 VGPRs and 216 B of scratch per lane — than as two
// teams of four at 216 VGPRs without scratch, and 10 % faster than as three teams of two (profiles/r05_ab_three_teams.txt).
template <int V2, int NW, bool DC = false, bool MEM = false, bool CHORD = false, int SPEC_ = 0, int MINW = OPFX_MIN_WAVES_PER_SIMD>
__global__ __launch_bounds__(WAVE * NW, MINW) void k_step(const DevPlan P, const DevEnv* __restrict__ Ep, StepIO io, Opts o,
                                                  long long B) {
  // The environment descriptor (about 50 pointers) stays in memory and is read where it is
  // needed: held in SGPRs it would be spilled to VGPR lanes across the whole Newton loop.
  const DevEnv& E = *Ep;
  constexpr int SPEC = V2 ? (SPEC_ | OPFX_FORCE_SPEC) : 0;      // (the first-generation kernel is not specialised)
  constexpr bool NOPV = (SPEC & SPEC_NO_PV) != 0, NOMOD = (SPEC & SPEC_NO_MOD) != 0;
  // (the polar shadow, see Polar: single-wave kernels at two wavefronts per SIMD without modifiers; the launch keeps grids of
  //  more than 64 POLAR_R buses off these instantiations, do_step)
  constexpr bool POLAR = V2 != 0 && NW == 1 && !MEM && !(DC && !OPFX_POLAR_DC) && !CHORD && MINW == 2 && NOMOD;
  constexpr bool PQREG = OPFX_PQREG && POLAR && NOPV && !(DC && !OPFX_PQREG_DC);      // scheduled P / Q in registers (see Polar)
  constexpr bool TPQ = OPFX_PQREG && OPFX_TPQ && V2 != 0 && NW > 1 && NOPV && !MEM;      // ... of a team's threads (newton2_coop)
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int NT = WAVE * NW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const Lds L = carve<V2, MEM>(P, E.na, E.nblk_d, smem, env_nacc(E.nc), E.max_mod);
  const int nb = P.nb;
  const double NaN = __builtin_nan("");
  double* const xs = L.rhs;                  // staged table row: [rhs | LU blocks] are free outside the solve
  if (V2) {
    for (int i = tid; i < nb; i += NT) L.dg[i] = (unsigned short)P.diag_blk[i];
    for (int i = tid; i < P.tail_n; i += NT) L.tl[i] = P.tail_ids[i];
    blk_sync<NW>();
  }
  OPFX_STAMP_INIT();
  // With many instances per workgroup (opfx_step decides, use_queue: twelve or more per single-wave workgroup, eight or more
  // per wave team) the instances beyond a workgroup's first come
  // from a QUEUE (an atomic counter in the context): instances differ in their Newton iteration count, and with a fixed
  // share per workgroup the launch ends with the unluckiest one (65 536 instances of config 4: 34.5 mean instance times
  // on the slowest of 2 048 wavefronts against a mean load of 32.0; a queue brings that to 32.6).  The next index is
  // fetched after the solve, so that the round trip hides behind the instance's epilogue — not earlier: a claim made an
  // instance ahead is a static assignment again (measured at four instances per wavefront: +12 %).  opfx_step zeroes the
  // counter — the environment's own — in front of such a launch.  With few
  // instances per workgroup a queue cannot help (four jobs per worker: the greedy makespan equals the static one) and
  // the shares stay fixed.
#ifdef OPFX_ENABLE_STAMPS
  const double t_start__ = (double)wall_clock64();
#endif
  int turn = NW == 1 ? (int)(__builtin_amdgcn_s_getreg(GETREG_HW_ID) & 1u) : 0;     // (HW_ID: the wave slot on its SIMD)
  if (NW > 1) {       // the two teams of a CU in turns as well (the slot parity of wavefront 0 stands for its team):
                      // -3.0 % on config 3 with fixed shares, nothing on top of the work queue
    if (tid == 0) L.acc[0] = (double)(__builtin_amdgcn_s_getreg(GETREG_HW_ID) & 1u);
    __syncthreads();
    turn = (int)L.acc[0];
    __syncthreads();
  }
  for (long long b = blockIdx.x; b < B;) {
    if (NW > 1) { if (turn & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); ++turn; }
    if (NW == 1) {
      // The SIMD's arbiter serves the OLDER of two ready wavefronts first: of the two single-wave instances that share a
      // SIMD one ran 6 % ahead of the other through the whole launch (per-workgroup busy times, scripts/probe_finish_times.py)
      // and the launch ended with the slower ones.  The two take the higher user priority in turns, instance by instance.
      if (turn & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
      ++turn;
    }
    double* xr = uniform_ptr(io.x + b * E.nx);           // (opaque row pointer: see uniform_ptr)
    const bool apply = io.mode != 1 && io.mode != 3;
    OPFX_STAMP(15);
    // ---- stage the row ------------------------------------------------------------
    OPFX_REP(0) {
      // (only the columns a descriptor of this environment names — the caller lays the columns it merely KEEPS per
      //  instance, e.g. intermediates of the reset programme, behind them: VERDICT r05 #5, HBM reads of config 2 -24 %)
      const int nx = E.nx_hot;
      int q = tid;
      for (; q + 7 * NT < nx; q += 8 * NT) {
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = ld_at(xr, (unsigned)(q + u * NT));
#pragma unroll
        for (int u = 0; u < 8; ++u) xs[q + u * NT] = t[u];
      }
      // (the rest in batches of clamped loads as well — eight per lane for a single wavefront, two for a team —: one load per
      //  pass of this loop is one memory round trip per pass; with the row cut at nx_hot the rest grew from 58 to 302 columns
      //  on config 2 and five serial round trips cost 0.8 % of the step)
      constexpr int TB = NW == 1 ? 8 : 2;
      for (; q < nx; q += TB * NT) {
        double t[TB];
#pragma unroll
        for (int u = 0; u < TB; ++u) t[u] = ld_at(xr, (unsigned)min(q + u * NT, nx - 1));
#pragma unroll
        for (int u = 0; u < TB; ++u) if (q + u * NT < nx) xs[q + u * NT] = t[u];
      }
    }
    blk_sync<NW>();       // staged row visible (one wave: compiler fence; team: all waves staged)
    OPFX_STAMP(16);
    // ---- apply actions (opf_env.py:421-491) -----------------------------------
    double corr = 0.0;
    OPFX_REP(1) if (wave == 0) {
      // (no action row in modes 1/3: any readable row keeps the loads unconditional)
      const double* act_row = uniform_ptr(apply ? io.action + b * E.na : xr);
      // reset applies its initial action as ABSOLUTE set-points (opf_env.py:207 passes no step size),
      // and clamps only without autoscaling (:464)
      const bool as_reset = io.mode == 2 || io.mode == 4 || io.mode == 5;
      const double diff_step = as_reset ? 0.0 : E.diff_step;
      const bool clamp = as_reset ? (E.clamp_enabled & 2) != 0 : (E.clamp_enabled & 1) != 0;
      for (int k0 = 0; k0 < E.na; k0 += 2 * WAVE) {
        int slot[2], los[2], his[2], cls_[2], chs[2], kind[2];
        double av[2], sc[2], loc[2], hic[2], clc[2], chc[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int k = k0 + u * WAVE + lane, kk = k < E.na ? k : E.na - 1;
          slot[u] = as_global(E.act_slot)[kk]; los[u] = as_global(E.act_lo_slot)[kk]; his[u] = as_global(E.act_hi_slot)[kk];
          sc[u] = as_global(E.act_scaling)[kk]; loc[u] = as_global(E.act_lo_const)[kk]; hic[u] = as_global(E.act_hi_const)[kk];
          av[u] = ld_at(act_row, (unsigned)kk);
          kind[u] = as_global(E.act_kind)[kk];
          cls_[u] = as_global(E.clamp_lo_slot)[kk]; chs[u] = as_global(E.clamp_hi_slot)[kk];
          clc[u] = as_global(E.clamp_lo_const)[kk]; chc[u] = as_global(E.clamp_hi_const)[kk];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int k = k0 + u * WAVE + lane;
          if (k >= E.na) continue;
          double xv = xs[slot[u]];
          if (apply) {
            double a = av[u];
            a = (a != a) ? a : fmin(fmax(a, 0.0), 1.0);                         // :429 (a NaN action stays NaN -> failed row)
            const double lo = los[u] >= 0 ? xs[los[u]] : loc[u];
            const double hi = his[u] >= 0 ? xs[his[u]] : hic[u];
            const double delta = hi - lo;
            double spt;
            if (diff_step != 0.0) spt = (a * 2.0 - 1.0) * diff_step * delta + xv * sc[u];       // :453-458
            else spt = a * delta + lo;                                                   // :461
            if (clamp) {                                                                 // :464-470
              if (chs[u] > -2) { const double m = chs[u] >= 0 ? xs[chs[u]] : chc[u]; if (spt > m) spt = m; }
              if (cls_[u] > -2) { const double m = cls_[u] >= 0 ? xs[cls_[u]] : clc[u]; if (spt < m) spt = m; }
            }
            xv = spt / sc[u];                                                            // :472-474
            if (kind[u] != OPFX_ACT_CONTINUOUS) {                                        // :476-481 (np.round: half to even)
              xv = rint(xv);
              if (kind[u] == OPFX_ACT_BOOLEAN) xv = xv != 0.0 ? 1.0 : 0.0;
            }
#ifndef OPFX_PROBE_NO_SETPOINT_WRITEBACK        // (probe build: what the scattered 8-byte set-point stores cost in HBM writes, EXPERIMENTS #33)
            st_at(xr, (unsigned)slot[u], xv);                                            // :483
#endif
            const double cur = (xv * sc[u] - lo) / delta;                                // :586
            corr += (delta != 0.0) ? fabs(cur - a) : 0.0;                                // D11 guard
          }
          L.sp[k] = xv;
        }
      }
      corr = E.na > 0 ? wave_sum_dpp(corr) / E.na : 0.0;                                   // :488-489
    }
    blk_sync<NW>();       // NW > 1: row staged by all waves, set-points by wave 0
    OPFX_STAMP(17);
    // (limits are read before any set-point of this step is written back: xs keeps the
    //  pre-step values, exactly as the reference reads min/max columns that actions never touch)
    // ---- table observations: do not depend on the solve ----------------------------------------
    // (the copy phases of a team's prologue and epilogue — table observations, injections, results, result observations — are
    //  shared by its wavefronts: segment by segment, or thread by thread; with one wavefront this is the code it always was)
    OPFX_REP(2) if (io.obs) for (int sg = wave; sg < E.n_oseg; sg += NW) {
      const int kind = as_global(E.oseg_kind)[sg], src = as_global(E.oseg_src)[sg], dst = as_global(E.oseg_dst)[sg], n = as_global(E.oseg_n)[sg];
      // (stores as wave-uniform base + 32-bit lane offset: no 64-bit per-lane address, which the compiler would compute
      //  once per kernel — pointer + lane * 8 — and keep in two VGPRs through every Newton loop)
      double* const ob = uniform_ptr(io.obs + b * E.nobs + dst);
      if (kind == 1) {
        if (io.mode == 2 || io.mode == 3) for (int j = lane; j < n; j += WAVE) st_at(ob, (unsigned)j, NaN);
        continue;
      }
      const double* from = (kind == 0 ? xs : L.sp) + src;
      for (int j = lane; j < n; j += WAVE) st_at(ob, (unsigned)j, from[j]);
    }
    if (io.mode == 2 || io.mode == 3) {
      // reset without power flow (opf_env.py:207,218): set-points applied, table observation only
      if (tid == 0 && io.mean_correction) io.mean_correction[b] = corr;
      b = next_instance<NW>(L.acc + env_nacc(E.nc) - 2, b, io.queued != 0, io.queued && tid == 0 ? __hip_atomic_fetch_add(P.queue, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0);
      continue;
    }
    OPFX_STAMP(18);
    Polar pol;                               // polar shadow / scheduled P, Q of the lane's buses (POLAR, PQREG kernels)
    double csum = 0.0;                       // this lane's share of the cost rows
    {
      // ---- bus injections (makeSbus): flat list, LDS accumulate ----------------------------------
      // accumulated in the voltage arrays (free until init_voltage), then written to this
      // workgroup's P/Q row
      double* const pacc = V2 ? L.vr : L.psp;
      double* const qacc = V2 ? L.vi : L.qsp;
      OPFX_REP(3) {
      for (int i = tid; i < nb; i += NT) { pacc[i] = 0.0; qacc[i] = 0.0; L.bt[i] = BT_PQ; }
      blk_sync<NW>();
      // (the entries of the next batch are requested before this batch is added up: one memory round trip for the list
      //  instead of one per batch of 256 entries)
      const int n_inj = E.n_inj;
      const uint4* const inj_pk = E.inj_pk;
      struct InjBatch { uint4 d[4]; };
      auto ld_inj = [&](int e0) {
        InjBatch r;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int e = e0 + u * NT + tid;
          typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
          const u32x4 w = ld_at(reinterpret_cast<const u32x4*>(inj_pk), (unsigned)(e < n_inj ? e : n_inj - 1));
          r.d[u] = make_uint4(w.x, w.y, w.z, w.w);
        }
        return r;
      };
      InjBatch nxt_inj{};
      if (n_inj > 0) nxt_inj = ld_inj(0);
      for (int e0 = 0; e0 < n_inj; e0 += 4 * NT) {
        const InjBatch cur_inj = nxt_inj;
        nxt_inj = ld_inj(e0 + 4 * NT);
        const uint4 (&d)[4] = cur_inj.d;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int e = e0 + u * NT + tid;
          if (e >= n_inj) continue;
          const int bq = d[u].x;
          const double v = u2d(d[u].z, d[u].w) * src_val(xs, L.sp, (int)d[u].y);
          lds_add(((bq >> 16) ? qacc : pacc) + (bq & 0xFFFF), v);
        }
      }
      if (PQREG) {
        blk_sync<NW>();
        // (no PV bus, no modifier: the sums never change after this point and stay with the lane that owns the bus)
#pragma unroll
        for (int r = 0; r < POLAR_R; ++r) { const int i = min(lane + WAVE * r, nb - 1); pol.p[r] = pacc[i]; pol.q[r] = qacc[i]; }
      } else if (TPQ) {
        blk_sync<NW>();
#pragma unroll
        for (int k = 0; k < TEAM_PQ_R; ++k) { const int i = min(tid + NT * k, nb - 1); pol.p[k] = pacc[i]; pol.q[k] = qacc[i]; }
        blk_sync<NW>();                      // (the sums are out of the voltage arrays before init_voltage writes them)
      } else if (V2) {
        blk_sync<NW>();
        // (scalar base + 32-bit lane offset: the per-workgroup rows never turn into 64-bit per-lane addresses that the
        //  compiler computes at kernel entry and keeps — or spills — through every Newton loop)
        for (int i = tid; i < nb; i += NT) { const unsigned io_ = opaque((unsigned)i); st_at(L.psp, io_, pacc[i]); st_at(L.qsp, io_, qacc[i]); }
        if (NW > 1) blk_sync<NW>();          // (the sums are out of the voltage arrays before wavefront 0 starts them: init_voltage)
      }
      }
    }
    if (wave == 0) {
      OPFX_STAMP(19);
      // ---- cost rows whose power is a table value / set-point (objective.py:34-54) --------------
      // (Every phase of this prologue starts with descriptor loads and so with a memory round trip of its own,
      // 17 k of the 135 k cycles of a 144-bus step in the cycle stamps.  Issuing each phase's first batch of loads
      // one phase ahead shortened the prologue by 12 % and the kernel by nothing, 0.3027 vs 0.3003 ms: the second
      // wavefront of the SIMD already fills those waits; what bounds the kernel is the instructions it issues.)
      OPFX_REP(4) for (int r0 = 0; r0 < E.ncost_pre; r0 += 2 * WAVE) {
        int meta[2], ps[2], qs[2], cb[2];
        double scl[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int r = r0 + u * WAVE + lane, rr = r < E.ncost_pre ? r : E.ncost_pre - 1;
          meta[u] = as_global(E.cost_meta)[rr]; ps[u] = as_global(E.cost_psrc)[rr]; qs[u] = as_global(E.cost_qsrc)[rr]; cb[u] = as_global(E.cost_cbase)[rr];
          scl[u] = as_global(E.cost_scale)[rr];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int r = r0 + u * WAVE + lane;
          if (r >= E.ncost_pre) continue;
          csum += cost_row(E, xs, meta[u], cb[u], src_val(xs, L.sp, ps[u]) * scl[u], src_val(xs, L.sp, qs[u]) * scl[u]);
        }
      }
    }
    // (the pre-solve rows are summed over the wavefront HERE: a wave-uniform double lives in two SGPRs through the solve,
    //  a per-lane partial sum in two VGPRs — and the wave-team kernels sit at the top of the register file)
    const double cost_pre = wave_sum_dpp(csum);
    csum = 0.0;
    OPFX_STAMP(0);
    // ---- base case + N-1 contingencies (security_constrained.py:37-68) --------
    double objective = 0.0, viol_acc = 0.0, pen_acc = 0.0;   // lane g < nc holds group g
    int valid_acc = 1;
    bool conv0 = false;
    int iters0 = 0, iters_all = 0;
    double nrm0 = 0.0;
    double min_piv = V2 ? 1.0 : __builtin_nan("");
    int min_piv_bus = -1;
    const int base_out = (!NOMOD && io.outage) ? io.outage[b] : -1;
    // modifiers of this instance: [env modifiers (taps, switches) | outage | contingency]
    int n_mod_base = 0;
    int n_rem_base = 0, isl_br_base = -1;       // removed branches so far / one of them that islands
    if (V2 && !NOMOD) for (int m = 0; m < E.n_bmod; ++m) {
      // stamps of this branch for the instance's state (tap position, switch / in_service flag)
      const int br = as_global(E.bmod_branch)[m];
      const int st = (int)rint(src_val(xs, L.sp, as_global(E.bmod_src)[m])) - as_global(E.bmod_lo)[m];
      const int row = as_global(E.bmod_ptr)[m] + min(max(st, 0), as_global(E.bmod_n)[m] - 1);
      if (br < 0) {
        // a bus shunt in steps (bus -1 - br): the row holds (…, dG, dB), the DIFFERENCE to the compiled shunt of the bus
        double dsh = 0.0;
        if (lane >= 6 && lane < 8) dsh = as_global(E.bmod_y)[row * 8 + lane];
        if (!__any(dsh != 0.0)) continue;
        if (wave == 0) mod_set_shunt(P, L, lane, n_mod_base, -1 - br, dsh);
        ++n_mod_base;
        continue;
      }
      double y = 0.0, dy = 0.0;
      if (lane < 8) { y = as_global(E.bmod_y)[row * 8 + lane]; dy = y - P.br_y[br * 8 + lane]; }
      if (br == base_out || !__any(dy != 0.0)) continue;      // outaged anyway / state = compiled state
      const bool removed = !__any(y != 0.0);
      const bool uncoupled = !__any(lane >= 2 && lane < 6 && y != 0.0);     // out of service or open-ended
      if (wave == 0) mod_set(P, L, lane, n_mod_base, br, dy, removed, 0, uncoupled);
      ++n_mod_base;
      if (uncoupled) { ++n_rem_base; if (P.br_island[br]) isl_br_base = br; }
    }
    if (V2 && base_out >= 0) { if (wave == 0) mod_set(P, L, lane, n_mod_base, base_out, 0.0, true, 0); ++n_mod_base; }
    if (base_out >= 0) { ++n_rem_base; if (P.br_island[base_out]) isl_br_base = base_out; }
    // (a reset runs the base case only: the reference's reset calls run_power_flow, opf_env.py:209-216;
    // the N-1 loop belongs to calculate_violations, security_constrained.py:37)
    const int n_cont_run = (NOMOD || io.mode == 2 || io.mode == 4) ? 0 : E.n_cont;
    for (int c = 0; c <= n_cont_run; ++c) {
      const int out_br = c == 0 ? base_out : as_global(E.cont_branch)[c - 1];
      if (c > 0 && out_br == base_out) continue;            // already out of service (:46-48)
      int n_mod = n_mod_base;
      if (V2 && c > 0) { if (wave == 0) mod_set(P, L, lane, n_mod, out_br, 0.0, true, n_mod_base); ++n_mod; }
      const int n_rem = n_rem_base + (c > 0 ? 1 : 0);
      const int isl_br = (c > 0 && P.br_island[out_br]) ? out_br : isl_br_base;
      // one branch out: its cut-off set is precomputed; several: connectivity is labelled per instance
      const bool multi = V2 != 0 && n_rem >= 2;
      const int isl = multi ? 1 : island_state(V2 != 0, n_rem, isl_br >= 0);
      // (the start voltages — and the base-case voltages a contingency starts from — are written by the whole team: with 250
      //  contingencies per step this runs 251 times per instance; the rare rest keeps to wavefront 0, between two barriers)
      OPFX_REP(5) init_voltage<V2, SPEC>(P, L, tid, E.qg_min, E.qg_max, o.enforce_q_lims == 1, NT);
      const bool rare_start = multi || isl == 1 || (!NOMOD && E.vset_src != nullptr);
      if (NW > 1 && rare_start) blk_sync<NW>();
      if (wave == 0) {
        if (POLAR) {                       // the polar shadow starts where init_voltage starts V (plan: vr0 + j vi0 = vm_set e^{j va_set})
#pragma unroll
          for (int r = 0; r < POLAR_R; ++r) {
            const unsigned i = (unsigned)min(lane + WAVE * r, nb - 1);
            pol.th[r] = ld_at(P.va_set, i); pol.vm[r] = ld_at(P.vm_set, i);
          }
        }
        if (multi) mark_islands_multi(P, L, lane, n_mod, E.qg_min, E.qg_max);
        else if (isl == 1) mark_island(P, L, lane, isl_br, E.qg_min, E.qg_max);
        if (!NOMOD && E.vset_src) for (int i = lane; i < nb; i += WAVE) {
          // per-instance |V| set-point of a REF / PV bus (a sampled ext_grid.vm_pu)
          const int src = as_global(E.vset_src)[i];
          if (src == NOSRC) continue;
          const double f = src_val_g(xr, L.sp, src) / P.vm_set[i];
          L.vr[i] *= f; L.vi[i] *= f;
          if (!V2) L.vm[i] *= f;
        }
      }
      if (NW > 1 && rare_start) blk_sync<NW>();
      if (c > 0 && o.contingency_start == 0) {
        // contingency cases start from the base-case solution (the reference restarts
        // pandapower from scratch for each one; the converged result is the same;
        // opfx_solve_opts::contingency_start = 1 does exactly what the reference does)
        // A PV bus keeps the MAGNITUDE init_voltage gave it (its set-point; the Newton loop never moves |V| of a PV row) and takes
        // only the angle: in the base case the bus may have run into a reactive limit and floated away from its set-point —
        // started from that |V| the contingency would hold the wrong voltage, possibly meet the other limit, and end at
        // another solution than pandapower's, which starts every power flow with all generators regulating (found by the fuzzer
        // on grids whose generators have narrow ranges, round 6).
        const double* wv = P.warm + (size_t)blockIdx.x * 2 * nb;
        for (int i = tid; i < nb; i += NT) {
          const unsigned io_ = opaque((unsigned)i);
          double wr = ld_at(wv, io_), wi = ld_at(wv, (unsigned)nb + io_);
          if (!NOPV && V2 && L.bt[i] == BT_PV) {
            const double f = sqrt((L.vr[i] * L.vr[i] + L.vi[i] * L.vi[i]) / (wr * wr + wi * wi));
            wr *= f; wi *= f;
          }
          L.vr[i] = wr; L.vi[i] = wi;
        }
      }
      // A contingency that starts from a DC power flow of its own (init = DC, contingency_start = 1: what the reference does,
      // security_constrained.py:53) — B' is the SAME matrix for every instance and every step, and the outage of branch
      // (f, t) is a rank-1 change of it: with w = B'^-1 (e_f - e_t), computed once per grid and contingency on the host,
      // Sherman-Morrison gives theta_c = theta_0 + w (Pfinj + b (theta_0f - theta_0t)) / (1 - b (w_f - w_t)) from the BASE
      // case's DC angles theta_0 (kept in the workgroup's scratch row by the base case's DC pass): one pass over the buses
      // instead of a factorisation and two substitutions through the block-LU schedule per contingency.  Only on the compiled
      // topology (no modifier in the base case, nothing islanded); otherwise the solve runs its own DC pass as before.
      bool dc_rank1 = false;
      if (DC && !MEM && V2 && !NOMOD && c > 0 && o.init == OPFX_INIT_DC && o.contingency_start == 1 && P.theta0 != nullptr
          && E.cont_dc_w != nullptr && n_mod_base == 0 && isl == 0 && conv0) {
        const double* const rec = E.cont_dc_k + 4 * (size_t)(c - 1);
        const double inv_d = rec[2];
        if (inv_d != 0.0) {
          dc_rank1 = true;
          const double* const th0 = P.theta0 + (size_t)blockIdx.x * 2 * nb;
          const int f = P.br_f[out_br], t = P.br_t[out_br];
          const double thf = L.bt[f] == BT_REF ? P.va_set[f] : ld_at(th0, (unsigned)f);
          const double tht = L.bt[t] == BT_REF ? P.va_set[t] : ld_at(th0, (unsigned)t);
          const double alpha = (rec[1] + rec[0] * (thf - tht)) * inv_d;
          const double* const w = E.cont_dc_w + (size_t)(c - 1) * nb;
          for (int i = tid; i < nb; i += NT) {
            if (L.bt[i] == BT_REF) continue;
            const unsigned io_ = opaque((unsigned)i);
            const double dth = ld_at(th0, io_) + alpha * ld_at(w, io_) - P.va_set[i];     // (turn the start voltage, as the DC pass's phase D)
            double sn, cs;
            sincos(dth, &sn, &cs);
            const double vr = L.vr[i], vi = L.vi[i];
            L.vr[i] = vr * cs - vi * sn;
            L.vi[i] = vr * sn + vi * cs;
          }
        }
      }
      blk_sync<NW>();
      int iters; double nrm;
      OPFX_STAMP_RESET();
      const bool conv = solve_instance<V2, NW, DC, MEM, CHORD, SPEC, POLAR>(P, L, o, lane, out_br, n_mod, E.qg_min, E.qg_max, &iters, &nrm, &min_piv, &min_piv_bus, isl, &pol, (c > 0 && o.contingency_start == 0) || dc_rank1);
      iters_all += iters;
      blk_sync<NW>();
      OPFX_STAMP(5);
      if (c == 0) {
        conv0 = conv; iters0 = iters; nrm0 = nrm;
        if (!conv) break;
        // (the DC angles the base case's DC pass stored are read by other lanes below: a team has met at a workgroup barrier since,
        //  a single wavefront waits for its own stores.  Compile-time for the kernels without the DC start: the test alone — a
        //  kernel-argument load and a branch per instance — cost config 3 0.8 %, profiles/r06_ab_bisect_c3.txt)
        if (DC && NW == 1 && P.theta0) __builtin_amdgcn_s_waitcnt(0);
      // (the DC angles the base case's DC pass stored: read by other lanes below)
        if (!NOMOD && E.n_cont > 0 && o.contingency_start == 0) {      // (the row holds the base case's DC angles otherwise)
          double* wv = P.warm + (size_t)blockIdx.x * 2 * nb;
          for (int i = tid; i < nb; i += NT) { const unsigned io_ = opaque((unsigned)i); st_at(wv, io_, L.vr[i]); st_at(wv, (unsigned)nb + io_, L.vi[i]); }
          __builtin_amdgcn_s_waitcnt(0);      // written and read back by the same threads (the same bus -> thread map)
        }
      }
      if (!conv) {
        // failed contingency: all invalid, +not_converged_penalty (sign as in the reference, D6)
        valid_acc = 0;
        viol_acc += E.not_converged_penalty;
        pen_acc += E.not_converged_penalty;
        continue;
      }
      double* R = L.stage;
      // (the result bank is filled by the whole team; constraints, costs and outputs by wavefront 0)
      // (voltage angles: for the result bank, which is written from the base case only, or when an observation /
      //  constraint / objective term reads them — not for the 250 contingency cases of an N-1 step otherwise)
      OPFX_REP(6) compute_results<V2, SPEC, POLAR, TPQ, PQREG>(P, L, tid, out_br, n_mod, E.qg_min, E.qg_max, R, true, (c == 0 && io.results != nullptr) || E.need_angle, NT, &pol);
      blk_sync<NW>();
      // (derived rows and the constraint pass are shared by the whole team as well — with 250 contingencies per step they run
      //  251 times per instance and were 5 % of an N-1 step on wavefront 0 alone, the other three parked at the barrier)
      OPFX_REP(7) for (int k = tid; k < E.n_xres; k += NT) {          // derived rows: unit power echoes, apparent power
        const double sc = as_global(E.xres_scale)[k];
        const int kind = as_global(E.xres_kind)[k];
        double v;
        if (kind == OPFX_XRES_MAX3) {                      // (rows of this kind come after the rows they read)
          v = nan_max(nan_max(R[as_global(E.xres_p)[k]], R[as_global(E.xres_q)[k]]), R[as_global(E.xres_r)[k]]);
        } else if (kind == OPFX_XRES_AFFINE) {
          // one generator's share of the reactive power generated at its bus (pypower pfsoln), zero on a de-energised bus
          const int bus = as_global(E.xres_q)[k];
          v = (!NOMOD && bus >= 0 && L.bt[bus] == BT_DEAD) ? 0.0 : as_global(E.xres_off)[k] + sc * R[as_global(E.xres_p)[k]];
        } else {
          const double pv_ = src_val_g(xr, L.sp, as_global(E.xres_p)[k]) * sc;
          v = pv_;
          if (kind == OPFX_XRES_S) { const double qv = src_val_g(xr, L.sp, as_global(E.xres_q)[k]) * sc; v = sqrt(pv_ * pv_ + qv * qv); }
        }
        R[E.nres_base + k] = v;
      }
      for (int q = tid; q < 5 * E.nc; q += NT) L.acc[q] = 0.0;
      blk_sync<NW>();
      OPFX_STAMP(6);
      // ---- constraints (constraints.py:70-128): one pass over all bounded values; the rare
      // violating lanes accumulate per-constraint sum / worst case / count in LDS
      OPFX_REP(8) {
        // (the descriptors of the next batch are requested before this batch is worked: one memory round trip for the pass
        //  instead of one per batch)
        struct ConBatch { int2 cd[2]; double lo[2], hi[2]; };
        const int ncel = E.ncel;
        const int2* const con_pk = E.con_pk; const double* const con_min = E.con_min; const double* const con_max = E.con_max;
        auto ld_con = [&](int e0) {
          ConBatch c;
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int e = e0 + u * NT + tid;
            const unsigned ee = (unsigned)(e < ncel ? e : ncel - 1);
            const long long pk = ld_at(reinterpret_cast<const long long*>(con_pk), ee);
            c.cd[u] = make_int2((int)(unsigned)pk, (int)(pk >> 32)); c.lo[u] = ld_at(con_min, ee); c.hi[u] = ld_at(con_max, ee);
          }
          return c;
        };
        ConBatch nxt_con{};
        if (ncel > 0) nxt_con = ld_con(0);
        for (int e0 = 0; e0 < ncel; e0 += 2 * NT) {
          const ConBatch cb_ = nxt_con;
          nxt_con = ld_con(e0 + 2 * NT);
          const int2 (&cd)[2] = cb_.cd;
          const double (&lo)[2] = cb_.lo, (&hi)[2] = cb_.hi;
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int e = e0 + u * NT + tid;
            if (e >= ncel) continue;
            const double v = R[cd[u].x];
            double* a = L.acc + 5 * cd[u].y;
            if (v < lo[u]) {
              const double d = fabs(v - lo[u]);
              lds_add(a + 0, d); lds_add(a + 4, 1.0);
              __hip_atomic_fetch_max(reinterpret_cast<unsigned long long*>(a + 2), (unsigned long long)__double_as_longlong(d),
                                     __ATOMIC_RELAXED, NW == 1 ? __HIP_MEMORY_SCOPE_WAVEFRONT : __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (v > hi[u]) {
              const double d = fabs(v - hi[u]);
              lds_add(a + 1, d); lds_add(a + 4, 1.0);
              __hip_atomic_fetch_max(reinterpret_cast<unsigned long long*>(a + 3), (unsigned long long)__double_as_longlong(d),
                                     __ATOMIC_RELAXED, NW == 1 ? __HIP_MEMORY_SCOPE_WAVEFRONT : __HIP_MEMORY_SCOPE_WORKGROUP);
            }
          }
        }
      }
      // (the result bank is complete and stays as it is until the next solve: written out by the whole team)
      if (c == 0) OPFX_REP(10) if (io.results) { double* const rb = uniform_ptr(io.results + b * E.nres); for (int q = tid; q < E.nres; q += NT) st_at(rb, (unsigned)q, R[q]); }
      // result observations reflect the LAST solved case (defect D7 of the reference)
      OPFX_REP(11) if (io.obs && E.n_oseg_res > 0) for (int sg = wave; sg < E.n_oseg; sg += NW) {
        if (as_global(E.oseg_kind)[sg] != 1) continue;
        const int src = as_global(E.oseg_src)[sg], dst = as_global(E.oseg_dst)[sg], n = as_global(E.oseg_n)[sg];
        double* const ob = uniform_ptr(io.obs + b * E.nobs + dst);
        for (int j = lane; j < n; j += WAVE) st_at(ob, (unsigned)j, R[src + j]);
      }
      blk_sync<NW>();
      if (wave == 0) {
      {
        if (lane < E.nc) {
          const int g = lane;
          const double* a = L.acc + 5 * g;
          const double cnt = a[4];
          double viol = as_global(E.con_worst)[g] ? a[2] + a[3] : a[0] + a[1];                  // :113-122, :93-98
          const double as = as_global(E.con_autoscale)[g];
          if (as != 0.0) viol *= as;                                                 // :82-83
          const double pw = as_global(E.con_ppow)[g];
          double pen = (pw == 1.0 ? viol : pow(viol, pw)) * as_global(E.con_pfac)[g];
          pen += cnt * as_global(E.con_cpen)[g];                                                // :124-128
          valid_acc = valid_acc && (cnt == 0.0);
          viol_acc += viol;
          pen_acc += -pen;
        }
      }
      OPFX_STAMP(7);
      if (c == 0) {
        // ---- cost rows fed by the solve: ext-grid P/Q, generator Q (objective.py:50-52) -----------
        const double* r_pe = R + 2 * nb + P.nbr;
        const double* r_qe = r_pe + P.nref;
        const double* r_qg = r_qe + P.nref;
        OPFX_REP(9) for (int r = E.ncost_pre + lane; r < E.ncost; r += WAVE) {
          const int meta = as_global(E.cost_meta)[r], pi = as_global(E.cost_psrc)[r];
          double pw_, qv_;
          if ((meta & 15) == OPFX_COST_EXT_GRID) { pw_ = r_pe[pi]; qv_ = r_qe[pi]; }
          else {                                               // generator: zero power on a de-energised bus (results_gen.py)
            pw_ = (!NOMOD && L.bt[pi] == BT_DEAD) ? 0.0 : src_val_g(xr, L.sp, as_global(E.cost_qsrc)[r]) * as_global(E.cost_scale)[r];
            qv_ = r_qg[pi];
          }
          if (E.cost_res) {                                    // (a unit that shares its bus with other generators: its own share)
            const int2 rr = E.cost_res[r];
            if (rr.x >= 0) pw_ = R[rr.x];
            if (rr.y >= 0) qv_ = R[rr.y];
          }
          csum += cost_row(E, xr, meta, as_global(E.cost_cbase)[r], pw_, qv_);
        }
        if (!NOMOD && isl != 0 && E.cost_bus) {
          // rare: this instance has a de-energised island.  Units on it report zero power (results_bus.py), so
          // the rows the prologue evaluated from their set-points are replaced by rows at zero power — on the
          // lane that added them, from the instance's row in global memory (the staged copy is gone)
          for (int r = lane; r < E.ncost_pre; r += WAVE) {
            const int bus = as_global(E.cost_bus)[r];
            if (bus < 0 || L.bt[bus] != BT_DEAD) continue;
            const int meta = as_global(E.cost_meta)[r], cbase = as_global(E.cost_cbase)[r];
            const double scl_ = as_global(E.cost_scale)[r];
            csum -= cost_row(E, xr, meta, cbase, src_val_g(xr, L.sp, as_global(E.cost_psrc)[r]) * scl_,
                             src_val_g(xr, L.sp, as_global(E.cost_qsrc)[r]) * scl_);
            csum += cost_row(E, xr, meta, cbase, 0.0, 0.0);
          }
        }
        for (int k = lane; k < E.n_qterm; k += WAVE) {            // objective_function seam: w (result - target)^2
          const double dv = R[as_global(E.qterm_idx)[k]] - as_global(E.qterm_target)[k];
          csum += as_global(E.qterm_weight)[k] * dv * dv;
        }
        objective = -(cost_pre + wave_sum_dpp(csum));                                    // opf_env.py:500
        if (E.diff_objective && io.initial_obj) objective -= io.initial_obj[b];      // :497-498
      }
      OPFX_STAMP(8);
      }
      blk_sync<NW>();
    }
    OPFX_STAMP(9);
    int nxt_v = 0;                            // (the queue's answer arrives while the reward and the flags are written)
    if (io.queued && tid == 0) nxt_v = __hip_atomic_fetch_add(P.queue, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // ---- reward (opf_env.py:515-530, reward.py:61-98) --------------------------
    if (wave == 0) {
    if (!conv0) {
      // opf_env.py:390-399: NaN observation and reward, terminated, all-invalid info
      __builtin_amdgcn_s_waitcnt(0);      // the table observations written above are overwritten
      if (io.obs) { double* const ob = uniform_ptr(io.obs + b * E.nobs); for (int k = lane; k < E.nobs; k += WAVE) st_at(ob, (unsigned)k, NaN); }
      if (lane < E.nc) {
        if (io.valids) st_at(uniform_ptr(io.valids + b * E.nc), (unsigned)lane, (unsigned char)0);
        if (io.violations) st_at(uniform_ptr(io.violations + b * E.nc), (unsigned)lane, 1.0);
        if (io.penalties) st_at(uniform_ptr(io.penalties + b * E.nc), (unsigned)lane, 1.0);
      }
      if (lane == 0) {
        if (io.reward) io.reward[b] = NaN;
        if (io.cost) io.cost[b] = NaN;
        if (io.objective) io.objective[b] = NaN;
        if (io.terminated) io.terminated[b] = 1;
        if (io.truncated) io.truncated[b] = 0;
      }
    } else {
      double pen_l = lane < E.nc ? pen_acc : 0.0;
      int inval_l = lane < E.nc ? !valid_acc : 0;
      const double penalty = wave_sum_dpp(pen_l);
      const bool valid = !wave_any(inval_l);
      if (lane < E.nc) {
        if (io.valids) st_at(uniform_ptr(io.valids + b * E.nc), (unsigned)lane, (unsigned char)(valid_acc ? 1 : 0));
        if (io.violations) st_at(uniform_ptr(io.violations + b * E.nc), (unsigned)lane, viol_acc);
        if (io.penalties) st_at(uniform_ptr(io.penalties + b * E.nc), (unsigned)lane, pen_acc);
      }
      if (lane == 0) {
        double obj = objective, pen = penalty;
        double cost_extra = 0.0;
        if (E.reward_kind == OPFX_REWARD_REPLACEMENT) {
          obj = valid ? obj + E.valid_reward : 0.0;                                  // reward.py:247-252
        } else if (E.reward_kind == OPFX_REWARD_PARAMETERIZED) {
          pen = valid ? pen + E.valid_reward : pen - E.invalid_penalty;              // :288-291
          if (!valid) obj *= E.invalid_objective_share;                              // :293-298
          cost_extra = E.invalid_penalty;
        } else if (E.reward_kind == OPFX_REWARD_ONLY_OBJECTIVE) {
          pen = 0.0;                                                                 // :316-317
        }
        obj = obj * E.objective_factor + E.objective_bias;                           // :83-86
        pen = pen * E.penalty_factor + E.penalty_bias;                               // :88-91
        double rew;
        if (E.penalty_weight != E.penalty_weight) rew = obj + pen;                   // :79-80
        else rew = obj * (1.0 - E.penalty_weight) + pen * E.penalty_weight;          // :81
        if (E.clip_lo == E.clip_lo) rew = fmin(fmax(rew, E.clip_lo), E.clip_hi);     // :70-71
        if (E.clipped_action_penalty != 0.0 && io.mode == 0)
          rew -= corr * E.clipped_action_penalty;                                    // opf_env.py:403-404
        if (io.reward) io.reward[b] = rew;
        if (io.cost) io.cost[b] = valid ? 0.0 : fabs(penalty * E.penalty_factor) + cost_extra;  // :93-98,301-305
        if (io.objective) io.objective[b] = objective;
        // (the counter is read only where it decides something: the load would be waited for together with every result and
        //  observation store issued above — one vmcnt counts both)
        const int spe = E.steps_per_episode;
        const int sie = (spe != 1 && io.step_in_episode) ? io.step_in_episode[b] : 1;
        unsigned char term = 0, trunc = 0;
        if (spe == 1) term = 1;                                                      // opf_env.py:406-414
        else if (sie >= spe) trunc = 1;
        if (io.terminated) io.terminated[b] = term;
        if (io.truncated) io.truncated[b] = trunc;
      }
    }
    if (lane == 0) {
      if (io.converged) io.converged[b] = conv0 ? 1 : 0;
      if (io.iterations) io.iterations[b] = iters0;
      if (io.max_mismatch) io.max_mismatch[b] = nrm0;
      if (io.mean_correction) io.mean_correction[b] = corr;
      if (io.total_iterations) io.total_iterations[b] = iters_all;
      if (io.min_pivot) io.min_pivot[b] = min_piv;
      if (io.min_pivot_bus) io.min_pivot_bus[b] = min_piv_bus;
    }
    }
    b = next_instance<NW>(L.acc + env_nacc(E.nc) - 2, b, io.queued != 0, nxt_v);
#ifdef OPFX_ENABLE_STAMPS
    if (P.stamps && tid == 0 && E.n_cont == 0) {      // developer probe: when each workgroup finished what (its warm row is free)
      double* w = P.warm + (size_t)blockIdx.x * 2 * nb;
      if (w[1] == 0.0) { w[3] = t_start__; w[4] = (double)__builtin_amdgcn_s_getreg(GETREG_HW_ID); w[5] = (double)__builtin_amdgcn_s_getreg(GETREG_XCC_ID); }   // HW_ID, XCC_ID
      w[0] = (double)wall_clock64(); w[1] += 1.0; w[2] += (double)iters0;
    }
#endif
  }
}

// ---------------------------------------------------------------------------
// instantiation tables of the kernel translation units (k_*.hip; declared in opfx_kernels.h)
// ---------------------------------------------------------------------------
// A translation unit instantiates its kernels by naming them here and hands their host-side handles (what hipLaunchKernel
// takes) to opfx.hip as plain pointers; nullptr = no such instantiation.  v2: block storage (0 first-generation kernel,
// 1 four-value, 2 two-value); team: wavefronts per instance; minw: wavefronts per SIMD the kernel is compiled for.
template <class K> const void* kernel_handle(K k) { return reinterpret_cast<const void*>(k); }

// the plain (DC = false) or DC-started step kernels of one SPEC: 2 storages x 3 teams at two wavefronts per SIMD, the team of
// four on two-value blocks and (plain only) the single wave compiled for three wavefronts per SIMD
template <bool DC, int SPEC>
const void* step_kernels(int v2, int team, int minw) {
  if (v2 != 1 && v2 != 2) return nullptr;
  if (minw == 3) {
    if (team == 4 && v2 == 2) return kernel_handle(k_step<2, 4, DC, false, false, SPEC, 3>);
    if constexpr (!DC) {
      if (team == 1) return v2 == 2 ? kernel_handle(k_step<2, 1, false, false, false, SPEC, 3>) : kernel_handle(k_step<1, 1, false, false, false, SPEC, 3>);
    }
    return nullptr;
  }
  if (minw != OPFX_MIN_WAVES_PER_SIMD) return nullptr;
  if (v2 == 2) return team == 4 ? kernel_handle(k_step<2, 4, DC, false, false, SPEC>) : team == 2 ? kernel_handle(k_step<2, 2, DC, false, false, SPEC>)
                    : team == 1 ? kernel_handle(k_step<2, 1, DC, false, false, SPEC>) : nullptr;
  return team == 4 ? kernel_handle(k_step<1, 4, DC, false, false, SPEC>) : team == 2 ? kernel_handle(k_step<1, 2, DC, false, false, SPEC>)
       : team == 1 ? kernel_handle(k_step<1, 1, DC, false, false, SPEC>) : nullptr;
}

}  // namespace

#endif  // OPFX_DEV_H
