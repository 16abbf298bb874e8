/*
 * opfx_debug.h — DEVELOPER entry points of libopfx.  Not part of the drop-in boundary (include/opfx.h); nothing a
 * binding of the reference needs is declared here.
 *
 * Until version 0.1 the library read sixteen OPFX_* environment variables (which kernel, which plan, which
 * scheduling) although opfx.h promised "no hidden global state".  From 0.2 on it reads none: the same switches
 * travel in an explicit `opfx_debug_opts` handed to the *_debug variants of opfx_plan_create / opfx_ctx_create, and a
 * context keeps its copy for everything created on it (environments, launches).  opfx_plan_create(c, out) is
 * opfx_plan_create_debug(c, NULL, out); likewise opfx_ctx_create.  Every member: 0 = the library's own choice.
 *
 * Users: tests (forcing wave teams of 2 / 4, the memory-resident kernels, the first-generation kernel, the work
 * queue on or off, the reset kernel's smaller teams on grids that would not take them), A/B scripts under scripts/, the
 * cycle-stamp probes.  The Python binding (opfgym_amd/capi.py: debug_from_env) fills the struct from OPFX_* variables
 * of ITS process for convenience — that is a feature of the binding's test harness, the C library itself stays
 * environment-free.
 */
#ifndef OPFX_DEBUG_H
#define OPFX_DEBUG_H

#include "opfx.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct opfx_debug_opts {
  uint32_t struct_size;      /* = sizeof(opfx_debug_opts)                                                     */
  /* ---- plan (opfx_plan_create_debug) ---------------------------------------------------------------------- */
  int32_t plan_search;       /* 0: default (15 to 63 extra tie-breaking rules, by size, on grids of 200-800 buses); n > 0: n extra
                              * rules, on any grid; -1: the first rule only                                     */
  int32_t plan_dcap_slack;   /* > 0: pin the degree slack of the level-scheduled minimum-degree ordering       */
  int32_t plan_seed;         /* > 0: pin the tie-breaking hash seed (pins the rule like plan_dcap_slack)        */
  int32_t plan_no_bank;      /* 1: keep the plan's own item order (no bank-aware dealing)                       */
  int32_t plan_no_pack;      /* 1: every block keeps four values                                                */
  int32_t plan_no_riders;    /* 1: forward-substitution terms as items of their own                             */
  int32_t plan_no_tail;      /* 1: no register chain for the dense tail                                         */
  int32_t plan_no_pairs;     /* 1: one update term per factor item (no second column of the same multiplier);
                              * 2: a second column wherever two terms share a multiplier (default: per level and
                              * stream, where it saves a round per wavefront)                                    */
  /* ---- context (opfx_ctx_create_debug): kept by the context, read by everything created on it -------------- */
  int32_t team;              /* 1 / 2 / 4: wavefronts per instance                                              */
  int32_t queue;             /* 1: work queue always, -1: fixed shares always                                   */
  int32_t packed;            /* 1: two-value block storage whenever it fits, -1: never                          */
  int32_t force_mem;         /* 1: the memory-resident form of the wave-team kernels on any grid                */
  int32_t kernel_v1;         /* 1: the first-generation kernel (plan walked through index arrays)               */
  int32_t waves_per_cu;      /* > 0: resident workgroups per CU instead of the occupancy query                  */
  int32_t verbose;           /* 1: launch geometry on stderr                                                    */
  int32_t stamps;            /* 1: allocate the cycle-stamp buffer (builds with -DOPFX_ENABLE_STAMPS fill it)   */
  int32_t reset_team;        /* 1 / 2 / 4: rows per workgroup of the reset kernel, when smaller than the default */
  /* ---- plan, appended in round 5 ------------------------------------------------------------------------------ */
  int32_t plan_share_slots;  /* 1: lower LU blocks that are dead give their LDS slot to fill blocks born later (plan.cpp share_slots;
                              * opfx_plan_info.n_shared): less LDS per instance, for the wave-team kernels with full Newton only.
                              * 0 (default): every block keeps a slot of its own.  opfgym_amd.BatchedOpfEnv builds such a plan
                              * where it lets a CU hold a third instance of the grid */
  /* ---- context, appended in round 6 ---------------------------------------------------------------------------- */
  int32_t no_rank1_dc;       /* 1: every N-1 contingency that starts from a DC power flow runs its own DC pass through the block-LU
                              * schedule (round 5's path) instead of the rank-1 update of the base case's DC angles (A/B runs, and the
                              * tests of the fallback that a base case with modifiers takes anyway) */
} opfx_debug_opts;

int opfx_plan_create_debug(const opfx_case* c, const opfx_debug_opts* dbg, opfx_plan** out);
int opfx_ctx_create_debug(const opfx_plan* p, int device, const opfx_debug_opts* dbg, opfx_ctx** out);

/* Cycle-stamp probes (diagnostic build, __graft_entry__.build_stamps): copy and clear the 32 per-phase cycle sums of
 * workgroup 0 / the per-workgroup finish records of the last step launches (six doubles per workgroup: wall clock at
 * its last instance, instances, Newton iterations, wall clock at its start, HW_ID, XCC_ID). */
int opfx_debug_read_stamps(opfx_ctx* ctx, unsigned long long* out32);
int opfx_debug_read_finish(opfx_ctx* ctx, double* out6, int n_wg);

#ifdef __cplusplus
}
#endif
#endif /* OPFX_DEBUG_H */
