// libopfx host side: contexts, environments, kernel selection and launches behind the C ABI (include/opfx.h).
// The kernels themselves live in opfx_dev.h and are instantiated by the translation units k_*.hip (opfx_kernels.h),
// compiled in parallel by __graft_entry__.build().
#include "opfx_dev.h"
#include "opfx_kernels.h"

// Weak fallbacks of the kernel accessors (opfx_kernels.h): a build that leaves a kernel translation unit out links, and the
// launches that would need its kernels are refused (kernel_missing).
#ifndef OPFX_SINGLE_TU        // (a probe build that #includes every kernel unit into this one has the strong definitions at hand)
#define OPFX_WEAK_KERNELS3(name) __attribute__((weak)) const void* name(int, int, int) { return nullptr; }
OPFX_WEAK_KERNELS3(opfx_k_step_plain0) OPFX_WEAK_KERNELS3(opfx_k_step_plain1) OPFX_WEAK_KERNELS3(opfx_k_step_plain2)
OPFX_WEAK_KERNELS3(opfx_k_step_plain3) OPFX_WEAK_KERNELS3(opfx_k_step_dc0) OPFX_WEAK_KERNELS3(opfx_k_step_dc1)
OPFX_WEAK_KERNELS3(opfx_k_step_dc2) OPFX_WEAK_KERNELS3(opfx_k_step_dc3) OPFX_WEAK_KERNELS3(opfx_k_step_other)
OPFX_WEAK_KERNELS3(opfx_k_solve)
__attribute__((weak)) const void* opfx_k_reset(int, int) { return nullptr; }
#undef OPFX_WEAK_KERNELS3
#endif

namespace {

using SolveKernel = void (*)(const DevPlan, SolveIO, Opts, long long);
using StepKernel = void (*)(const DevPlan, const DevEnv*, StepIO, Opts, long long);
using ResetKernel = void (*)(const DevReset*, const DevEnv*, ResetIO, long long, int);
template <class K> K as_kernel(const void* h) { return reinterpret_cast<K>(const_cast<void*>(h)); }
int kernel_missing(const char* who, const char* what) {
  opfx_set_error(std::string(who) + ": the kernel this launch needs (" + what + ") is not part of this build of libopfx");
  return OPFX_ERR_INVALID;
}

// ---------------------------------------------------------------------------
// host objects
// ---------------------------------------------------------------------------
// Device copies of the plan / environment arrays.  Small arrays are carved out of CHUNKS (one hipMalloc per 256 KB, 256-byte
// aligned pieces) instead of one hipMalloc each: a context holds ~55 arrays and an allocation costs at least a page table
// entry's worth of memory and tens of microseconds — 2.5 MB and 55 calls per context before, which is what the 256 cached
// plans of the batch-1 plug-in and the 64 topology twins of a bus-bus-switch environment multiplied
// (tests/test_gpu_footprint.py).  `ptrs`: buffers allocated on their own (scratch rows that grow, the probe's stamps).
struct DevArena {
  std::vector<void*> ptrs;
  std::vector<void*> chunks;
  char* cur = nullptr;
  size_t left = 0;
  static constexpr size_t CHUNK = 256 * 1024, ALIGN = 256;
  ~DevArena() { for (void* p : ptrs) (void)hipFree(p); for (void* p : chunks) (void)hipFree(p); }
  int take(size_t bytes, void** out) {
    bytes = (std::max<size_t>(bytes, 1) + ALIGN - 1) / ALIGN * ALIGN;
    if (bytes > left) {
      if (bytes >= CHUNK / 2) {              // a large array: its own allocation (the current chunk keeps its remainder)
        void* d = nullptr;
        HIP_TRY(hipMalloc(&d, bytes));
        chunks.push_back(d);
        *out = d;
        return OPFX_OK;
      }
      void* d = nullptr;
      HIP_TRY(hipMalloc(&d, CHUNK));
      chunks.push_back(d);
      cur = static_cast<char*>(d);
      left = CHUNK;
    }
    *out = cur;
    cur += bytes;
    left -= bytes;
    return OPFX_OK;
  }
  template <typename T>
  int put(const T* host, size_t n, const T** dev) {
    *dev = nullptr;
    void* d = nullptr;
    const int rc = take(n * sizeof(T), &d);
    if (rc != OPFX_OK) return rc;
    if (n) HIP_TRY(hipMemcpy(d, host, n * sizeof(T), hipMemcpyHostToDevice));
    *dev = static_cast<const T*>(d);
    return OPFX_OK;
  }
  template <typename T>
  int put(const std::vector<T>& v, const T** dev) { return put(v.data(), v.size(), dev); }
};

}  // namespace

struct opfx_ctx {
  opfx_debug_opts dbg{};               // developer switches (include/opfx_debug.h); all zero: the library's own choices
  int device = 0;
  int n_cu = 0;
  int solve_per_cu = 0;
  int solve_per_cu_dc = 0;
  int solve_per_cu_chord = 0;
  int solve_per_cu_mem = 0;
  double* blk_mem = nullptr;           // memory-resident kernels: LU block values, one row per resident workgroup
  size_t blk_mem_rows = 0;
  // per-workgroup scratch rows in global memory (L2), one per workgroup of the launch grid (ensure_scratch): scheduled P/Q
  // of the workgroup's instance, and — N-1 environments only — the base-case voltages its contingency solves start from
  double *pq = nullptr, *warm = nullptr;
  size_t pq_rows = 0, warm_rows = 0;
  opfx_plan plan;     // host copy
  DevPlan dp{};
  const DevPlan* d_dp = nullptr;   // device copy of dp (kernels take it by pointer)
  bool v2 = true;
  DevArena arena;
};

struct opfx_env {
  int reset_team = 1;                  // wavefronts (= rows) per workgroup of the reset kernel (opfx_env_set_reset)
  int reset_per_cu[2] = {0, 0};        // resident workgroups per CU of the plain / full reset kernel (cached)
  opfx_ctx* ctx = nullptr;
  DevEnv de{};
  const DevEnv* d_de = nullptr;
  DevReset dr{};
  const DevReset* d_dr = nullptr;      // device copy of dr (the reset kernel takes it by pointer)
  bool has_reset = false;

  DevArena arena;
  size_t lds_bytes = 0;
  int per_cu = 0;        // resident workgroups per CU of the kernel that ran last (report: opfx_env_get_info)
  int per_cu_spec[4] = {0, 0, 0, 0};   // ... cached per specialisation of the plain step kernel (SPEC)
  int spec = 0;          // SPEC bits the environment itself allows (opfx_env_create)
  int per_cu_dc[4] = {0, 0, 0, 0};     // (the same for the kernels compiled with the DC start)
  int per_cu_chord = 0;  // (and for those compiled with chord steps)
  bool mem = false;      // memory-resident step kernel (the LU blocks of this grid do not fit the LDS)
  int* queue = nullptr;  // this environment's own work-queue counter (two environments of a context do not share one)
  std::vector<int32_t> h_oseg[4];      // observation segments (kind, source, destination, length): host copy for the reset's element list
  int n_full = 0;        // four-value blocks this environment's kernels run with (choose_block_storage)
};

namespace {

size_t solver_lds_bytes(const opfx_plan& p, int na, int nres, bool v2, int nacc, int nmod, int n_full, bool mem = false) {
  const size_t nbe = ((size_t)p.nb + 1) & ~(size_t)1;
  const size_t bs = ((size_t)p.n_blk + 1) & ~(size_t)1, nfs = ((size_t)n_full + 1) & ~(size_t)1;
  // (memory-resident kernels: the block values are in global memory, the area only stages the table row / result bank)
  size_t blk = (std::max<size_t>(mem ? 0 : (v2 ? 2 * bs + 2 * nfs : 4 * bs), (size_t)nres) + 1) & ~(size_t)1;
  size_t d = (v2 ? 4 : 8) * nbe + blk + (size_t)na + (size_t)nacc + (size_t)12 * nmod;
  size_t bytes = d * sizeof(double) + (((size_t)p.nb + 1) & ~(size_t)1) +       // + bus types, diagonal block ids, tail table
                 (v2 ? 2 * ((((size_t)p.nb + 1) & ~(size_t)1) + p.tail_ids.size()) + (p.tail_ids.empty() ? 0 : 16) : 0);
  return (bytes + 15) & ~(size_t)15;
}

// Wavefronts per instance: grids whose LDS image leaves room for only a few instances per
// CU get a team of waves per instance so that the SIMDs are not left idle.
int instances_per_cu(size_t lds) {
  const size_t granule = 1024;
  return (int)((160 * 1024) / ((lds + granule - 1) / granule * granule));
}

// Two-value storage of the plain off-diagonal blocks (plan.cpp renumber_blocks) costs a divergent
// branch per block read; it is used only when the LDS it saves lets a CU hold more instances.
// Returns the LDS bytes per instance and the number of four-value blocks to run with.
template <typename F>
size_t choose_block_storage(const opfx_plan& p, const opfx_debug_opts& dbg, F lds_for, int* n_full_out) {
  const size_t full = lds_for(p.n_blk), packed = lds_for(p.n_full);
  bool use_packed = packed <= 160 * 1024 && (full > 160 * 1024 || instances_per_cu(packed) > instances_per_cu(full));
  if (dbg.packed) use_packed = dbg.packed > 0 && packed <= 160 * 1024;     // developer probe
  *n_full_out = use_packed ? p.n_full : p.n_blk;
  return use_packed ? packed : full;
}

// Instances from the context's work queue instead of fixed shares: from 12 instances per single-wave workgroup on,
// from 8 per wave team.  Measured against fixed shares WITH the priority turns of the kernels (profiles/r03_ab_queue.txt):
// single wave +4 % / +2 % / 0 / -1.5 % / -2.5 % / -4 % at 5 / 8 / 12 / 16 / 24 / 32 instances per wavefront; teams of four
// -0.4 % at 8, -0.9 % at 16 per team.  opfx_debug_opts.queue forces the choice (A/B runs).
int use_queue(const opfx_debug_opts& dbg, long long B, int grid, int team) {
  return dbg.queue ? dbg.queue > 0 : B >= (team > 1 ? 8LL : 12LL) * grid;
}

// shared: the plan shares LDS slots over time (plan.cpp share_slots): wave teams only (their items carry the zero stores)
int pick_team(const opfx_debug_opts& dbg, size_t lds, bool v2, bool shared = false) {
  if (!v2) return 1;
  if (dbg.team == 2 || dbg.team == 4 || (dbg.team == 1 && !shared)) return dbg.team;
  const size_t granule = 1024;
  const int inst = (int)((160 * 1024) / ((lds + granule - 1) / granule * granule));
  if (inst <= 2) return 4;
  if (inst <= 4 || shared) return 2;
  return 1;
}

template <typename K>
int launch_geometry(const opfx_debug_opts& dbg, K kernel, size_t lds, int n_cu, long long B, int* grid, int* per_cu_cache, int threads = WAVE) {
  if (*per_cu_cache > 0) {
    *grid = (int)std::max<long long>(1, std::min<long long>((long long)*per_cu_cache * n_cu, B));
    return OPFX_OK;
  }
  if (lds > 160 * 1024) {
    opfx_set_error("grid too large for the LDS-resident kernel (needs " + std::to_string(lds) + " B of LDS per instance)");
    return OPFX_ERR_TOO_LARGE;
  }
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int per_cu = 0;
  HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds));
  // The occupancy query can be one block high when the LDS request is not a multiple of the
  // allocation granule; an oversubscribed persistent grid runs its surplus blocks as a tail.
  {
    const size_t granule = 1024;
    const int by_lds = (int)((160 * 1024) / ((lds + granule - 1) / granule * granule));
    if (by_lds >= 1 && per_cu > by_lds) per_cu = by_lds;
  }
  if (per_cu < 1) per_cu = 1;
  if (per_cu > 16) per_cu = 16;       // the per-workgroup scratch rows (warm start, scheduled P/Q) are sized for 16 per CU
  if (dbg.waves_per_cu > 0) per_cu = dbg.waves_per_cu;     // developer override
  if (dbg.verbose) fprintf(stderr, "[opfx] lds=%zu B/instance, resident waves per CU=%d, CUs=%d\n", lds, per_cu, n_cu);
  *per_cu_cache = per_cu;
  long long g = (long long)per_cu * n_cu;
  *grid = (int)std::max<long long>(1, std::min<long long>(g, B));
  return OPFX_OK;
}

}  // namespace

// Memory-resident form (grids past the LDS): block values of every resident workgroup in one row of global memory.
static bool wants_mem(const opfx_debug_opts& dbg, size_t lds_resident, bool v2) {
  return v2 && (lds_resident > 160 * 1024 || dbg.force_mem != 0);    // (force_mem: developer / test switch)
}
static size_t blk_mem_stride(const opfx_plan& p) { return 4 * ((((size_t)p.n_blk + 1) & ~(size_t)1)); }
// Rows for `grid` workgroups (a batch-1 context needs ONE; a full launch n_cu x resident workgroups per CU <= 16).  Grown by
// doubling; a smaller buffer stays with the arena until the context goes (a launch in flight may still use it).
static int grow_rows(opfx_ctx* ctx, double** buf, size_t* have, size_t rows, size_t row_doubles, const char* what) {
  if (rows <= *have) return OPFX_OK;
  const size_t cap = (size_t)ctx->n_cu * 16;
  const size_t want = std::min(std::max(rows, 2 * *have), std::max(cap, rows));
  void* d = nullptr;
  if (hipMalloc(&d, want * row_doubles * sizeof(double)) != hipSuccess) { opfx_set_error(std::string("hipMalloc(") + what + ") failed"); return OPFX_ERR_HIP; }
  ctx->arena.ptrs.push_back(d);
  *buf = static_cast<double*>(d);
  *have = want;
  return OPFX_OK;
}
static int ensure_scratch(opfx_ctx* ctx, int grid, bool warm) {
  const size_t nbe = ((size_t)ctx->plan.nb + 1) & ~(size_t)1;
  int rc = grow_rows(ctx, &ctx->pq, &ctx->pq_rows, (size_t)grid, 2 * nbe, "P/Q scratch");
  if (rc == OPFX_OK && (warm || ctx->dbg.stamps)) rc = grow_rows(ctx, &ctx->warm, &ctx->warm_rows, (size_t)grid, 2 * (size_t)ctx->plan.nb, "warm-start scratch");
  return rc;
}
static int ensure_blk_mem(opfx_ctx* ctx, int rows) {
  if ((size_t)rows <= ctx->blk_mem_rows) return OPFX_OK;
  void* d = nullptr;
  const size_t bytes = (size_t)rows * blk_mem_stride(ctx->plan) * sizeof(double);
  if (hipMalloc(&d, bytes) != hipSuccess) {
    opfx_set_error("hipMalloc(block values of the memory-resident kernel, " + std::to_string(bytes >> 20) + " MiB) failed");
    return OPFX_ERR_HIP;
  }
  ctx->arena.ptrs.push_back(d);          // (an earlier, smaller buffer stays with the arena until the context goes)
  ctx->blk_mem = static_cast<double*>(d);
  ctx->blk_mem_rows = (size_t)rows;
  return OPFX_OK;
}

extern "C" int opfx_ctx_create(const opfx_plan* p, int device, opfx_ctx** out) { return opfx_ctx_create_debug(p, device, nullptr, out); }

extern "C" int opfx_ctx_create_debug(const opfx_plan* p, int device, const opfx_debug_opts* dbg_in, opfx_ctx** out) {
  if (!p || !out) { opfx_set_error("opfx_ctx_create: null argument"); return OPFX_ERR_INVALID; }
  opfx_debug_opts dbg{};
  if (dbg_in) { const int rc_ = opfx_take(dbg_in, &dbg, "opfx_ctx_create_debug(opfx_debug_opts)"); if (rc_ != OPFX_OK) return rc_; }
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) {
    opfx_set_error("opfx_ctx_create: no HIP device available");
    return OPFX_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= n_dev) { opfx_set_error("opfx_ctx_create: bad device ordinal"); return OPFX_ERR_INVALID; }
  HIP_TRY(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  auto* c = new opfx_ctx();
  c->dbg = dbg;
  c->device = device;
  c->n_cu = prop.multiProcessorCount;
  c->plan = *p;
  DevPlan& d = c->dp;
  d.nb = p->nb; d.nbr = p->nbr; d.nref = p->nref; d.nblk = p->n_blk; d.nlev = p->n_levels();
  d.nfill = (int)p->fill_blk.size(); d.npv = p->npv; d.base_mva = p->base_mva;
  d.n_shared = p->n_shared;
  DevArena& A = c->arena;
  int rc = OPFX_OK;
#define PUT(field, vec) if (rc == OPFX_OK) rc = A.put(p->vec, &d.field)
  PUT(bus_type, bus_type); PUT(y_ptr, y_ptr); PUT(y_col, y_col); PUT(y_blk, y_blk);
  PUT(diag_blk, diag_blk); PUT(fill_blk, fill_blk); PUT(lev_tptr, lev_tptr); PUT(tgt_blk, tgt_blk);
  PUT(tgt_sptr, tgt_sptr); PUT(src_ik, src_ik); PUT(src_kk, src_kk); PUT(src_kj, src_kj);
  PUT(lev_pptr, lev_pptr); PUT(piv_bus, piv_bus); PUT(piv_uptr, piv_uptr); PUT(u_blk, u_blk);
  PUT(u_col, u_col); PUT(br_f, br_f); PUT(br_t, br_t); PUT(br_pos, br_pos); PUT(br_island, br_island); PUT(isl_ptr, isl_ptr); PUT(isl_bus, isl_bus); PUT(ref_bus, ref_bus);
  PUT(ref_ord, ref_ord); PUT(vm_set, vm_set); PUT(va_set, va_set); PUT(y_g, y_g); PUT(y_b, y_b);
  PUT(br_y, br_y); PUT(br_kf, br_kf); PUT(br_kt, br_kt);
  d.nfull = p->n_full;
  d.fill_lo = p->n_full - (int)p->fill_blk.size();
  d.ra = p->ra; d.rh = p->rh; d.rb = p->rb_pad; d.rc = p->rc_pad;      // device: padded round counts of lp_bc
  d.stamps = nullptr;
  if (dbg.stamps) {
    void* st = nullptr;
    if (hipMalloc(&st, 32 * sizeof(unsigned long long)) == hipSuccess) { (void)hipMemset(st, 0, 32 * sizeof(unsigned long long)); A.ptrs.push_back(st); d.stamps = static_cast<unsigned long long*>(st); }
  }
  c->v2 = p->rb >= 0 && !dbg.kernel_v1;   // (kernel_v1: developer switch to the first-generation kernel)
  if (p->n_shared > 0 && !c->v2) { delete c; opfx_set_error("opfx_ctx_create: a plan with shared slots runs on the wave-team kernels only"); return OPFX_ERR_INVALID; }
  PUT(lp_bc, lp_bc); PUT(lp_apk, lp_apk); PUT(lp_hpk, lp_hpk); PUT(lp_hrows, lp_hrows);
  PUT(lp_bcc, lp_bcc);
  d.rf = p->rf_pad;
  d.lp_dc = nullptr; d.lp_hdc = nullptr; d.br_bdc = nullptr; d.br_pfinj = nullptr;
  d.blk_mem = nullptr; d.blk_mem_stride = 0;
  if (!p->lp_dc.empty()) { PUT(lp_dc, lp_dc); PUT(lp_hdc, lp_hdc); PUT(br_bdc, br_bdc); PUT(br_pfinj, br_pfinj); }
  d.n_hrows = (int)p->lp_hrows.size();
  if (rc == OPFX_OK) rc = A.put(p->lp_team[0], &d.lp_team2);
  if (rc == OPFX_OK) rc = A.put(p->lp_team[1], &d.lp_team4);
  if (rc == OPFX_OK) rc = A.put(p->lp_teamc[0], &d.lp_teamc2);
  if (rc == OPFX_OK) rc = A.put(p->lp_teamc[1], &d.lp_teamc4);
  d.team_roundsc2 = p->team_rounds_c[0]; d.team_roundsc4 = p->team_rounds_c[1];
  d.team_kbc2 = p->team_kb_c[0]; d.team_kbc4 = p->team_kb_c[1];
  d.team_rounds2 = p->team_rounds[0]; d.team_rounds4 = p->team_rounds[1];
  d.team_kb2 = p->team_kb[0]; d.team_kb4 = p->team_kb[1];
  d.tail_m = p->tail_m; d.tail_n = (int)p->tail_ids.size();
  if (rc == OPFX_OK) rc = A.put(p->tail_bus, &d.tail_bus);
  if (rc == OPFX_OK) rc = A.put(p->tail_ids, &d.tail_ids);
  d.warm = nullptr; d.pq = nullptr;          // (per-workgroup scratch rows: sized by the launch grid, ensure_scratch)
  {
    void* qu = nullptr;
    if (rc == OPFX_OK) rc = A.take(2 * sizeof(int), &qu);
    if (rc == OPFX_OK && hipMemset(qu, 0, 2 * sizeof(int)) != hipSuccess) { delete c; opfx_set_error("hipMemset(work queue) failed"); return OPFX_ERR_HIP; }
    d.queue = static_cast<int*>(qu);
  }
  {
    std::vector<double> vr0(p->nb), vi0(p->nb);
    for (int i = 0; i < p->nb; ++i) { vr0[i] = p->vm_set[i] * std::cos(p->va_set[i]); vi0[i] = p->vm_set[i] * std::sin(p->va_set[i]); }
    if (rc == OPFX_OK) rc = A.put(vr0, &d.vr0);
    if (rc == OPFX_OK) rc = A.put(vi0, &d.vi0);
  }
#undef PUT
  if (rc == OPFX_OK) rc = A.put(&c->dp, 1, &c->d_dp);
  if (rc != OPFX_OK) { delete c; return rc; }
  *out = c;
  return OPFX_OK;
}

extern "C" void opfx_ctx_destroy(opfx_ctx* ctx) { delete ctx; }

extern "C" int opfx_solve(opfx_ctx* ctx, int64_t B, const double* p_inj, const double* q_inj,
                          const double* qg_min, const double* qg_max, const int32_t* outage,
                          const opfx_solve_opts* opts, double* vm, double* va, double* loading,
                          double* s_ref, double* q_gen, uint8_t* converged, int32_t* iterations,
                          double* max_mismatch, double* min_pivot, int32_t* min_pivot_bus, void* stream) {
  if (ctx && B == 0) return OPFX_OK;                  // empty batch: nothing to do (its buffers may be null)
  if (!ctx || !p_inj || !q_inj || B < 0) { opfx_set_error("opfx_solve: bad argument"); return OPFX_ERR_INVALID; }
  opfx_solve_opts so;
  if (opts) { const int rc_ = opfx_take(opts, &so, "opfx_solve(opfx_solve_opts)"); if (rc_ != OPFX_OK) return rc_; opts = &so; }
  HIP_TRY(hipSetDevice(ctx->device));
  Opts o{opts ? opts->tol : 1e-8, opts ? opts->max_iter : 10, opts ? opts->enforce_q_lims : 0, 0, opts ? opts->init : 0,
         opts ? opts->jacobian_reuse_tol : 0.0};
  if (!(o.reuse_tol >= 0.0)) { opfx_set_error("opfx_solve: jacobian_reuse_tol must be >= 0"); return OPFX_ERR_INVALID; }
  if (o.init == OPFX_INIT_DC && !ctx->dp.lp_dc) { opfx_set_error("opfx_solve: init = OPFX_INIT_DC needs a case with br_bdc / br_pfinj"); return OPFX_ERR_INVALID; }
  if (o.enforce_q_lims && (!qg_min || !qg_max)) o.enforce_q_lims = 0;
  const int nres = 3 * ctx->plan.nb + ctx->plan.nbr + 2 * ctx->plan.nref;
  int n_full = ctx->plan.n_blk;
  size_t lds = choose_block_storage(ctx->plan, ctx->dbg, [&](int nf) { return solver_lds_bytes(ctx->plan, 0, nres, ctx->v2, 8, 1, nf); }, &n_full);
  DevPlan dp = ctx->dp;
  if (wants_mem(ctx->dbg, lds, ctx->v2)) {
    if (ctx->plan.n_shared > 0) { opfx_set_error("opfx_solve: a plan with shared slots does not run on the memory-resident kernels"); return OPFX_ERR_INVALID; }
    // the LU blocks do not fit the LDS beside the state vectors: wave team of four, four-value blocks in global memory
    n_full = ctx->plan.n_blk;
    lds = solver_lds_bytes(ctx->plan, 0, nres, true, 8, 1, n_full, true);
    dp.nfull = n_full;
    int grid = 0;
    const SolveKernel kern = as_kernel<SolveKernel>(opfx_k_solve(OPFX_KS_MEM, 1, 4));
    if (!kern) return kernel_missing("opfx_solve", "k_solve, memory-resident blocks");
    int rc = launch_geometry(ctx->dbg, kern, lds, ctx->n_cu, B, &grid, &ctx->solve_per_cu_mem, WAVE * 4);
    if (rc != OPFX_OK) return rc;
    rc = ensure_blk_mem(ctx, ctx->solve_per_cu_mem * ctx->n_cu);
    if (rc != OPFX_OK) return rc;
    dp.blk_mem = ctx->blk_mem; dp.blk_mem_stride = (long long)blk_mem_stride(ctx->plan);
    if (o.init == OPFX_INIT_DC) o.init = OPFX_INIT_FLAT;          // (no DC start in the memory-resident form)
    o.reuse_tol = 0.0;                                            // (nor chord steps)
    rc = ensure_scratch(ctx, grid, false);
    if (rc != OPFX_OK) return rc;
    dp.pq = ctx->pq; dp.warm = ctx->warm;
    SolveIO io{p_inj, q_inj, qg_min, qg_max, outage, vm, va, loading, s_ref, q_gen, max_mismatch, converged, iterations, min_pivot, min_pivot_bus, 0};
    if ((io.queued = use_queue(ctx->dbg, B, grid, 4))) HIP_TRY(hipMemsetAsync(ctx->dp.queue, 0, sizeof(int), static_cast<hipStream_t>(stream)));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVE * 4), lds, static_cast<hipStream_t>(stream), dp, io, o, (long long)B);
    HIP_TRY(hipGetLastError());
    return OPFX_OK;
  }
  dp.nfull = n_full;
  int grid = 0;
  const int team = pick_team(ctx->dbg, lds, ctx->v2, ctx->plan.n_shared > 0);
  const bool packed = n_full < ctx->plan.n_blk;
  const int v2s = packed ? 2 : 1;
  SolveKernel kern = as_kernel<SolveKernel>(!ctx->v2 ? opfx_k_solve(OPFX_KS_V1, 0, 1) : opfx_k_solve(OPFX_KS_PLAIN, v2s, team));
  if (!kern) return kernel_missing("opfx_solve", "k_solve");
  int rc = launch_geometry(ctx->dbg, kern, lds, ctx->n_cu, B, &grid, &ctx->solve_per_cu, WAVE * team);
  if (rc != OPFX_OK) return rc;
  if (ctx->plan.n_shared > 0) o.reuse_tol = 0.0;      // (chord iterations re-read lower blocks whose slots have new tenants by then)
  if (o.init == OPFX_INIT_DC && ctx->v2) {           // the kernels compiled with the DC start (same launch geometry)
    kern = as_kernel<SolveKernel>(opfx_k_solve(OPFX_KS_DC, v2s, team));
    if (!kern) return kernel_missing("opfx_solve", "k_solve, DC start");
    int per_cu_dc = ctx->solve_per_cu_dc;
    rc = launch_geometry(ctx->dbg, kern, lds, ctx->n_cu, B, &grid, &per_cu_dc, WAVE * team);
    ctx->solve_per_cu_dc = per_cu_dc;
    if (rc != OPFX_OK) return rc;
  } else if (o.reuse_tol > 0.0 && ctx->v2) {         // the kernels compiled with chord steps (not together with the DC start)
    kern = as_kernel<SolveKernel>(opfx_k_solve(OPFX_KS_CHORD, v2s, team));
    if (!kern) return kernel_missing("opfx_solve", "k_solve, chord steps");
    int per_cu_c = ctx->solve_per_cu_chord;
    rc = launch_geometry(ctx->dbg, kern, lds, ctx->n_cu, B, &grid, &per_cu_c, WAVE * team);
    ctx->solve_per_cu_chord = per_cu_c;
    if (rc != OPFX_OK) return rc;
  }
  if (o.init == OPFX_INIT_DC || !ctx->v2 || ctx->plan.n_shared > 0) o.reuse_tol = 0.0;
  rc = ensure_scratch(ctx, grid, false);
  if (rc != OPFX_OK) return rc;
  dp.pq = ctx->pq; dp.warm = ctx->warm;
  SolveIO io{p_inj, q_inj, qg_min, qg_max, outage, vm, va, loading, s_ref, q_gen, max_mismatch, converged, iterations, min_pivot, min_pivot_bus, 0};
  if ((io.queued = use_queue(ctx->dbg, B, grid, team))) HIP_TRY(hipMemsetAsync(ctx->dp.queue, 0, sizeof(int), static_cast<hipStream_t>(stream)));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVE * team), lds, static_cast<hipStream_t>(stream), dp, io, o,
                     (long long)B);
  HIP_TRY(hipGetLastError());
  return OPFX_OK;
}

extern "C" int opfx_env_create(opfx_ctx* ctx, const opfx_env_desc* d_in, opfx_env** out) {
  if (!ctx || !d_in || !out) { opfx_set_error("opfx_env_create: null argument"); return OPFX_ERR_INVALID; }
  opfx_env_desc desc;
  { const int rc_ = opfx_take(d_in, &desc, "opfx_env_create(opfx_env_desc)", offsetof(opfx_env_desc, xres_offset)); if (rc_ != OPFX_OK) return rc_; }   // (0.3.0 ended before xres_offset)
  const opfx_env_desc* const d = &desc;
  HIP_TRY(hipSetDevice(ctx->device));
  const opfx_plan& p = ctx->plan;
  const int nb = p.nb;
  if (d->nx <= 0 || d->na < 0 || d->nc < 0 || d->nc > WAVE || d->nobs < 0 || !d->pinj_ptr || !d->qinj_ptr) {
    opfx_set_error("opfx_env_create: inconsistent descriptor");
    return OPFX_ERR_INVALID;
  }
  auto* e = new opfx_env();
  e->ctx = ctx;
  DevEnv& E = e->de;
  E.nx = d->nx; E.na = d->na; E.npoly = d->npoly; E.npwl = d->npwl; E.nseg = d->nseg;
  E.nc = d->nc; E.nobs = d->nobs; E.ncost = d->npoly + d->npwl;
  E.nres_base = 3 * nb + p.nbr + 2 * p.nref;
  E.n_xres = d->n_xres;
  E.nres = E.nres_base + d->n_xres;
  E.reward_kind = d->reward_kind; E.diff_objective = d->diff_objective;
  E.steps_per_episode = d->steps_per_episode; E.clamp_enabled = d->clamp_enabled;
  E.penalty_weight = d->penalty_weight; E.clip_lo = d->clip_lo; E.clip_hi = d->clip_hi;
  E.objective_factor = d->objective_factor; E.objective_bias = d->objective_bias;
  E.penalty_factor = d->penalty_factor; E.penalty_bias = d->penalty_bias;
  E.valid_reward = d->valid_reward; E.invalid_penalty = d->invalid_penalty;
  E.invalid_objective_share = d->invalid_objective_share;
  E.diff_step = d->diff_action_step_size; E.clipped_action_penalty = d->clipped_action_penalty;
  E.n_cont = d->n_cont; E.not_converged_penalty = d->not_converged_penalty;
  E.n_bmod = d->n_bmod;
  E.max_mod = d->n_bmod + 2;
  if (d->n_bmod > 0 && !ctx->v2) { delete e; opfx_set_error("opfx_env_create: branch state columns need the lane-programme kernels"); return OPFX_ERR_INVALID; }
  if (d->n_bmod > 24) { delete e; opfx_set_error("opfx_env_create: at most 24 branch state columns"); return OPFX_ERR_INVALID; }
  // slot -> action map: a column written by an action is read from the set-point, not from x
  std::vector<int32_t> slot_act(d->nx, -1);
  for (int k = 0; k < d->na; ++k) {
    if (d->act_slot[k] < 0 || d->act_slot[k] >= d->nx) { delete e; opfx_set_error("opfx_env_create: act_slot out of range"); return OPFX_ERR_INVALID; }
    slot_act[d->act_slot[k]] = k;
  }
  auto act_of = [&](const int32_t* slots, size_t n) {
    std::vector<int32_t> v(n, -1);
    for (size_t i = 0; i < n; ++i) if (slots[i] >= 0 && slots[i] < d->nx) v[i] = slot_act[slots[i]];
    return v;
  };
  DevArena& A = e->arena;
  int rc = OPFX_OK;
  const size_t np_ = d->pinj_ptr[nb], nq_ = d->qinj_ptr[nb];
#define PUTN(field, ptr, n) if (rc == OPFX_OK) rc = A.put(ptr, (size_t)(n), &E.field)
  auto src_of = [&](int32_t slot) { return slot < 0 ? NOSRC : (slot_act[slot] >= 0 ? ~slot_act[slot] : slot); };
  {
    // flat injection list sorted by column; one 16-byte record per entry
    struct Inj { int32_t slot, bus; double coef; };
    std::vector<Inj> inj;
    for (int i = 0; i < nb; ++i) {
      for (int e = d->pinj_ptr[i]; e < d->pinj_ptr[i + 1]; ++e) inj.push_back({d->pinj_slot[e], i, d->pinj_coef[e]});
      for (int e = d->qinj_ptr[i]; e < d->qinj_ptr[i + 1]; ++e) inj.push_back({d->qinj_slot[e], i | (1 << 16), d->qinj_coef[e]});
    }
    std::stable_sort(inj.begin(), inj.end(), [](const Inj& a, const Inj& b) { return a.slot < b.slot; });
    std::vector<uint4> pk;
    for (auto& q : inj) {
      if (q.slot < 0 || q.slot >= d->nx) { delete e; opfx_set_error("opfx_env_create: injection slot out of range"); return OPFX_ERR_INVALID; }
      uint32_t w[2]; std::memcpy(w, &q.coef, 8);
      pk.push_back(make_uint4((uint32_t)q.bus, (uint32_t)src_of(q.slot), w[0], w[1]));
    }
    E.n_inj = (int)inj.size();
    if (rc == OPFX_OK) rc = A.put(pk, &E.inj_pk);
  }
  if (d->qg_min && d->qg_max) { PUTN(qg_min, d->qg_min, nb); PUTN(qg_max, d->qg_max, nb); }
  PUTN(act_slot, d->act_slot, d->na); PUTN(act_scaling, d->act_scaling, d->na);
  PUTN(act_lo_slot, d->act_lo_slot, d->na); PUTN(act_hi_slot, d->act_hi_slot, d->na);
  PUTN(act_lo_const, d->act_lo_const, d->na); PUTN(act_hi_const, d->act_hi_const, d->na);
  {
    // always present on the device (slot -2 = no clamp) so that the kernel loads them unconditionally
    std::vector<int32_t> cls_(d->na, -2), chs(d->na, -2);
    std::vector<double> clc(d->na, 0.0), chc(d->na, 0.0);
    if (d->clamp_enabled) {
      cls_.assign(d->clamp_lo_slot, d->clamp_lo_slot + d->na); chs.assign(d->clamp_hi_slot, d->clamp_hi_slot + d->na);
      clc.assign(d->clamp_lo_const, d->clamp_lo_const + d->na); chc.assign(d->clamp_hi_const, d->clamp_hi_const + d->na);
    }
    if (rc == OPFX_OK) rc = A.put(cls_, &E.clamp_lo_slot);
    if (rc == OPFX_OK) rc = A.put(chs, &E.clamp_hi_slot);
    if (rc == OPFX_OK) rc = A.put(clc, &E.clamp_lo_const);
    if (rc == OPFX_OK) rc = A.put(chc, &E.clamp_hi_const);
  }
  const size_t ncost = (size_t)d->npoly + d->npwl;
  const size_t ncoef = (size_t)d->npoly * 6 + (size_t)d->npwl * d->nseg * 3;
  PUTN(cost_coef, d->cost_coef, ncoef);
  {
    std::vector<int32_t> cx(ncoef, -1);
    for (int k = 0; k < d->nprice; ++k) {
      if (d->price_coef[k] < 0 || (size_t)d->price_coef[k] >= ncoef || d->price_slot[k] < 0 || d->price_slot[k] >= d->nx) {
        rc = OPFX_ERR_INVALID; opfx_set_error("opfx_env_create: price_coef/price_slot out of range"); break;
      }
      cx[d->price_coef[k]] = d->price_slot[k];
    }
    if (rc == OPFX_OK) rc = A.put(cx, &E.coef_xslot);
    std::vector<int32_t> meta, ps, qs, cb, cbus;
    std::vector<int2> cres;
    bool any_cres = false;
    std::vector<double> scl;
    for (int pass = 0; pass < 2; ++pass) {
      for (size_t r = 0; r < ncost; ++r) {
        const int kind = d->cost_kind[r];
        if ((kind == OPFX_COST_UNIT) != (pass == 0)) continue;
        const bool pwl = r >= (size_t)d->npoly;
        const size_t w = pwl ? r - d->npoly : 0;
        meta.push_back(kind | (pwl ? 16 : 0) | (pwl && d->pwl_is_q[w] ? 32 : 0));
        cb.push_back(pwl ? (int32_t)(d->npoly * 6 + w * d->nseg * 3) : (int32_t)(r * 6));
        scl.push_back(d->cost_scale[r]);
        {
          const int32_t pr = (d->cost_pres && kind == OPFX_COST_EXT_GRID) ? d->cost_pres[r] : -1;
          const int32_t qr = (d->cost_qres && kind != OPFX_COST_UNIT) ? d->cost_qres[r] : -1;
          if (pr >= E.nres || qr >= E.nres) { rc = OPFX_ERR_INVALID; opfx_set_error("opfx_env_create: cost_pres / cost_qres out of range"); }
          cres.push_back(make_int2(pr < 0 ? -1 : pr, qr < 0 ? -1 : qr));
          any_cres = any_cres || pr >= 0 || qr >= 0;
        }
        if (kind == OPFX_COST_UNIT) {
          ps.push_back(src_of(d->cost_pidx[r])); qs.push_back(src_of(d->cost_qidx[r]));
          const int32_t bus = d->cost_bus ? d->cost_bus[r] : -1;
          if (bus >= ctx->plan.nb) { rc = OPFX_ERR_INVALID; opfx_set_error("opfx_env_create: cost_bus out of range"); }
          cbus.push_back(bus);
        }
        else if (kind == OPFX_COST_GEN) { ps.push_back(d->cost_pidx[r]); qs.push_back(src_of(d->cost_qidx[r])); }   // gen: qidx = slot of p_mw
        else { ps.push_back(d->cost_pidx[r]); qs.push_back(NOSRC); }
      }
      if (pass == 0) E.ncost_pre = (int)meta.size();
    }
    if (rc == OPFX_OK) rc = A.put(meta, &E.cost_meta);
    if (rc == OPFX_OK) rc = A.put(ps, &E.cost_psrc);
    if (rc == OPFX_OK) rc = A.put(qs, &E.cost_qsrc);
    if (rc == OPFX_OK) rc = A.put(cb, &E.cost_cbase);
    if (rc == OPFX_OK) rc = A.put(scl, &E.cost_scale);
    E.cost_bus = nullptr;
    if (d->cost_bus && rc == OPFX_OK) rc = A.put(cbus, &E.cost_bus);
    E.cost_res = nullptr;
    if (any_cres && rc == OPFX_OK) rc = A.put(cres, &E.cost_res);
  }
  const size_t ncel = d->nc ? d->con_ptr[d->nc] : 0;
  {
    std::vector<int2> pk(ncel);
    for (int g = 0; g < d->nc; ++g) for (int e = d->con_ptr[g]; e < d->con_ptr[g + 1]; ++e) pk[e] = make_int2(d->con_src[e], g);
    E.ncel = (int)ncel;
    if (rc == OPFX_OK) rc = A.put(pk, &E.con_pk);
  }
  {
    // NaN bound = absent boundary: comparisons against it must be false
    std::vector<double> lo(d->con_min, d->con_min + ncel), hi(d->con_max, d->con_max + ncel);
    for (auto& v : lo) if (v != v) v = -INFINITY;
    for (auto& v : hi) if (v != v) v = INFINITY;
    if (rc == OPFX_OK) rc = A.put(lo, &E.con_min);
    if (rc == OPFX_OK) rc = A.put(hi, &E.con_max);
  }
  PUTN(con_autoscale, d->con_autoscale, d->nc); PUTN(con_pfac, d->con_penalty_factor, d->nc);
  PUTN(con_ppow, d->con_penalty_power, d->nc); PUTN(con_cpen, d->con_count_penalty, d->nc);
  PUTN(con_worst, d->con_worst_case, d->nc);
  {
    // observation as maximal contiguous runs: kind 0 = x[src..], 1 = result bank, 2 = action set-points
    std::vector<int32_t> sk, ss, sd, sn;
    E.need_angle = 0;
    for (int k = 0; k < d->nobs; ++k) {
      int kind = d->obs_kind[k] == OPFX_SRC_X ? 0 : 1, src = d->obs_idx[k];
      if (kind == 0 && src >= 0 && src < d->nx && slot_act[src] >= 0) { kind = 2; src = slot_act[src]; }
      if (kind == 1 && src >= nb && src < 2 * nb) E.need_angle = 1;
      if (!sk.empty() && sk.back() == kind && ss.back() + sn.back() == src && sd.back() + sn.back() == k) sn.back()++;
      else { sk.push_back(kind); ss.push_back(src); sd.push_back(k); sn.push_back(1); }
    }
    E.n_oseg = (int)sk.size();
    E.n_oseg_res = (int)std::count(sk.begin(), sk.end(), 1);
    e->h_oseg[0] = sk; e->h_oseg[1] = ss; e->h_oseg[2] = sd; e->h_oseg[3] = sn;
    if (rc == OPFX_OK) rc = A.put(sk, &E.oseg_kind);
    if (rc == OPFX_OK) rc = A.put(ss, &E.oseg_src);
    if (rc == OPFX_OK) rc = A.put(sd, &E.oseg_dst);
    if (rc == OPFX_OK) rc = A.put(sn, &E.oseg_n);
    for (size_t i = 0; i < ncel; ++i) if (d->con_src[i] >= nb && d->con_src[i] < 2 * nb) E.need_angle = 1;
  }
  PUTN(cont_branch, d->cont_branch, d->n_cont);
  E.cont_dc_w = nullptr; E.cont_dc_k = nullptr;
  if (rc == OPFX_OK && d->n_cont > 0 && !p.br_bdc.empty() && nb <= 1536) {
    // DC start of the contingencies by a rank-1 update (k_step): B' of the compiled grid (pypower makeBdc, the free buses),
    // factorised ONCE here on the host — it is the same matrix for every instance and step —, then w = B'^-1 u per
    // contingency (f, t), u = e_f - e_t on the free buses, and 1 / (1 - b u'w); 0 there marks a contingency that keeps
    // its own DC pass (no DC model of the branch, or a bridge: 1 - b u'w = 0 is the islanding case)
    std::vector<int> pos(nb, -1);
    int nf = 0;
    for (int i = 0; i < nb; ++i) if (p.bus_type[i] != OPFX_REF) pos[i] = nf++;
    std::vector<double> B((size_t)nf * nf, 0.0);
    for (int k = 0; k < p.nbr; ++k) {
      const double b = p.br_bdc[k];
      if (b == 0.0) continue;
      const int f = pos[p.br_f[k]], t = pos[p.br_t[k]];
      if (f >= 0) B[(size_t)f * nf + f] += b;
      if (t >= 0) B[(size_t)t * nf + t] += b;
      if (f >= 0 && t >= 0) { B[(size_t)f * nf + t] -= b; B[(size_t)t * nf + f] -= b; }
    }
    // dense LU with partial pivoting (nf <= 1536: at most a few 1e9 flops, once per environment)
    std::vector<int> perm(nf);
    bool regular = true;
    for (int i = 0; i < nf; ++i) perm[i] = i;
    for (int c = 0; c < nf && regular; ++c) {
      int piv = c;
      for (int r = c + 1; r < nf; ++r) if (std::fabs(B[(size_t)r * nf + c]) > std::fabs(B[(size_t)piv * nf + c])) piv = r;
      if (std::fabs(B[(size_t)piv * nf + c]) < 1e-12) { regular = false; break; }
      if (piv != c) { for (int j = 0; j < nf; ++j) std::swap(B[(size_t)c * nf + j], B[(size_t)piv * nf + j]); std::swap(perm[c], perm[piv]); }
      const double inv = 1.0 / B[(size_t)c * nf + c];
      for (int r = c + 1; r < nf; ++r) {
        const double m = B[(size_t)r * nf + c] * inv;
        if (m == 0.0) continue;
        B[(size_t)r * nf + c] = m;
        double* rr = &B[(size_t)r * nf];
        const double* rc_ = &B[(size_t)c * nf];
        for (int j = c + 1; j < nf; ++j) rr[j] -= m * rc_[j];
      }
    }
    if (regular) {
      std::vector<double> W((size_t)d->n_cont * nb, 0.0), K((size_t)d->n_cont * 4, 0.0), y(nf);
      for (int c = 0; c < d->n_cont; ++c) {
        const int br = d->cont_branch[c];
        if (br < 0 || br >= p.nbr) continue;
        const double b = p.br_bdc[br];
        const int f = pos[p.br_f[br]], t = pos[p.br_t[br]];
        if (b == 0.0 || (f < 0 && t < 0)) continue;
        for (int i = 0; i < nf; ++i) { const int src = perm[i]; y[i] = (src == f ? 1.0 : 0.0) - (src == t ? 1.0 : 0.0); }
        for (int i = 0; i < nf; ++i) { double v = y[i]; const double* r = &B[(size_t)i * nf]; for (int j = 0; j < i; ++j) v -= r[j] * y[j]; y[i] = v; }
        for (int i = nf - 1; i >= 0; --i) { double v = y[i]; const double* r = &B[(size_t)i * nf]; for (int j = i + 1; j < nf; ++j) v -= r[j] * y[j]; y[i] = v / r[i]; }
        const double s_ = (f >= 0 ? y[f] : 0.0) - (t >= 0 ? y[t] : 0.0);
        const double den = 1.0 - b * s_;
        if (std::fabs(den) < 1e-9) continue;
        for (int i = 0; i < nb; ++i) if (pos[i] >= 0) W[(size_t)c * nb + i] = y[pos[i]];
        K[(size_t)c * 4] = b; K[(size_t)c * 4 + 1] = p.br_pfinj[br]; K[(size_t)c * 4 + 2] = 1.0 / den;
      }
      if (rc == OPFX_OK) rc = A.put(W, &E.cont_dc_w);
      if (rc == OPFX_OK) rc = A.put(K, &E.cont_dc_k);
    }
  }
  {
    std::vector<int32_t> kind(d->na, OPFX_ACT_CONTINUOUS);
    if (d->act_kind) kind.assign(d->act_kind, d->act_kind + d->na);
    if (rc == OPFX_OK) rc = A.put(kind, &E.act_kind);
  }
  E.vset_src = nullptr;
  if (d->vset_slot) {
    std::vector<int32_t> src(nb, NOSRC);
    bool any = false;
    for (int i = 0; i < nb; ++i) if (d->vset_slot[i] >= 0) {
      if (d->vset_slot[i] >= d->nx) { rc = OPFX_ERR_INVALID; opfx_set_error("opfx_env_create: vset_slot out of range"); break; }
      src[i] = src_of(d->vset_slot[i]); any = true;
    }
    if (any && rc == OPFX_OK) rc = A.put(src, &E.vset_src);
  }
  if (d->n_xres > 0) {
    std::vector<int32_t> ps(d->n_xres), qs(d->n_xres), rs(d->n_xres, 0);
    for (int k = 0; k < d->n_xres; ++k) {
      if (d->xres_kind[k] == OPFX_XRES_AFFINE) {        // an entry of the solver's result bank, the bus that may be de-energised
        if (d->xres_p[k] < 0 || d->xres_p[k] >= E.nres_base || d->xres_q[k] >= nb) { rc = OPFX_ERR_INVALID; opfx_set_error("opfx_env_create: xres (affine) index out of range"); break; }
        ps[k] = d->xres_p[k]; qs[k] = d->xres_q[k] < 0 ? -1 : d->xres_q[k];
        continue;
      }
      if (d->xres_kind[k] == OPFX_XRES_MAX3) {          // three entries of the solver's result bank
        const int32_t lim = E.nres_base;
        if (!d->xres_r || d->xres_p[k] < 0 || d->xres_p[k] >= lim || d->xres_q[k] < 0 || d->xres_q[k] >= lim ||
            d->xres_r[k] < 0 || d->xres_r[k] >= lim) { rc = OPFX_ERR_INVALID; opfx_set_error("opfx_env_create: xres result index out of range"); break; }
        ps[k] = d->xres_p[k]; qs[k] = d->xres_q[k]; rs[k] = d->xres_r[k];
        continue;
      }
      if (d->xres_p[k] < 0 || d->xres_p[k] >= d->nx || d->xres_q[k] >= d->nx) { rc = OPFX_ERR_INVALID; opfx_set_error("opfx_env_create: xres column out of range"); break; }
      ps[k] = src_of(d->xres_p[k]); qs[k] = src_of(d->xres_q[k]);
    }
    if (rc == OPFX_OK) rc = A.put(rs, &E.xres_r);
    PUTN(xres_kind, d->xres_kind, d->n_xres); PUTN(xres_scale, d->xres_scale, d->n_xres);
    {
      std::vector<double> off(d->n_xres, 0.0);
      if (d->xres_offset) off.assign(d->xres_offset, d->xres_offset + d->n_xres);
      if (rc == OPFX_OK) rc = A.put(off, &E.xres_off);
    }
    if (rc == OPFX_OK) rc = A.put(ps, &E.xres_p);
    if (rc == OPFX_OK) rc = A.put(qs, &E.xres_q);
  }
  {
    // the columns the step kernel reads: everything a descriptor names (the reset kernel owns the whole row)
    int hot = 0;
    auto see = [&](int32_t slot) { if (slot >= 0 && slot < d->nx) hot = std::max(hot, slot + 1); };
    for (size_t q = 0; q < np_; ++q) see(d->pinj_slot[q]);
    for (size_t q = 0; q < nq_; ++q) see(d->qinj_slot[q]);
    for (int k = 0; k < d->na; ++k) {
      see(d->act_slot[k]); see(d->act_lo_slot[k]); see(d->act_hi_slot[k]);
      if (d->clamp_enabled) { see(d->clamp_lo_slot[k]); see(d->clamp_hi_slot[k]); }
    }
    for (size_t r = 0; r < ncost; ++r) { if (d->cost_kind[r] == OPFX_COST_UNIT) see(d->cost_pidx[r]); if (d->cost_kind[r] != OPFX_COST_EXT_GRID) see(d->cost_qidx[r]); }
    for (int k = 0; k < d->nprice; ++k) see(d->price_slot[k]);
    for (int k = 0; k < d->nobs; ++k) if (d->obs_kind[k] == OPFX_SRC_X) see(d->obs_idx[k]);
    if (d->vset_slot) for (int i = 0; i < nb; ++i) see(d->vset_slot[i]);
    for (int k = 0; k < d->n_xres; ++k) if (d->xres_kind[k] == OPFX_XRES_P || d->xres_kind[k] == OPFX_XRES_S) { see(d->xres_p[k]); see(d->xres_q[k]); }
    for (int m = 0; m < d->n_bmod; ++m) see(d->bmod_slot[m]);
    E.nx_hot = hot;
  }
  E.n_qterm = d->n_qterm;
  if (d->n_qterm > 0) {
    for (int k = 0; k < d->n_qterm; ++k)
      if (d->qterm_idx[k] < 0 || d->qterm_idx[k] >= E.nres) { rc = OPFX_ERR_INVALID; opfx_set_error("opfx_env_create: qterm_idx out of range"); break; }
    PUTN(qterm_idx, d->qterm_idx, d->n_qterm); PUTN(qterm_target, d->qterm_target, d->n_qterm);
    PUTN(qterm_weight, d->qterm_weight, d->n_qterm);
    for (int k = 0; k < d->n_qterm; ++k) if (d->qterm_idx[k] >= nb && d->qterm_idx[k] < 2 * nb) E.need_angle = 1;
  }
  if (d->n_bmod > 0) {
    size_t rows = 0;
    std::vector<int32_t> src(d->n_bmod);
    for (int m = 0; m < d->n_bmod; ++m) {
      if (d->bmod_branch[m] < -p.nb || d->bmod_branch[m] >= p.nbr || d->bmod_slot[m] < 0 || d->bmod_slot[m] >= d->nx || d->bmod_n[m] < 1) {
        rc = OPFX_ERR_INVALID; opfx_set_error("opfx_env_create: bad branch state column"); break;
      }
      for (int j = 0; j < m; ++j) if (d->bmod_branch[j] == d->bmod_branch[m]) { rc = OPFX_ERR_INVALID; opfx_set_error("opfx_env_create: two state columns for one branch / bus shunt"); }
      rows = std::max(rows, (size_t)d->bmod_ptr[m] + d->bmod_n[m]);
      src[m] = src_of(d->bmod_slot[m]);
    }
    PUTN(bmod_branch, d->bmod_branch, d->n_bmod); PUTN(bmod_lo, d->bmod_lo, d->n_bmod);
    PUTN(bmod_n, d->bmod_n, d->n_bmod); PUTN(bmod_ptr, d->bmod_ptr, d->n_bmod);
    PUTN(bmod_y, d->bmod_y, rows * 8);
    if (rc == OPFX_OK) rc = A.put(src, &E.bmod_src);
  }
#undef PUTN
  for (size_t i = 0; rc == OPFX_OK && i < ncel; ++i)
    if (d->con_src[i] < 0 || d->con_src[i] >= E.nres) { rc = OPFX_ERR_INVALID; opfx_set_error("opfx_env_create: con_src out of range"); }
  {
    // the table row is staged over [rhs | LU blocks]: make that region large enough
    const int nbe = (nb + 1) & ~1;
    E.nblk_d = std::max(E.nres, d->nx - 2 * nbe);
  }
  if (rc == OPFX_OK) rc = A.put(&e->de, 1, &e->d_de);
  if (rc == OPFX_OK) { const int zero[2] = {0, 0}; const int* q = nullptr; rc = A.put(zero, 2, &q); e->queue = const_cast<int*>(q); }
  if (rc != OPFX_OK) { delete e; return rc; }
  // what this environment fixes for the whole batch (SPEC): no PV bus in the plan; no branch-state column, no contingency and no
  // per-instance |V| set-point in the descriptor (an outage array arrives with the call: do_step)
  e->spec = (p.npv == 0 ? SPEC_NO_PV : 0) | ((d->n_bmod == 0 && d->n_cont == 0 && !E.vset_src) ? SPEC_NO_MOD : 0);
  e->lds_bytes = choose_block_storage(p, ctx->dbg, [&](int nf) { return solver_lds_bytes(p, d->na, E.nblk_d, ctx->v2, env_nacc(d->nc), E.max_mod, nf); }, &e->n_full);
  if (wants_mem(ctx->dbg, e->lds_bytes, ctx->v2)) {
    if (p.n_shared > 0) { delete e; opfx_set_error("opfx_env_create: a plan with shared slots does not run on the memory-resident kernels (build it without plan_share_slots)"); return OPFX_ERR_INVALID; }
    e->mem = true;
    e->n_full = p.n_blk;
    e->lds_bytes = solver_lds_bytes(p, d->na, E.nblk_d, true, env_nacc(d->nc), E.max_mod, e->n_full, true);
  }
  *out = e;
  return OPFX_OK;
}

extern "C" void opfx_env_destroy(opfx_env* env) { delete env; }

// the plain / DC-started step kernel of a SPEC (one translation unit per family and SPEC, opfx_kernels.h)
static StepKernel step_kernel(bool dc, int spec, int v2s, int team, int minw) {
  using Acc = const void* (*)(int, int, int);
  static const Acc plain[4] = {opfx_k_step_plain0, opfx_k_step_plain1, opfx_k_step_plain2, opfx_k_step_plain3};
  static const Acc dcs[4] = {opfx_k_step_dc0, opfx_k_step_dc1, opfx_k_step_dc2, opfx_k_step_dc3};
  return as_kernel<StepKernel>((dc ? dcs : plain)[spec & 3](v2s, team, minw));
}

// Three instances of the grid fit a CU and its blocks are stored two-value: the plain step kernel runs them as three teams of
// FOUR wavefronts compiled for three wavefronts per SIMD (k_step<2,4,...,MINW=3>) instead of three teams of two.
static bool env_three_teams_of_four(const opfx_env* env) {
  if (!env->ctx->v2 || env->mem || env->ctx->dbg.team != 0 || env->ctx->dbg.waves_per_cu != 0) return false;
  if (!(env->n_full < env->ctx->plan.n_blk)) return false;
  const size_t granule = 1024;
  return (160 * 1024) / ((env->lds_bytes + granule - 1) / granule * granule) == 3;
}

// A SMALL grid on the single-wave kernel: LDS would hold ten or more instances per CU, but at 171-190 VGPRs only two
// wavefronts fit a SIMD, i.e. eight per CU.  The kernel compiled for three wavefronts per SIMD holds twelve: 33-bus grid
// 0.250 -> 0.196 ms per 16 384 instances (65.5 -> 83.7 M step/s), 15-bus grid 0.193 -> 0.153 ms (round 5,
// scripts/probe_small_grid_occupancy.py).  Below ten instances the spills cost more than the extra wavefronts give (144-bus
// grid, eight by LDS either way: 0.224 -> 0.235 ms).
static bool env_small_grid(const opfx_env* env) {
  if (!env->ctx->v2 || env->mem || env->ctx->dbg.waves_per_cu != 0 || (env->ctx->dbg.team != 0 && env->ctx->dbg.team != 1)) return false;
  const size_t granule = 1024;
  return (160 * 1024) / ((env->lds_bytes + granule - 1) / granule * granule) >= 10;
}

static int do_step(opfx_env* env, int64_t B, const opfx_step_io* io, const opfx_solve_opts* opts,
                   int32_t mode, void* stream) {
  Opts o{opts ? opts->tol : 1e-8, opts ? opts->max_iter : 10, opts ? opts->enforce_q_lims : 1, opts ? opts->contingency_start : 0,
         opts ? opts->init : 0, opts ? opts->jacobian_reuse_tol : 0.0};
  if (!(o.reuse_tol >= 0.0)) { opfx_set_error("opfx_step: jacobian_reuse_tol must be >= 0"); return OPFX_ERR_INVALID; }
  if (o.init == OPFX_INIT_DC && !env->ctx->dp.lp_dc) { opfx_set_error("opfx_step: init = OPFX_INIT_DC needs a case with br_bdc / br_pfinj"); return OPFX_ERR_INVALID; }
  if (o.enforce_q_lims && !env->de.qg_min) o.enforce_q_lims = 0;
  int grid = 0;
  // (the chord kernels exist for two wavefronts per SIMD only: such launches keep the teams of two)
  const bool chord_launch = o.init != OPFX_INIT_DC && o.reuse_tol > 0.0 && env->ctx->plan.n_shared == 0;
  const bool plain_newton = o.init != OPFX_INIT_DC && !chord_launch;
  const bool w3 = !chord_launch && env_three_teams_of_four(env);
  const int team = env->mem ? 4 : (w3 ? 4 : pick_team(env->ctx->dbg, env->lds_bytes, env->ctx->v2, env->ctx->plan.n_shared > 0));
  DevPlan dp = env->ctx->dp;
  dp.nfull = env->n_full;
  const bool packed = env->n_full < env->ctx->plan.n_blk;
  // the specialisation of this launch (SPEC): what the environment fixes for the whole batch, less what the call brings along
  int spec = env->ctx->v2 && !env->mem ? env->spec : 0;
  if (io->outage) spec &= ~SPEC_NO_MOD;
  if (team == 1 && env->ctx->plan.nb > WAVE * POLAR_R) spec &= ~SPEC_NO_MOD;      // (the no-modifier single-wave kernels keep a polar shadow of POLAR_R bus rounds)
  if (team > 1 && env->ctx->plan.nb > WAVE * team * TEAM_PQ_R) spec &= ~SPEC_NO_PV;  // (the no-PV team kernels keep P / Q of TEAM_PQ_R bus rounds per wavefront in registers)
  spec &= OPFX_SPEC_MASK;
  const int v2s = packed ? 2 : 1;
  // (three wavefronts per SIMD: three teams of four on two-value blocks, w3; a SMALL grid on the single wave, env_small_grid)
  const int minw = (plain_newton && (w3 || (team == 1 && env_small_grid(env)))) ? 3 : OPFX_MIN_WAVES_PER_SIMD;
  StepKernel kern = !env->ctx->v2 ? as_kernel<StepKernel>(opfx_k_step_other(OPFX_K_V1, 0, 1))
                  : env->mem ? as_kernel<StepKernel>(opfx_k_step_other(OPFX_K_MEM, 1, 4))
                  : step_kernel(false, spec, v2s, team, minw);
  if (!kern) return kernel_missing("opfx_step", "k_step");
  int rc = launch_geometry(env->ctx->dbg, kern, env->lds_bytes, env->ctx->n_cu, B, &grid, &env->per_cu_spec[spec], WAVE * team);
  env->per_cu = env->per_cu_spec[spec];
  if (rc != OPFX_OK) return rc;
  if (env->mem) {
    rc = ensure_blk_mem(env->ctx, env->per_cu * env->ctx->n_cu);
    if (rc != OPFX_OK) return rc;
    dp.blk_mem = env->ctx->blk_mem; dp.blk_mem_stride = (long long)blk_mem_stride(env->ctx->plan);
    if (o.init == OPFX_INIT_DC) o.init = OPFX_INIT_FLAT;          // (no DC start in the memory-resident form)
    o.reuse_tol = 0.0;                                            // (nor chord steps)
  }
  if (o.init == OPFX_INIT_DC || !env->ctx->v2 || env->ctx->plan.n_shared > 0) o.reuse_tol = 0.0;      // (chord steps: not together with the DC start, nor on shared slots)
  if (o.reuse_tol > 0.0) {                           // the kernels compiled with chord steps (same launch geometry)
    kern = as_kernel<StepKernel>(opfx_k_step_other(OPFX_K_CHORD, v2s, team));
    if (!kern) return kernel_missing("opfx_step", "k_step, chord steps");
    int per_cu_c = env->per_cu_chord;
    rc = launch_geometry(env->ctx->dbg, kern, env->lds_bytes, env->ctx->n_cu, B, &grid, &per_cu_c, WAVE * team);
    env->per_cu_chord = per_cu_c;
    if (rc != OPFX_OK) return rc;
  }
  if (o.init == OPFX_INIT_DC && env->ctx->v2) {      // the kernels compiled with the DC start (same launch geometry)
    kern = step_kernel(true, spec, v2s, team, w3 ? 3 : OPFX_MIN_WAVES_PER_SIMD);
    if (!kern) return kernel_missing("opfx_step", "k_step, DC start");
    rc = launch_geometry(env->ctx->dbg, kern, env->lds_bytes, env->ctx->n_cu, B, &grid, &env->per_cu_dc[spec], WAVE * team);
    if (rc != OPFX_OK) return rc;
    env->per_cu = env->per_cu_dc[spec];          // (what opfx_env_get_info reports: the launch that ran last)
  }
  StepIO s{};
  s.x = io->x; s.action = io->action; s.initial_obj = io->initial_obj;
  s.step_in_episode = io->step_in_episode; s.outage = io->outage;
  s.obs = io->obs; s.reward = io->reward; s.violations = io->violations; s.penalties = io->penalties;
  s.cost = io->cost; s.objective = io->objective; s.results = io->results;
  s.mean_correction = io->mean_correction; s.max_mismatch = io->max_mismatch;
  s.terminated = io->terminated; s.truncated = io->truncated; s.valids = io->valids;
  s.converged = io->converged; s.iterations = io->iterations; s.mode = mode;
  s.total_iterations = io->total_iterations; s.min_pivot = io->min_pivot; s.min_pivot_bus = io->min_pivot_bus;
  rc = ensure_scratch(env->ctx, grid, env->de.n_cont > 0);
  if (rc != OPFX_OK) return rc;
  dp.pq = env->ctx->pq; dp.warm = env->ctx->warm;
  // (contingencies that start from a DC power flow of their own take it from the base case's by a rank-1 update where the
  //  environment holds the tables: the scratch row then keeps the base case's DC angles, see k_step)
  dp.theta0 = (env->de.cont_dc_w && o.init == OPFX_INIT_DC && o.contingency_start == 1 && env->ctx->dbg.no_rank1_dc == 0) ? env->ctx->warm : nullptr;
  dp.queue = env->queue;
  if ((s.queued = use_queue(env->ctx->dbg, B, grid, team))) HIP_TRY(hipMemsetAsync(env->queue, 0, sizeof(int), static_cast<hipStream_t>(stream)));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVE * team), env->lds_bytes, static_cast<hipStream_t>(stream),
                     dp, env->d_de, s, o, (long long)B);
  HIP_TRY(hipGetLastError());
  return OPFX_OK;
}

extern "C" int opfx_step(opfx_env* env, int64_t B, const opfx_step_io* io, const opfx_solve_opts* opts,
                         int32_t mode, void* stream) {
  opfx_step_io sio;
  opfx_solve_opts so;
  if (io) { const int rc_ = opfx_take(io, &sio, "opfx_step(opfx_step_io)"); if (rc_ != OPFX_OK) return rc_; io = &sio; }
  if (opts) { const int rc_ = opfx_take(opts, &so, "opfx_step(opfx_solve_opts)"); if (rc_ != OPFX_OK) return rc_; opts = &so; }
  if (env && io && B == 0 && mode >= 0 && mode <= 5) return OPFX_OK;       // empty batch (buffers may be null)
  if (!env || !io || !io->x || B < 0 || mode < 0 || mode > 5 || ((mode == 0 || mode == 2 || mode == 4 || mode == 5) && env->de.na > 0 && !io->action)) {
    opfx_set_error("opfx_step: bad argument");
    return OPFX_ERR_INVALID;
  }
  HIP_TRY(hipSetDevice(env->ctx->device));
  return do_step(env, B, io, opts, mode, stream);
}

extern "C" int opfx_time_steps(opfx_env* env, int64_t B, const opfx_step_io* io, const opfx_solve_opts* opts,
                               int32_t reps, void* stream, float* elapsed_ms) {
  opfx_step_io sio;
  opfx_solve_opts so;
  if (io) { const int rc_ = opfx_take(io, &sio, "opfx_time_steps(opfx_step_io)"); if (rc_ != OPFX_OK) return rc_; io = &sio; }
  if (opts) { const int rc_ = opfx_take(opts, &so, "opfx_time_steps(opfx_solve_opts)"); if (rc_ != OPFX_OK) return rc_; opts = &so; }
  if (!env || !io || !io->x || B <= 0 || reps <= 0 || !elapsed_ms) {
    opfx_set_error("opfx_time_steps: bad argument");
    return OPFX_ERR_INVALID;
  }
  HIP_TRY(hipSetDevice(env->ctx->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  // (the two events are released on every path out, HIP failures included)
  struct Events {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ~Events() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
  } ev;
  HIP_TRY(hipEventCreate(&ev.e0));
  HIP_TRY(hipEventCreate(&ev.e1));
  HIP_TRY(hipEventRecord(ev.e0, st));
  int rc = OPFX_OK;
  for (int r = 0; r < reps && rc == OPFX_OK; ++r) rc = do_step(env, B, io, opts, 0, stream);
  HIP_TRY(hipEventRecord(ev.e1, st));
  HIP_TRY(hipEventSynchronize(ev.e1));
  HIP_TRY(hipEventElapsedTime(elapsed_ms, ev.e0, ev.e1));
  return rc;
}

extern "C" int opfx_env_get_info(const opfx_env* env, int32_t* waves_per_instance, int64_t* lds_bytes_per_instance,
                                 int32_t* instances_per_cu) {
  if (!env) { opfx_set_error("opfx_env_get_info: null environment"); return OPFX_ERR_INVALID; }
  if (waves_per_instance) *waves_per_instance = (env->mem || env_three_teams_of_four(env)) ? 4 : pick_team(env->ctx->dbg, env->lds_bytes, env->ctx->v2, env->ctx->plan.n_shared > 0);
  if (lds_bytes_per_instance) *lds_bytes_per_instance = (int64_t)env->lds_bytes;
  if (instances_per_cu) *instances_per_cu = env->per_cu;
  return OPFX_OK;
}

extern "C" int opfx_env_get_spec(const opfx_env* env, int32_t* spec) {
  if (!env) { opfx_set_error("opfx_env_get_spec: null environment"); return OPFX_ERR_INVALID; }
  if (spec) {
    *spec = (env->ctx->v2 && !env->mem) ? (env->spec & OPFX_SPEC_MASK) : 0;
    if (env->ctx->plan.nb > WAVE * POLAR_R && !env_three_teams_of_four(env)
        && pick_team(env->ctx->dbg, env->lds_bytes, env->ctx->v2, env->ctx->plan.n_shared > 0) == 1) *spec &= ~SPEC_NO_MOD;      // (as do_step)
    {
      const int team = (env->mem || env_three_teams_of_four(env)) ? 4 : pick_team(env->ctx->dbg, env->lds_bytes, env->ctx->v2, env->ctx->plan.n_shared > 0);
      if (team > 1 && env->ctx->plan.nb > WAVE * team * TEAM_PQ_R) *spec &= ~SPEC_NO_PV;
    }
  }
  return OPFX_OK;
}

// The per-workgroup scratch rows of a context grow on demand (ensure_scratch): the first launch — and any launch with a larger
// grid — would otherwise call hipMalloc on the hot path, which synchronises and cannot happen inside a stream capture
// (ADVICE r05).  Sized for the largest grid a launch of B instances can take: min(B, CUs x 16 resident workgroups).
extern "C" int opfx_env_prepare(opfx_env* env, int64_t B) {
  if (!env || B < 0) { opfx_set_error("opfx_env_prepare: bad argument"); return OPFX_ERR_INVALID; }
  if (B == 0) return OPFX_OK;
  HIP_TRY(hipSetDevice(env->ctx->device));
  const int grid = (int)std::min<long long>((long long)B, (long long)env->ctx->n_cu * 16);
  return ensure_scratch(env->ctx, grid, env->de.n_cont > 0);
}

extern "C" int opfx_env_get_row_io(const opfx_env* env, int32_t* columns_read) {
  if (!env) { opfx_set_error("opfx_env_get_row_io: null environment"); return OPFX_ERR_INVALID; }
  if (columns_read) *columns_read = env->de.nx_hot;
  return OPFX_OK;
}

extern "C" int opfx_env_get_storage(const opfx_env* env, int32_t* n_blk, int32_t* n_four_value) {
  if (!env) { opfx_set_error("opfx_env_get_storage: null environment"); return OPFX_ERR_INVALID; }
  if (n_blk) *n_blk = env->ctx->plan.n_blk;
  if (n_four_value) *n_four_value = env->ctx->v2 ? env->n_full : env->ctx->plan.n_blk;
  return OPFX_OK;
}

// wavefronts (= rows) per workgroup of the reset kernel: as many as leave room for four workgroups per CU
// (opfx_debug_opts.reset_team = 1|2|4: the smaller teams on a row that would not need them — tests)
static int reset_team(const opfx_env* env) {
  const size_t row_bytes = (size_t)(((env->de.nx + 1) & ~1) + ((env->de.na + 1) & ~1)) * sizeof(double);
  int team = 4 * row_bytes <= 64 * 1024 ? 4 : (2 * row_bytes <= 64 * 1024 ? 2 : 1);
  { const int t = env->ctx->dbg.reset_team; if ((t == 1 || t == 2 || t == 4) && t < team) team = t; }
  return team;
}

extern "C" int opfx_env_set_reset(opfx_env* env, const opfx_reset_desc* d_in) {
  opfx_reset_desc desc;
  if (d_in) { const int rc_ = opfx_take(d_in, &desc, "opfx_env_set_reset(opfx_reset_desc)"); if (rc_ != OPFX_OK) return rc_; }
  const opfx_reset_desc* const d = d_in ? &desc : nullptr;
  if (!env || !d || d->n_tables < 0 || d->n_tables > MAX_TABLES) {
    opfx_set_error("opfx_env_set_reset: bad argument (at most 8 profile tables)");
    return OPFX_ERR_INVALID;
  }
  // the profile tables: an array of versioned structs whose stride is the struct_size of its elements
  std::vector<opfx_profile_desc> tables((size_t)d->n_tables);
  for (int t = 0; t < d->n_tables; ++t) {
    const uint32_t stride = d->tables ? d->tables[0].struct_size : 0;
    const auto* src = reinterpret_cast<const opfx_profile_desc*>(reinterpret_cast<const char*>(d->tables) + (size_t)t * stride);
    const int rc_ = opfx_take(d->tables ? src : nullptr, &tables[(size_t)t], "opfx_env_set_reset(opfx_profile_desc)");
    if (rc_ != OPFX_OK) return rc_;
  }
  HIP_TRY(hipSetDevice(env->ctx->device));
  DevReset& R = env->dr;
  DevArena& A = env->arena;
  R = DevReset{};
  R.n_tables = d->n_tables; R.n_uniform = d->n_uniform; R.nx = env->de.nx;
  R.init_off = d->init_off;
  R.n_normal = d->n_normal;
  R.has_mode = d->op_mode != nullptr;
  if (d->init_off >= 0 && d->init_off + R.nx > d->n_consts) { opfx_set_error("opfx_env_set_reset: init template out of range"); return OPFX_ERR_INVALID; }
  int rc = OPFX_OK;
  // ---- profile columns of all tables as one list, chunks of up to 64 columns of one table ------------------------
  std::vector<char> covered((size_t)R.nx, 0);
  {
    std::vector<int32_t> typ, slot, pch;
    std::vector<double> peak, lo, hi;
    for (int t = 0; t < d->n_tables && rc == OPFX_OK; ++t) {
      const opfx_profile_desc& T = tables[(size_t)t];
      for (int j = 0; j < T.n_cols; ++j)
        if (T.slot[j] < 0 || T.slot[j] >= R.nx || T.typ[j] < 0 || T.typ[j] >= T.n_types) {
          opfx_set_error("opfx_env_set_reset: profile slot/type out of range");
          return OPFX_ERR_INVALID;
        }
      const double* d_rel = nullptr;
      rc = A.put(T.rel, (size_t)T.n_steps * T.n_types, &d_rel);
      const unsigned long long addr = (unsigned long long)reinterpret_cast<uintptr_t>(d_rel);
      for (int j0 = 0; j0 < T.n_cols; j0 += 64) {
        const int32_t words[8] = {(int32_t)(uint32_t)(addr & 0xFFFFFFFFull), (int32_t)(uint32_t)(addr >> 32), (int32_t)typ.size() + j0,
                                  std::min(64, T.n_cols - j0), T.n_types, T.n_steps, t, 0};
        pch.insert(pch.end(), words, words + 8);
      }
      // (the chunk's first column number was computed before this table's columns were appended)
      for (int j = 0; j < T.n_cols; ++j) {
        typ.push_back(T.typ[j]); slot.push_back(T.slot[j]); covered[(size_t)T.slot[j]] = 1;
        peak.push_back(T.peak[j]); lo.push_back(T.col_min[j]); hi.push_back(T.col_max[j]);
      }
    }
    R.n_tcols = (int)typ.size();
    R.n_noise = R.n_tcols;
    R.n_pch = (int)(pch.size() / 8);
    const int32_t* d_pch = nullptr;
    if (rc == OPFX_OK) rc = A.put(pch, &d_pch);
    R.pch = reinterpret_cast<const i32x8*>(d_pch);
    if (rc == OPFX_OK) rc = A.put(typ, &R.tc_typ);
    if (rc == OPFX_OK) rc = A.put(slot, &R.tc_slot);
    if (rc == OPFX_OK) rc = A.put(peak, &R.tc_peak);
    if (rc == OPFX_OK) rc = A.put(lo, &R.tc_lo);
    if (rc == OPFX_OK) rc = A.put(hi, &R.tc_hi);
  }
  // ---- vector ops -> stages of mutually independent ops, cut into chunks of up to 64 elements ----------------------
  {
    const int n_ops = d->n_ops;
    auto reads_row = [](int code) { return code != OPFX_OP_SET_CONST && code != OPFX_OP_UNIFORM && code != OPFX_OP_NORMAL; };
    auto is_long = [](int code) { return code == OPFX_OP_NORMINV || code == OPFX_OP_TRUNCNORM || code == OPFX_OP_NORMAL; };
    auto overlap = [](int a0, int an, int b0, int bn) { return an > 0 && bn > 0 && a0 < b0 + bn && b0 < a0 + an; };
    std::vector<int> stage(n_ops, 0);
    int n_stages = 0;
    for (int k = 0; k < n_ops; ++k) {
      if (d->op_dst[k] < 0 || d->op_n[k] < 0 || d->op_dst[k] + d->op_n[k] > R.nx ||
          (reads_row(d->op_code[k]) && (d->op_a[k] < 0 || d->op_a[k] + d->op_n[k] > R.nx))) {
        opfx_set_error("opfx_env_set_reset: op range out of the row"); return OPFX_ERR_INVALID;
      }
      for (int q = 0; q < 3; ++q) {
        const int32_t off = q == 0 ? d->op_c0[k] : (q == 1 ? d->op_c1[k] : d->op_c2[k]);
        if (off >= 0 && off + d->op_n[k] > d->n_consts) { opfx_set_error("opfx_env_set_reset: op constants out of range"); return OPFX_ERR_INVALID; }
      }
      const int rk = reads_row(d->op_code[k]) ? d->op_n[k] : 0;
      for (int j = 0; j < k; ++j) {
        const int rj = reads_row(d->op_code[j]) ? d->op_n[j] : 0;
        const bool dep = overlap(d->op_dst[j], d->op_n[j], d->op_a[k], rk)            // reads what j writes
                      || overlap(d->op_dst[j], d->op_n[j], d->op_dst[k], d->op_n[k])   // writes what j writes
                      || overlap(d->op_a[j], rj, d->op_dst[k], d->op_n[k]);            // writes what j reads
        if (dep) stage[k] = std::max(stage[k], stage[j] + 1);
      }
      n_stages = std::max(n_stages, stage[k] + 1);
      for (int j = 0; j < d->op_n[k]; ++j) covered[(size_t)d->op_dst[k] + j] = 1;
    }
    if (n_stages > MAX_STAGES) { opfx_set_error("opfx_env_set_reset: more than 12 dependent stages of vector ops"); return OPFX_ERR_INVALID; }
    std::vector<int32_t> ptr{0}, och;
    const int32_t zero_off = d->n_consts;                 // 64 zeros behind the caller's constants: "no constant" (0.0)
    // Wavefront w of a team of `team` runs entries w, w + team, ... of a stage's list.  Each op's chunks are padded to a
    // whole number of rounds (empty chunks), so chunk i of EVERY op runs on wavefront i mod team; a dependency that pairs
    // chunk i with chunk i (the later op starts where the earlier one's range starts) then stays inside one wavefront and
    // its stage needs no workgroup barrier.
    const int team = env->reset_team = reset_team(env);
    env->reset_per_cu[0] = env->reset_per_cu[1] = 0;
    R.st_barrier = 1u;                                     // (stage 0 follows the profile pass: always)
    for (int k = 0; k < n_ops; ++k) {
      const int rk = reads_row(d->op_code[k]) ? d->op_n[k] : 0;
      for (int j = 0; j < k; ++j) {
        if (stage[j] == stage[k]) continue;
        const int rj = reads_row(d->op_code[j]) ? d->op_n[j] : 0;
        const bool raw = overlap(d->op_dst[j], d->op_n[j], d->op_a[k], rk), waw = overlap(d->op_dst[j], d->op_n[j], d->op_dst[k], d->op_n[k]),
                   war = overlap(d->op_a[j], rj, d->op_dst[k], d->op_n[k]);
        const bool aligned = (!raw || d->op_a[k] == d->op_dst[j]) && (!waw || d->op_dst[k] == d->op_dst[j]) && (!war || d->op_dst[k] == d->op_a[j]);
        if ((raw || waw || war) && !aligned) R.st_barrier |= 1u << stage[k];
      }
    }
    for (int sg = 0; sg < n_stages; ++sg) {
      for (int pass = 0; pass < 2; ++pass) {               // the chunks with a long function last
        for (int k = 0; k < n_ops; ++k) {
          if (stage[k] != sg || (int)is_long(d->op_code[k]) != pass) continue;
          const int mask = d->op_mode ? (d->op_mode[k] & 7) : 7;
          const int n_chunks = (d->op_n[k] + 63) / 64, n_padded = (n_chunks + team - 1) / team * team;
          for (int j0 = 0; j0 < d->op_n[k]; j0 += 64) {
            const int32_t words[8] = {d->op_code[k] | (mask << 8) | (reads_row(d->op_code[k]) ? OCH_READS_ROW : 0) | (pass ? OCH_LONG : 0),
                                      std::min(64, d->op_n[k] - j0), d->op_dst[k] + j0, d->op_a[k] + j0,
                                      d->op_c0[k] >= 0 ? d->op_c0[k] + j0 : zero_off, d->op_c1[k] >= 0 ? d->op_c1[k] + j0 : zero_off,
                                      d->op_c2[k] >= 0 ? d->op_c2[k] + j0 : zero_off, 0};
            och.insert(och.end(), words, words + 8);
          }
          for (int q = n_chunks; q < n_padded; ++q) {      // empty chunks (no element; valid addresses)
            const int32_t words[8] = {OPFX_OP_SET_CONST, 0, 0, 0, zero_off, zero_off, zero_off, 0};
            och.insert(och.end(), words, words + 8);
          }
        }
        ptr.push_back((int32_t)(och.size() / 8));
      }
    }
    R.n_stages = n_stages;
    for (size_t q = 0; q < ptr.size(); ++q) R.st_ptr[q] = ptr[q];
    if (och.empty()) och.assign(8, 0);
    const int32_t* d_och = nullptr;
    if (rc == OPFX_OK) rc = A.put(och, &d_och);
    R.och = reinterpret_cast<const i32x8*>(d_och);
    std::vector<double> cz(d->consts, d->consts + d->n_consts);
    cz.resize(cz.size() + 64, 0.0);
    if (rc == OPFX_OK) rc = A.put(cz, &R.consts);
  }
  // the row template is not needed where every column is written anyway (per-instance data sources: an op may be
  // skipped, so no)
  R.skip_template = !R.has_mode && std::all_of(covered.begin(), covered.end(), [](char c) { return c != 0; });
  // ---- observation elements ---------------------------------------------------------------------------------------
  {
    std::vector<int32_t> oe((size_t)env->de.nobs, -1);
    const int nxe = (R.nx + 1) & ~1;
    for (size_t sg = 0; sg < env->h_oseg[0].size(); ++sg)
      for (int j = 0; j < env->h_oseg[3][sg]; ++j) {
        const int kind = env->h_oseg[0][sg], idx = env->h_oseg[1][sg] + j;      // 0 row, 1 result entry, 2 action set-point
        oe[(size_t)env->h_oseg[2][sg] + j] = kind == 0 ? idx : (kind == 2 ? nxe + idx : -1);
      }
    R.n_oel = (int)oe.size();
    if (rc == OPFX_OK) rc = A.put(oe, &R.oe_src);
  }
  if (rc == OPFX_OK) rc = A.put(&env->dr, 1, &env->d_dr);
  if (rc != OPFX_OK) return rc;
  env->has_reset = true;
  return OPFX_OK;
}

extern "C" int opfx_reset(opfx_env* env, int64_t B, const opfx_reset_io* io, void* stream) {
  opfx_reset_io rio;
  if (io) { const int rc_ = opfx_take(io, &rio, "opfx_reset(opfx_reset_io)"); if (rc_ != OPFX_OK) return rc_; io = &rio; }
  if (env && env->has_reset && io && B == 0) return OPFX_OK;               // empty batch (buffers may be null)
  if (!env || !env->has_reset || !io || (!io->step_idx && !(io->step_pool && io->n_step_pool > 0)) || !io->x || B < 0) {
    opfx_set_error("opfx_reset: bad argument or opfx_env_set_reset not called");
    return OPFX_ERR_INVALID;
  }
  // (uniform / normal == NULL with draws to make: the kernel draws them itself from rng_seed, see opfx_reset_io)
  if (B == 0) return OPFX_OK;
  HIP_TRY(hipSetDevice(env->ctx->device));
  const int na = env->de.na, nx = env->dr.nx;
  const int row_doubles = ((nx + 1) & ~1) + ((na + 1) & ~1);
  const size_t row_bytes = (size_t)row_doubles * sizeof(double);
  if (row_bytes > 160 * 1024) { opfx_set_error("opfx_reset: table row does not fit the LDS"); return OPFX_ERR_TOO_LARGE; }
  ResetIO r{io->step_idx, io->noise, io->interp, io->uniform, io->normal, io->normal_noise_factor, io->x, io->mode,
            io->action, io->obs, io->keep_state, io->step_pool, io->n_step_pool, (unsigned long long)io->rng_seed, io->step_out,
            env->ctx->dp.stamps};
  const int wpb = env->reset_team;
  const size_t lds = wpb * row_bytes + 2 * wpb * sizeof(int32_t);       // rows; time steps and data sources of the rows
  const bool full = io->interp != nullptr || io->noise != nullptr || (io->mode != nullptr && env->dr.has_mode);
  auto launch = [&](auto kernel) -> int {
    if (lds > 64 * 1024)
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // as many workgroups as are resident at once (LDS / registers); each walks its share of the batch
    int& per_cu = env->reset_per_cu[full ? 1 : 0];
    if (per_cu <= 0) HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 64 * wpb, lds));
    const int grid = (int)std::min<long long>((B + wpb - 1) / wpb, (long long)env->ctx->n_cu * std::max(per_cu, 1));
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(64 * wpb), lds, static_cast<hipStream_t>(stream), env->d_dr, env->d_de, r,
                       (long long)B, row_doubles);
    return OPFX_OK;
  };
  const ResetKernel rk = as_kernel<ResetKernel>(opfx_k_reset(wpb, full ? 1 : 0));
  if (!rk) return kernel_missing("opfx_reset", "k_reset");
  const int lrc = launch(rk);
  if (lrc != OPFX_OK) return lrc;
  HIP_TRY(hipGetLastError());
  return OPFX_OK;
}

// developer probe (stamps build; declared in include/opfx_debug.h, not in the ABI header): per workgroup of the last step launches — wall clock
// (100 MHz) at its last instance, instances and Newton iterations it processed since the last read, wall clock at its
// start, HW_ID and XCC_ID registers (six doubles per workgroup); clears them
extern "C" int opfx_debug_read_finish(opfx_ctx* ctx, double* out3, int n_wg) {
  if (!ctx || !ctx->warm || (size_t)n_wg > ctx->warm_rows) return OPFX_ERR_INVALID;
  HIP_TRY(hipDeviceSynchronize());
  const size_t stride = 2 * (size_t)ctx->plan.nb;
  std::vector<double> row(3);
  for (int g = 0; g < n_wg; ++g) {
    HIP_TRY(hipMemcpy(out3 + 6 * g, ctx->warm + g * stride, 6 * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(ctx->warm + g * stride, 0, 6 * sizeof(double)));
  }
  return OPFX_OK;
}

// developer probe (include/opfx_debug.h): copies and clears the cycle sums of a context created with opfx_debug_opts.stamps
extern "C" int opfx_debug_read_stamps(opfx_ctx* ctx, unsigned long long* out32) {
  if (!ctx || !ctx->dp.stamps) return OPFX_ERR_INVALID;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out32, ctx->dp.stamps, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemset(ctx->dp.stamps, 0, 32 * sizeof(unsigned long long)));
  return OPFX_OK;
}
