// Internal host-side representation of a compiled grid plan (not part of the ABI).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "opfx.h"
#include "opfx_debug.h"

struct opfx_plan {
  opfx_debug_opts dbg{};               // developer switches this plan was built with (all zero: the defaults)
  int32_t nb = 0, nbr = 0, nref = 0, npv = 0, npq = 0;
  double base_mva = 1.0;
  // case copy
  std::vector<int32_t> bus_type;
  std::vector<double> vm_set, va_set;
  std::vector<int32_t> br_f, br_t;
  std::vector<double> br_y, br_kf, br_kt;
  std::vector<double> br_bdc, br_pfinj;  // DC model of the branches (empty: none given)
  std::vector<double> gs_copy;           // [nb] bus shunt conductance, p.u. (DC right-hand side)
  std::vector<int32_t> ref_bus;        // REF buses, increasing order
  std::vector<int32_t> ref_ord;        // [nb] ordinal among REF buses or -1
  // Ybus block CSR (all buses, diagonal included, columns sorted)
  std::vector<int32_t> y_ptr, y_col, y_blk, y_diag;
  std::vector<double> y_g, y_b;
  // per-branch positions of its four stamps in the CSR (for per-instance outages)
  std::vector<int32_t> br_pos;         // [nbr*4] ff, ft, tf, tt
  std::vector<int32_t> br_island;      // [nbr] 1 = outage of this branch leaves a bus without a path to a REF bus
  std::vector<int32_t> isl_ptr, isl_bus;   // CSR: buses that outage k cuts off (empty for most branches)
  // block LU pattern
  int32_t n_blk = 0;
  int32_t n_full = 0;                      // blocks [0, n_full) hold four values, the rest two (plan.cpp renumber_blocks)
  std::vector<int32_t> blk_row, blk_col;   // [n_blk]
  std::vector<int32_t> diag_blk;           // [nb] (-1 for REF)
  std::vector<int32_t> fill_blk;
  // shared slots (plan.cpp share_slots): n_shared fill blocks live in the id of a block that is dead by the time they are
  // born; such an id is zeroed during level zero_lev[q] (zero-at-birth, carried by a team item of that level)
  int32_t n_shared = 0;
  std::vector<int32_t> zero_lev, zero_id;
  // forward elimination schedule
  std::vector<int32_t> lev_tptr;           // [nlev+1] -> targets
  std::vector<int32_t> tgt_blk;            // block id, or -1-bus for an rhs target
  std::vector<int32_t> tgt_sptr;           // [ntgt+1] -> sources
  std::vector<int32_t> src_ik, src_kk, src_kj;   // src_kj = pivot bus for rhs targets
  // backward substitution schedule (same level partition, walked in reverse)
  std::vector<int32_t> lev_pptr;           // [nlev+1] -> pivots
  std::vector<int32_t> piv_bus, piv_uptr, u_blk, u_col;
  // ---- lane programmes for the register-resident kernel (k_*2) ------------------
  // Every array is [round][lane] (64 lanes per round) so that a wave loads its
  // descriptors with one coalesced access per round, once per kernel.
#ifndef OPFX_ELL_WIDTH
#define OPFX_ELL_WIDTH 2
#endif
  // ELL width: off-diagonal Ybus entries per bus row held in the row's own lane; longer rows put the rest into the
  // flat overflow list.  Padding slots cost a full entry evaluation each, and power grids are sparse (144-bus MV
  // grid: 26/102/14/2 buses of degree 1/2/3/7; 306-bus HV: 131 of degree 1): width 4 evaluates 13 slot-rounds per
  // pass of the 144-bus grid for 286 entries, width 2 seven (6 + one overflow round).
  static constexpr int KA = OPFX_ELL_WIDTH;
  static constexpr int APK_VECS = KA + 2;  // 16-byte vectors per (round, lane) of lp_apk
  int32_t ra = 0, rh = 0, rb = 0, rc = 0;  // rounds: bus rows, heavy-row overflow, factor/forward, backward
  int32_t rb_pad = 0, rc_pad = 0;          // rounds of the two parts of lp_bc (multiples of 4)
  std::vector<uint32_t> lp_a_ent;          // [ra][KA][64]  j | blk<<16   (0xFFFF = none)
  std::vector<double> lp_a_y;              // [ra][KA][64][2] g,b
  std::vector<double> lp_a_ydiag;          // [ra][64][2]
  std::vector<uint32_t> lp_a_dblk;         // [ra][64]  diag blk | heavy flag<<16
  std::vector<uint32_t> lp_h_ent;          // [rh][64]  j | blk<<16 ; row bus in lp_h_row
  std::vector<uint32_t> lp_h_row;          // [rh]      bus whose overflow entries this round holds
  std::vector<double> lp_h_y;              // [rh][64][2]
  std::vector<uint32_t> lp_b;              // [rb][64][2]  tb|ik<<16 , kk|kj<<16
  std::vector<uint32_t> lp_b2;             // [rb][64]     i|k<<16: the item also applies its multiplier to y_k -> y_i (0xFFFF|0xFFFF<<16: no)
  std::vector<uint32_t> lp_b3;             // [rb][64]     a SECOND column of the same multiplier: target2 | A_kj2 << 16 (0xFFFF both: none)
  std::vector<uint32_t> lp_c;              // [rc][64][2]  (0x8000|k)|blk(k,j)<<16 , blk(j,j)|j<<16  (back substitution, same item form as lp_b)
  // device-facing packed forms (16-byte vectors, one coalesced KB per wave-load):
  std::vector<uint32_t> lp_bc;             // [rb_pad+rc_pad][64][4]  w0, w1, second column, round flags | rider bits (pad: empty items)
  // chord steps (opfx_solve_opts.jacobian_reuse_tol): forward substitution alone + the same back substitution
  int32_t rf = 0, rf_pad = 0;              // rounds of the forward substitution (right-hand-side items of every level), padded to 4
  std::vector<uint32_t> lp_bcc;            // [rf_pad+rc_pad][64][4]
  std::vector<uint32_t> lp_apk;            // [ra][KA+2][64][4]  ent0..ent(KA-1), dblk | y0 | .. | y(KA-1) | ydiag
  std::vector<uint32_t> lp_hpk;            // [rh][2][64][4]  y(g,b) | j|blk<<16, row bus (0xFFFF none), 0, 0
  std::vector<int32_t> lp_hrows;           // buses whose rows have overflow entries (their sums start at 0)
  // DC start (opfx_solve_opts.init = OPFX_INIT_DC): B' on the Ybus pattern, laid out like lp_a_y / lp_h_y
  std::vector<double> lp_dc;               // [ra][KA+2][64]  B'_ij (ELL slots), B'_ii, constant part of the right-hand side
  std::vector<double> lp_hdc;              // [rh][64]        B'_ij of the overflow entries
  // cooperative kernels (2 or 4 wavefronts per instance): the rounds of each group dealt round-robin
  // to the waves and laid out per wave, [round][wave][64][4]; word 3 of an item carries flags:
  // bit 0 = workgroup barrier after this round (end of a group that another wavefront continues).
  std::vector<uint32_t> lp_team[2];        // [0]: 2 waves, [1]: 4 waves
  int32_t team_rounds[2] = {0, 0};         // rounds per wave (multiples of 4)
  int32_t team_barriers[2] = {0, 0};       // rounds that end with a workgroup barrier
  int32_t team_kb[2] = {0, 0};             // rounds of the first part of the stream (the tail chain runs after it); = team_rounds without a tail
  std::vector<uint32_t> lp_teamc[2];       // chord stream of the teams: [forward-substitution groups | (tail chain) | back-substitution groups]
  int32_t team_rounds_c[2] = {0, 0}, team_barriers_c[2] = {0, 0}, team_kb_c[2] = {0, 0};
  int32_t n_groups = 0;                    // independent groups of the B/C stream
  // dense tail of the elimination (levels with one pivot each at the end), solved in registers by the wave teams
  static constexpr int TAIL_MAX = 32;
  int32_t tail_m = 0;                      // pivots in the tail (0: none / not worth it)
  std::vector<uint32_t> tail_bus;          // [TAIL_MAX] bus | diagonal block << 16 of tail pivot e (elimination order, e = tail_m-1 last)
  std::vector<int32_t> tail_ids32;         // (the same, widened: read-back for tests)
  std::vector<uint16_t> tail_ids;          // [tail_m][M] id of U-block (row e, column s) at [e * M + s], 0xFFFF = none; M = tail_m rounded up to a multiple of 8
  std::vector<int32_t> lp_groups;          // round offsets into lp_bc: rounds of one group are mutually
                                           // independent (an elimination level / the back-substitution terms of a level)
  int32_t nnz_j = 0;
  int32_t max_level_width = 0;
  int32_t n_levels() const { return (int32_t)lev_tptr.size() - 1; }
};

void opfx_set_error(const std::string& msg);

// Versioned structs (include/opfx.h, VERSIONING): copy the caller's struct into a zeroed one of THIS library's layout.
// Accepted sizes: [min_size, sizeof(T)] — members are only ever appended, a shorter struct of the same series leaves the
// new ones zero (their defaults); anything else is a caller built against another header and is refused.
// `min_size` is the struct's size in the FIRST layout of the current major.minor series.  The default, sizeof(T), is right
// for as long as nothing has been appended since the last MINOR bump (true for 0.3); whoever appends a member without
// bumping passes the previous sizeof here — and whoever changes a function signature or reorders members bumps MINOR
// (include/opfx.h) together with ABI_VERSION in opfgym_amd/capi.py.
#include <cstring>
template <class T>
inline int opfx_take(const T* in, T* out, const char* what, size_t min_size = sizeof(T)) {
  std::memset(static_cast<void*>(out), 0, sizeof(T));
  if (!in) { opfx_set_error(std::string(what) + ": null struct"); return OPFX_ERR_INVALID; }
  const uint32_t sz = in->struct_size;
  if (sz < min_size || sz > sizeof(T)) {
    opfx_set_error(std::string(what) + ": struct_size " + std::to_string(sz) + " is not a layout of libopfx " +
                   std::to_string(OPFX_VERSION_MAJOR) + "." + std::to_string(OPFX_VERSION_MINOR) + " (expected " +
                   (min_size == sizeof(T) ? std::to_string(sizeof(T)) : std::to_string(min_size) + " .. " + std::to_string(sizeof(T))) +
                   " bytes: set struct_size = sizeof(struct), OPFX_INIT, and build against this library's include/opfx.h)");
    return OPFX_ERR_INVALID;
  }
  std::memcpy(static_cast<void*>(out), in, sz);
  out->struct_size = (uint32_t)sizeof(T);
  return OPFX_OK;
}
