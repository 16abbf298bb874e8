// Internal host-side representation of a compiled grid plan (not part of the ABI).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "opfx.h"

struct opfx_plan {
  int32_t nb = 0, nbr = 0, nref = 0, npv = 0, npq = 0;
  double base_mva = 1.0;
  // case copy
  std::vector<int32_t> bus_type;
  std::vector<double> vm_set, va_set;
  std::vector<int32_t> br_f, br_t;
  std::vector<double> br_y, br_kf, br_kt;
  std::vector<int32_t> ref_bus;        // REF buses, increasing order
  std::vector<int32_t> ref_ord;        // [nb] ordinal among REF buses or -1
  // Ybus block CSR (all buses, diagonal included, columns sorted)
  std::vector<int32_t> y_ptr, y_col, y_blk, y_diag;
  std::vector<double> y_g, y_b;
  // per-branch positions of its four stamps in the CSR (for per-instance outages)
  std::vector<int32_t> br_pos;         // [nbr*4] ff, ft, tf, tt
  // block LU pattern
  int32_t n_blk = 0;
  std::vector<int32_t> blk_row, blk_col;   // [n_blk]
  std::vector<int32_t> diag_blk;           // [nb] (-1 for REF)
  std::vector<int32_t> fill_blk;
  // forward elimination schedule
  std::vector<int32_t> lev_tptr;           // [nlev+1] -> targets
  std::vector<int32_t> tgt_blk;            // block id, or -1-bus for an rhs target
  std::vector<int32_t> tgt_sptr;           // [ntgt+1] -> sources
  std::vector<int32_t> src_ik, src_kk, src_kj;   // src_kj = pivot bus for rhs targets
  // backward substitution schedule (same level partition, walked in reverse)
  std::vector<int32_t> lev_pptr;           // [nlev+1] -> pivots
  std::vector<int32_t> piv_bus, piv_uptr, u_blk, u_col;
  int32_t nnz_j = 0;
  int32_t max_level_width = 0;
  int32_t n_levels() const { return (int32_t)lev_tptr.size() - 1; }
};

void opfx_set_error(const std::string& msg);
