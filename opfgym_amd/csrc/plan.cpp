// Host-side symbolic analysis: Ybus block CSR, level-scheduled multiple-minimum-
// degree ordering on the bus graph, symbolic 2x2-block LU, elimination schedule.
//
// Replaces the per-call structure work inside pandapower.runpp (third party;
// call site /root/reference/opfgym/opf_env.py:703): `_pd2ppc` bus typing,
// `makeYbus`, and the symbolic half of the sparse factorisation that
// `newtonpf` redoes in every Newton iteration.  In the benchmark environments
// only injections change between instances and steps (voltage_control.py:56-57,
// eco_dispatch.py:54-55), so all of it is compiled once per grid here.
#include "plan.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <unordered_map>

static thread_local std::string g_last_error;
void opfx_set_error(const std::string& msg) { g_last_error = msg; }
extern "C" const char* opfx_last_error(void) { return g_last_error.c_str(); }
extern "C" void opfx_version(int* major, int* minor, int* patch) {
  if (major) *major = OPFX_VERSION_MAJOR;
  if (minor) *minor = OPFX_VERSION_MINOR;
  if (patch) *patch = OPFX_VERSION_PATCH;
}

namespace {

struct BlockMap {
  int32_t nb;
  std::unordered_map<int64_t, int32_t> ids;
  std::vector<int32_t>* rows;
  std::vector<int32_t>* cols;
  int32_t get(int32_t i, int32_t j) const {
    auto it = ids.find((int64_t)i * nb + j);
    return it == ids.end() ? -1 : it->second;
  }
  int32_t get_or_add(int32_t i, int32_t j, bool* added) {
    int64_t key = (int64_t)i * nb + j;
    auto it = ids.find(key);
    if (it != ids.end()) { if (added) *added = false; return it->second; }
    int32_t id = (int32_t)rows->size();
    ids.emplace(key, id);
    rows->push_back(i);
    cols->push_back(j);
    if (added) *added = true;
    return id;
  }
};

// Lane programmes: the same schedule, re-expressed as fixed-size per-lane work
// descriptors grouped in rounds of 64 so that the kernel keeps them in registers.
//  A  bus rows in ELL form (KA off-diagonal entries per row; longer rows spill
//     their remaining entries into "heavy" rounds reduced across the wave);
//  B  per elimination level: one item per update term
//     A_ij -= A_ik Kinv A_kj / y_i -= A_ik Kinv y_k (accumulated with LDS atomics);
//  C  per level in reverse: extra U-term items (accumulated with LDS atomics), then
//     solve items with up to two U-terms inline.
// Block ids, in this order:
//   [0, n_full - n_fill)      blocks that need all four values: diagonal blocks, off-diagonal
//                             Jacobian blocks that an update targets or whose row bus may be PV
//   [n_full - n_fill, n_full) fill blocks (zeroed as one contiguous range, no index list)
//   [n_full, n_blk)           "plain" off-diagonal Jacobian blocks of PQ rows that no update ever
//                             targets: they keep the shape [[a, b], [-b, a]] of dS/d(theta, ln|V|),
//                             so the kernels store only (a, b) for them (a quarter of the LU
//                             values of a radial grid: 26.2 -> 21.7 KB of LDS per 144-bus instance).
// The dense tail of the elimination: the run of final levels that hold ONE pivot each (see build_lane_programs).
int tail_length(const opfx_plan* p) {
  const int nlev = p->n_levels();
  int m = 0;
  for (int lev = nlev - 1; lev >= 0 && p->lev_pptr[lev + 1] - p->lev_pptr[lev] == 1 && m < opfx_plan::TAIL_MAX; --lev) ++m;
  return (m < 4 || p->dbg.plan_no_tail) ? 0 : m;
}

void renumber_blocks(opfx_plan* p) {
  const int32_t n = p->n_blk;
  std::vector<char> is_fill(n, 0), needs_full(n, 0);
  for (int32_t f : p->fill_blk) is_fill[f] = 1;
  for (int32_t b = 0; b < n; ++b)
    if (p->blk_row[b] == p->blk_col[b] || p->bus_type[p->blk_row[b]] != OPFX_PQ) needs_full[b] = 1;
  for (int32_t tb : p->tgt_blk) if (tb >= 0) needs_full[tb] = 1;
  {   // blocks inside the dense tail: the register chain reads them without the two-value case distinction
    const int m = tail_length(p), nlev = p->n_levels();
    std::vector<char> in_tail(p->nb, 0);
    for (int e = 0; e < m; ++e) in_tail[p->piv_bus[p->lev_pptr[nlev - m + e]]] = 1;
    for (int32_t b = 0; b < n; ++b) if (in_tail[p->blk_row[b]] && in_tail[p->blk_col[b]]) needs_full[b] = 1;
  }
  if (p->dbg.plan_no_pack) std::fill(needs_full.begin(), needs_full.end(), 1);   // developer probe
  std::vector<int32_t> perm(n);
  int32_t next = 0;
  for (int32_t b = 0; b < n; ++b) if (!is_fill[b] && needs_full[b]) perm[b] = next++;
  for (int32_t b = 0; b < n; ++b) if (is_fill[b]) perm[b] = next++;
  p->n_full = next;
  for (int32_t b = 0; b < n; ++b) if (!is_fill[b] && !needs_full[b]) perm[b] = next++;
  auto map = [&](std::vector<int32_t>& v) { for (auto& x : v) if (x >= 0) x = perm[x]; };
  map(p->y_blk); map(p->diag_blk); map(p->fill_blk); map(p->tgt_blk);
  map(p->src_ik); map(p->src_kk); map(p->u_blk);
  const int32_t ntgt = (int32_t)p->tgt_blk.size();
  for (int32_t t = 0; t < ntgt; ++t)
    if (p->tgt_blk[t] >= 0)                      // src_kj of an rhs target is a bus, not a block
      for (int32_t s = p->tgt_sptr[t]; s < p->tgt_sptr[t + 1]; ++s) p->src_kj[s] = perm[p->src_kj[s]];
  std::vector<int32_t> row(n), col(n);
  for (int32_t b = 0; b < n; ++b) { row[perm[b]] = p->blk_row[b]; col[perm[b]] = p->blk_col[b]; }
  p->blk_row.swap(row); p->blk_col.swap(col);
  std::sort(p->fill_blk.begin(), p->fill_blk.end());
}

// Shared slots (round 5).  What pins the wave teams of a meshed grid at two instances per CU is LDS, and a LOWER block (i, j)
// — column j eliminated before row i — is read for the last time at level(j), as the multiplier A_ik of that level, while fill
// blocks are born throughout the elimination (a fill block is first touched at the level that creates it).  So a fill block
// born at level b may live in the id (= LDS slot) of a four-value lower block that died at a level d <= b - 2; what it needs
// is a ZERO in the slot at its birth (its first updates are atomic adds): the plan records (level z, id) pairs with
// d < z < b, and a team item of level z carries the store (build_lane_programs: the rider bits of a team item are free).
// Greedy over the fill blocks in birth order (optimal for intervals up to the one-level gap); a level takes as many zero
// stores as it has items.  Ids afterwards, as before:  [four-value ids whose first tenant is an original block |
// ids whose first tenant is a fill block (zeroed as one range in phase A) | two-value ids].
// Such a plan is for the wave-team kernels with full Newton: the single-wave stream has no room for the stores (its items
// carry riders) and chord iterations re-read the lower blocks.
// Whether it pays — whether a CU then holds a third instance — depends on the environment's own LDS needs as well: the caller
// decides (opfgym_amd.BatchedOpfEnv builds both plans where the grid is in the two-instances-per-CU regime and keeps the better).
void share_slots(opfx_plan* p) {
  const int32_t n = p->n_blk, nlev = p->n_levels(), nfull = p->n_full;
  if (n == 0 || nlev < 3) return;
  std::vector<int32_t> lev_of(p->nb, -1);
  for (int lev = 0; lev < nlev; ++lev)
    for (int q = p->lev_pptr[lev]; q < p->lev_pptr[lev + 1]; ++q) lev_of[p->piv_bus[q]] = lev;
  std::vector<char> is_fill(n, 0);
  for (int32_t f : p->fill_blk) is_fill[f] = 1;
  std::vector<int32_t> birth(n, -1);                       // fill blocks: the first level that targets them
  std::vector<int32_t> cap(nlev, 0);                       // items of a level (with second columns: the smaller count)
  for (int lev = 0; lev < nlev; ++lev) {
    std::unordered_map<int32_t, int32_t> per_ik;
    for (int t = p->lev_tptr[lev]; t < p->lev_tptr[lev + 1]; ++t) {
      const int32_t tb = p->tgt_blk[t];
      const int32_t ns = p->tgt_sptr[t + 1] - p->tgt_sptr[t];
      if (tb < 0) { cap[lev] += ns; continue; }
      if (is_fill[tb] && birth[tb] < 0) birth[tb] = lev;
      for (int s_ = p->tgt_sptr[t]; s_ < p->tgt_sptr[t + 1]; ++s_) ++per_ik[p->src_ik[s_]];
    }
    for (auto& kv : per_ik) cap[lev] += (kv.second + 1) / 2;
  }
  auto death = [&](int32_t b) {                            // level of the last read of a lower block, else -1 (lives on)
    const int32_t i = p->blk_row[b], j = p->blk_col[b];
    return (i != j && lev_of[i] >= 0 && lev_of[j] >= 0 && lev_of[j] < lev_of[i]) ? lev_of[j] : -1;
  };
  // releases: (first level at which the slot may be zeroed, slot)
  std::vector<std::vector<int32_t>> release_at(nlev + 2);
  for (int32_t b = 0; b < nfull; ++b) if (!is_fill[b]) { const int d = death(b); if (d >= 0) release_at[d + 1].push_back(b); }
  std::vector<int32_t> fills;
  for (int32_t b = 0; b < n; ++b) if (is_fill[b] && birth[b] >= 0) fills.push_back(b);
  std::stable_sort(fills.begin(), fills.end(), [&](int32_t a, int32_t b) { return birth[a] < birth[b]; });
  std::vector<int32_t> host(n, -1);                        // fill block -> the id it moves into
  std::vector<std::pair<int32_t, int32_t>> zero_ops;       // (level, slot in OLD ids)
  std::vector<std::pair<int32_t, int32_t>> free_slots;     // (zero-able from level, slot)
  std::vector<int32_t> cap_left = cap;
  size_t fi = 0;
  for (int lev = 0; lev < nlev; ++lev) {
    for (int32_t sl : release_at[lev]) free_slots.push_back({lev, sl});
    for (; fi < fills.size() && birth[fills[fi]] == lev; ++fi) {
      const int32_t f = fills[fi];
      int32_t slot = f;
      // a free slot whose window [from, lev - 1] holds a level with a store to spare (the latest such level)
      for (size_t q = 0; q < free_slots.size(); ++q) {
        int z = -1;
        for (int l = lev - 1; l >= free_slots[q].first; --l) if (cap_left[l] > 0) { z = l; break; }
        if (z < 0) continue;
        slot = free_slots[q].second;
        host[f] = slot;
        --cap_left[z];
        zero_ops.push_back({z, slot});
        free_slots.erase(free_slots.begin() + (long)q);
        break;
      }
      const int d = death(f);
      if (d >= 0) release_at[std::min(d + 1, nlev + 1)].push_back(slot);
    }
  }
  const int32_t n_hosted = (int32_t)zero_ops.size();
  if (n_hosted == 0) return;
  // ---- new ids --------------------------------------------------------------------------------------------------
  std::vector<int32_t> perm(n, -1);
  int32_t next = 0;
  for (int32_t b = 0; b < nfull; ++b) if (!is_fill[b]) perm[b] = next++;                   // originals, four-value
  const int32_t fill_lo = next;
  for (int32_t b = 0; b < nfull; ++b) if (is_fill[b] && host[b] < 0) perm[b] = next++;     // fill blocks in a slot of their own
  const int32_t n_full_new = next;
  for (int32_t b = nfull; b < n; ++b) perm[b] = next++;                                    // two-value
  const int32_t n_new = next;
  for (int32_t f : fills) if (host[f] >= 0) {               // (hosts are resolved in birth order: a host's own id is final)
    int32_t h = host[f];
    while (host[h] >= 0) h = host[h];
    perm[f] = perm[h];
  }
  auto map = [&](std::vector<int32_t>& v) { for (auto& x : v) if (x >= 0) x = perm[x]; };
  map(p->y_blk); map(p->diag_blk); map(p->tgt_blk); map(p->src_ik); map(p->src_kk); map(p->u_blk);
  const int32_t ntgt = (int32_t)p->tgt_blk.size();
  for (int32_t t = 0; t < ntgt; ++t)
    if (p->tgt_blk[t] >= 0)
      for (int32_t s_ = p->tgt_sptr[t]; s_ < p->tgt_sptr[t + 1]; ++s_) p->src_kj[s_] = perm[p->src_kj[s_]];
  std::vector<int32_t> row(n_new, -1), col(n_new, -1);
  for (int32_t b = 0; b < n; ++b) if (host[b] < 0) { row[perm[b]] = p->blk_row[b]; col[perm[b]] = p->blk_col[b]; }   // (the FIRST tenant)
  p->blk_row.swap(row); p->blk_col.swap(col);
  p->fill_blk.clear();
  for (int32_t id = fill_lo; id < n_full_new; ++id) p->fill_blk.push_back(id);
  for (auto& zo : zero_ops) { p->zero_lev.push_back(zo.first); p->zero_id.push_back(perm[zo.second]); }
  p->n_blk = n_new;
  p->n_full = n_full_new;
  p->n_shared = n_hosted;
}

// LDS bank conflicts of the B/C items.  A wave's 64-bit LDS access is served in two groups of 32 lanes over 32
// eight-byte banks, and every operand array of the kernels is indexed by an id, so within one access the bank of a lane
// is (array base + id) mod 32: two lanes of a group collide when their ids differ but agree mod 32 (equal ids are a
// broadcast).  The items of one group of the stream (an elimination level, the back-substitution terms of a level) are
// mutually independent, so which round and which half-wave runs which item is free: every item goes, in stream order,
// to the half-round where it adds the fewest extra LDS cycles over its accesses (A_ik, A_kk, A_kj or y_k, the target's
// atomics, a rider's y_k / y_i; a group of 32 lanes takes as many cycles as its busiest bank holds addresses).
// `base`: n_rounds x 64 items x 4 words, in place; word 3 (flags) stays with its round.  opfx_debug_opts.plan_no_bank keeps the
// plan's own order.
// Word 3 of an item: bits 0-1 flags of its ROUND, bits 2-16 / 17-31 the rider's buses i / k (0x7FFF: none).
constexpr uint32_t RIDER_NONE15 = 0x7FFFu, NO_RIDER_BITS = (RIDER_NONE15 << 2) | (RIDER_NONE15 << 17);
inline uint32_t rider_bits(uint32_t rider_ik) {        // from the host form i | k << 16 (0xFFFF both: none)
  const uint32_t i = rider_ik & 0xFFFFu, k = rider_ik >> 16;
  return (i == 0xFFFFu || k == 0xFFFFu) ? NO_RIDER_BITS : ((i << 2) | (k << 17));
}

void spread_over_banks(uint32_t* base, int n_rounds, int n_full) {
  constexpr uint32_t NONE = 0xFFFFu;
  constexpr int NC = 8;
  struct It { uint32_t w[3]; uint32_t rider; int id[NC]; int wt[NC]; };
  std::vector<It> live;
  for (int l = 0; l < 64 * n_rounds; ++l) {
    const uint32_t* w = base + 4 * l;
    if ((w[0] & 0xFFFF) == NONE) continue;
    It it{};
    for (int q = 0; q < 3; ++q) it.w[q] = w[q];
    it.rider = w[3] & ~3u;
    const uint32_t tb = w[0] & 0xFFFF, ik = w[0] >> 16, kk = w[1] & 0xFFFF, kj = w[1] >> 16;
    const bool rhs_t = (tb & 0x8000u) != 0;
    // access classes: 0 blocks read through ik, 1 through kk, 2 through kj (block) / 3 (right-hand side), 4 target
    // atomics on blocks / 5 on the right-hand side, 6 / 7 the second column's block and target (weights: LDS
    // instructions of that access; atomics count double)
    for (int c = 0; c < NC; ++c) { it.id[c] = -1; it.wt[c] = 0; }
    it.id[0] = (int)ik; it.wt[0] = (int)ik < n_full ? 4 : 2;
    it.id[1] = (int)kk; it.wt[1] = 4;
    if (rhs_t) { it.id[3] = (int)kj; it.wt[3] = 2; it.id[5] = (int)(tb & 0x7FFF); it.wt[5] = 4; }
    else { it.id[2] = (int)kj; it.wt[2] = (int)kj < n_full ? 4 : 2; it.id[4] = (int)tb; it.wt[4] = 8; }
    if ((w[2] & 0xFFFF) != NONE) { const int kj2 = (int)(w[2] >> 16); it.id[6] = kj2; it.wt[6] = kj2 < n_full ? 4 : 2; it.id[7] = (int)(w[2] & 0xFFFF); it.wt[7] = 8; }
    if (it.rider != NO_RIDER_BITS) { it.id[3] = (int)(it.rider >> 17); it.wt[3] += 2; if (it.id[5] < 0) { it.id[5] = (int)((it.rider >> 2) & 0x7FFFu); it.wt[5] = 4; } }
    live.push_back(it);
  }
  const int n = (int)live.size(), nbin = 2 * n_rounds;
  if (n < 2) return;
  // rounds actually needed stay as they are: the items fill the first ceil(n / 64) rounds (a round that is empty in
  // the plan's order stays empty: padding)
  const int used_rounds = (n + 63) / 64, used_bins = 2 * used_rounds;
  struct Banks { std::array<std::vector<int>, 32> ids; int worst = 0; };
  std::vector<std::array<Banks, NC>> tab(used_bins);
  auto is_read = [](int c) { return c < 4 || c == 6; };
  std::vector<int> cnt(used_bins, 0), bin_of(n, 0);
  auto price = [&](int h, const It& it) {
    int c_ = 0;
    for (int c = 0; c < NC; ++c) {
      if (it.id[c] < 0) continue;
      const auto& v = tab[h][c].ids[it.id[c] & 31];
      // (reads of one address are a broadcast; atomic adds on one address are serialised like any other conflict)
      if (is_read(c) && std::find(v.begin(), v.end(), it.id[c]) != v.end()) continue;
      if ((int)v.size() + 1 > std::max(tab[h][c].worst, 1)) c_ += it.wt[c];
    }
    return c_;
  };
  auto put = [&](int h, const It& it) {
    for (int c = 0; c < NC; ++c) {
      if (it.id[c] < 0) continue;
      auto& v = tab[h][c].ids[it.id[c] & 31];
      if (!is_read(c) || std::find(v.begin(), v.end(), it.id[c]) == v.end()) { v.push_back(it.id[c]); tab[h][c].worst = std::max(tab[h][c].worst, (int)v.size()); }
    }
  };
  auto rebuild = [&](int h, int skip) {          // the tables of half-round h without item `skip`
    for (int c = 0; c < NC; ++c) { for (auto& v : tab[h][c].ids) v.clear(); tab[h][c].worst = 0; }
    for (int q = 0; q < n; ++q) if (bin_of[q] == h && q != skip) put(h, live[q]);
  };
  auto choose = [&](const It& it) {
    int best = -1, best_cost = 0;
    for (int h = 0; h < used_bins; ++h) {
      if (cnt[h] >= 32) continue;
      const int c = 64 * price(h, it) + cnt[h];              // (ties: the emptier half-round)
      if (best < 0 || c < best_cost) { best = h; best_cost = c; }
    }
    return best;
  };
  std::fill(bin_of.begin(), bin_of.end(), -1);
  for (int q = 0; q < n; ++q) { const int h = choose(live[q]); bin_of[q] = h; ++cnt[h]; put(h, live[q]); }
  // local search: take an item that sits on a busiest bank of its half-round out and place it again
  for (int pass = 0; pass < 1; ++pass) {
    bool moved = false;
    for (int q = 0; q < n; ++q) {
      const int h0 = bin_of[q];
      bool hot = false;
      for (int c = 0; c < NC && !hot; ++c)
        if (live[q].id[c] >= 0 && tab[h0][c].worst > 1 && (int)tab[h0][c].ids[live[q].id[c] & 31].size() == tab[h0][c].worst) hot = true;
      if (!hot) continue;
      rebuild(h0, q); --cnt[h0]; bin_of[q] = -1;
      const int before = 64 * price(h0, live[q]) + cnt[h0];
      int h1 = choose(live[q]);
      if (64 * price(h1, live[q]) + cnt[h1] >= before) h1 = h0;
      bin_of[q] = h1; ++cnt[h1]; put(h1, live[q]);
      moved = moved || h1 != h0;
    }
    if (!moved) break;
  }
  (void)nbin;
  for (int l = 0; l < 64 * n_rounds; ++l) { uint32_t* w = base + 4 * l; w[0] = NONE | (NONE << 16); w[1] = NONE | (NONE << 16); w[2] = NONE | (NONE << 16); w[3] = (w[3] & 3u) | NO_RIDER_BITS; }
  std::vector<int> at(used_bins);
  for (int h = 0; h < used_bins; ++h) at[h] = (h / 2) * 64 + (h & 1) * 32;
  for (int q = 0; q < n; ++q) {
    uint32_t* w = base + 4 * (at[bin_of[q]]++);
    for (int k = 0; k < 3; ++k) w[k] = live[q].w[k];
    w[3] = (w[3] & 3u) | live[q].rider;
  }
}

void build_lane_programs(opfx_plan* p) {
  constexpr int KA = opfx_plan::KA;
  constexpr uint32_t NONE = 0xFFFFu;
  const int nb = p->nb;
  if (nb > 0x7FFF || p->n_blk > 0x7FFF) { p->ra = p->rb = p->rc = -1; return; }   // 16-bit descriptors
  // ---- A ---------------------------------------------------------------------
  p->ra = (nb + 63) / 64;
  p->lp_a_ent.assign((size_t)p->ra * KA * 64, NONE | (NONE << 16));
  for (int r = 0; r < p->ra; ++r)                   // padding slots read the row's own bus (Y = 0): no select in the kernel
    for (int k = 0; k < KA; ++k)
      for (int l = 0; l < 64; ++l) {
        const int i = r * 64 + l;
        p->lp_a_ent[((size_t)r * KA + k) * 64 + l] = (uint32_t)(i < nb ? i : 0) | (NONE << 16);
      }
  p->lp_a_y.assign((size_t)p->ra * KA * 64 * 2, 0.0);
  p->lp_a_ydiag.assign((size_t)p->ra * 64 * 2, 0.0);
  p->lp_a_dblk.assign((size_t)p->ra * 64, NONE);
  std::vector<std::array<double, 2>> hy;
  for (int i = 0; i < nb; ++i) {
    const int r = i / 64, lane = i % 64;
    int k = 0;
    std::vector<int> overflow;
    for (int e = p->y_ptr[i]; e < p->y_ptr[i + 1]; ++e) {
      const int j = p->y_col[e];
      if (j == i) {
        p->lp_a_ydiag[((size_t)r * 64 + lane) * 2] = p->y_g[e];
        p->lp_a_ydiag[((size_t)r * 64 + lane) * 2 + 1] = p->y_b[e];
        continue;
      }
      if (k < KA) {
        const size_t o = ((size_t)r * KA + k) * 64 + lane;
        const uint32_t blk = p->y_blk[e] >= 0 ? (uint32_t)p->y_blk[e] : NONE;
        p->lp_a_ent[o] = (uint32_t)j | (blk << 16);
        p->lp_a_y[o * 2] = p->y_g[e];
        p->lp_a_y[o * 2 + 1] = p->y_b[e];
        ++k;
      } else {
        overflow.push_back(e);
      }
    }
    uint32_t d = p->diag_blk[i] >= 0 ? (uint32_t)p->diag_blk[i] : NONE;
    if (!overflow.empty()) d |= 1u << 16;
    p->lp_a_dblk[(size_t)r * 64 + lane] = d;
    if (!overflow.empty()) p->lp_hrows.push_back(i);
    for (int e : overflow) {                     // flat list of overflow entries, any row per lane
      const uint32_t blk = p->y_blk[e] >= 0 ? (uint32_t)p->y_blk[e] : NONE;
      p->lp_h_ent.push_back((uint32_t)p->y_col[e] | (blk << 16));
      p->lp_h_row.push_back((uint32_t)i);
      p->lp_h_y.push_back(p->y_g[e]);
      p->lp_h_y.push_back(p->y_b[e]);
    }
  }
  while (p->lp_h_ent.size() % 64) {
    p->lp_h_ent.push_back(NONE | (NONE << 16));
    p->lp_h_row.push_back(NONE);
    p->lp_h_y.push_back(0.0);
    p->lp_h_y.push_back(0.0);
  }
  p->rh = (int32_t)(p->lp_h_ent.size() / 64);
  // ---- DC start: B' = Cft' diag(b) Cft on the same pattern (pypower makeBdc), the constant part of the right-hand
  // side per bus: Pbusinj (phase shifters) + Gs + sum over REF columns of B'_ir theta_r --------------------------------
  if (!p->br_bdc.empty()) {
    std::vector<std::map<int32_t, double>> bdc(nb);
    std::vector<double> cst(nb, 0.0);
    for (int32_t k = 0; k < p->nbr; ++k) {
      const double b = p->br_bdc[k];
      if (b == 0.0) continue;
      const int32_t f = p->br_f[k], t = p->br_t[k];
      bdc[f][f] += b; bdc[t][t] += b; bdc[f][t] -= b; bdc[t][f] -= b;
      cst[f] += p->br_pfinj[k]; cst[t] -= p->br_pfinj[k];
    }
    for (int i = 0; i < nb; ++i) {
      for (auto& kv : bdc[i]) if (p->bus_type[kv.first] == OPFX_REF && kv.first != i) cst[i] += kv.second * p->va_set[kv.first];
    }
    for (int i = 0; i < nb; ++i) cst[i] += p->gs_copy.empty() ? 0.0 : p->gs_copy[i];
    p->lp_dc.assign((size_t)p->ra * (KA + 2) * 64, 0.0);
    p->lp_hdc.assign((size_t)p->rh * 64, 0.0);
    auto val = [&](int i, int j) { auto it = bdc[i].find(j); return it == bdc[i].end() ? 0.0 : it->second; };
    for (int i = 0; i < nb; ++i) {
      const int r = i / 64, lane = i % 64;
      for (int k = 0; k < KA; ++k) {
        const uint32_t ent = p->lp_a_ent[((size_t)r * KA + k) * 64 + lane];
        const int j = (int)(ent & 0xFFFF);
        p->lp_dc[((size_t)r * (KA + 2) + k) * 64 + lane] = j == i ? 0.0 : val(i, j);
      }
      p->lp_dc[((size_t)r * (KA + 2) + KA) * 64 + lane] = val(i, i);
      p->lp_dc[((size_t)r * (KA + 2) + KA + 1) * 64 + lane] = cst[i];
    }
    for (size_t q = 0; q < p->lp_h_ent.size(); ++q) {
      const uint32_t ent = p->lp_h_ent[q];
      if ((ent & 0xFFFF) == NONE) continue;
      p->lp_hdc[q] = val((int)p->lp_h_row[q], (int)(ent & 0xFFFF));
    }
  }
  // ---- B ---------------------------------------------------------------------
  // per level: every (target, source) update term is one item; terms that share a
  // target may sit in the same round (the kernel accumulates with LDS atomics).
  // The forward substitution rides along: the term  y_i -= A_ik A_kk^-1 y_k  needs the same multiplier as the
  // block terms of the pair (i, k), and one of those always exists — the one on the diagonal block (i, i), because
  // the pattern is symmetric (A_ki is there whenever A_ik is).  That item carries the pair (i, k) in its third
  // word (lp_b2) and applies the multiplier to y_k as well: no separate right-hand-side items (1 of 7 items on
  // the 306-bus grid, 1 of 4 on the radial 144-bus grid), no second evaluation of A_kk^-1 for them.
  // Items with a rider first, so that a round is mostly homogeneous; right-hand-side terms that found no
  // carrier (none on a symmetric pattern) would follow as items of their own.
  // A rider lengthens its item (two more LDS reads, two more atomics), so it only pays where it saves a round:
  // each stream — one wavefront, teams of two and of four — decides per level by the rounds a wavefront walks
  // (`level_for`); a level whose items fit one round per wavefront either way keeps separate items.
  const int nlev = p->n_levels();
  std::vector<int32_t> b_bounds{0}, c_bounds;
  // An item: w0 = target | A_ik << 16, w1 = A_kk | A_kj << 16, w2 = rider i | k << 16 (host form), w3 = a SECOND column
  // target2 | A_kj2 << 16 of the same multiplier (round 4): the terms A_ij -= A_ik A_kk^-1 A_kj of one pair (i, k) share
  // -A_ik A_kk^-1, two thirds of an item's arithmetic and eight of its twelve LDS reads, so two of them travel in one item
  // (375 -> 219 block items on the 144-bus grid, 8 -> 5 factorisation rounds; 7 115 -> 3 834 on the 372-bus grid).
  struct Item3 { uint32_t w0, w1, w2, w3; };
  constexpr uint32_t NO_RIDER = NONE | (NONE << 16), NO_SECOND = NONE | (NONE << 16);
  // The items of one level in the four forms a stream may take them: with / without second columns, with / without riders.
  std::vector<std::vector<Item3>> lev_rhs(nlev);      // the forward substitution alone (chord steps)
  auto level_items = [&](int lev, bool pairs, bool riders) {
    const int t0 = p->lev_tptr[lev], t1 = p->lev_tptr[lev + 1];
    // right-hand-side terms of the level by the block A_ik they multiply with: (i, k)
    std::unordered_map<int32_t, std::pair<int32_t, int32_t>> rhs_of;
    if (riders)
      for (int t = t0; t < t1; ++t)
        if (p->tgt_blk[t] < 0)
          for (int s = p->tgt_sptr[t]; s < p->tgt_sptr[t + 1]; ++s) rhs_of[p->src_ik[s]] = {-1 - p->tgt_blk[t], p->src_kj[s]};
    // block terms by their multiplier block A_ik, in the order the targets come (the diagonal target first: it carries
    // the rider)
    struct Term { int32_t tb, kk, kj; bool diagonal; };
    std::vector<int32_t> ik_order;
    std::unordered_map<int32_t, std::vector<Term>> by_ik;
    for (int t = t0; t < t1; ++t) {
      const int tb = p->tgt_blk[t];
      if (tb < 0) continue;
      const bool diagonal = p->blk_row[tb] == p->blk_col[tb];
      for (int s = p->tgt_sptr[t]; s < p->tgt_sptr[t + 1]; ++s) {
        auto& v = by_ik[p->src_ik[s]];
        if (v.empty()) ik_order.push_back(p->src_ik[s]);
        if (diagonal) v.insert(v.begin(), Term{tb, p->src_kk[s], p->src_kj[s], true});
        else v.push_back(Term{tb, p->src_kk[s], p->src_kj[s], false});
      }
    }
    std::vector<Item3> carriers, plain, rhs_left;
    for (int32_t ik : ik_order) {
      const auto& v = by_ik[ik];
      for (size_t q = 0; q < v.size(); q += pairs ? 2 : 1) {
        const bool two = pairs && q + 1 < v.size();
        const Item3 it{(uint32_t)v[q].tb | ((uint32_t)ik << 16), (uint32_t)v[q].kk | ((uint32_t)v[q].kj << 16), NO_RIDER,
                       two ? ((uint32_t)v[q + 1].tb | ((uint32_t)v[q + 1].kj << 16)) : NO_SECOND};
        auto f = (q == 0 && v[0].diagonal) ? rhs_of.find(ik) : rhs_of.end();
        if (f != rhs_of.end() && f->second.first == p->blk_row[v[0].tb]) {
          carriers.push_back({it.w0, it.w1, (uint32_t)f->second.first | ((uint32_t)f->second.second << 16), it.w3});
          rhs_of.erase(f);
        } else {
          plain.push_back(it);
        }
      }
    }
    for (const Item3& it : lev_rhs[lev])
      if (!riders || rhs_of.count(it.w0 >> 16)) rhs_left.push_back(it);          // (riders: the terms without a carrier)
    std::vector<Item3> out = carriers;                                             // items with a rider first, rhs targets last
    out.insert(out.end(), plain.begin(), plain.end());
    out.insert(out.end(), rhs_left.begin(), rhs_left.end());
    return out;
  };
  for (int lev = 0; lev < nlev; ++lev)
    for (int t = p->lev_tptr[lev]; t < p->lev_tptr[lev + 1]; ++t) {
      const int tb = p->tgt_blk[t];
      if (tb >= 0) continue;
      for (int s = p->tgt_sptr[t]; s < p->tgt_sptr[t + 1]; ++s)
        lev_rhs[lev].push_back({(0x8000u | (uint32_t)(-1 - tb)) | ((uint32_t)p->src_ik[s] << 16),
                                (uint32_t)p->src_kk[s] | ((uint32_t)p->src_kj[s] << 16), NO_RIDER, NO_SECOND});
    }
  auto rounds_of = [](size_t n) { return (int)((n + 63) / 64); };
  // A stream of `nw` wavefronts takes a level in the plainest form that needs the fewest rounds PER WAVEFRONT: second columns
  // and riders make a round heavier (four more reads and atomics; the rider's two tests), which only pays where whole rounds
  // go — a wave team walks most levels in one round per wavefront either way (config 3, second columns everywhere: 1.78 ->
  // 1.85 ms although the team's rounds fell from 48 to 40).  plan_no_pairs 1: never second columns, 2: wherever there is one.
  auto level_for = [&](int lev, int nw, bool may_ride) {
    std::vector<Item3> best;
    int best_r = INT32_MAX;
    for (int form = 0; form < 4; ++form) {
      const bool pr = form & 2, rd = form & 1;
      if ((pr && p->dbg.plan_no_pairs == 1) || (rd && (!may_ride || p->dbg.plan_no_riders))) continue;
      if (!pr && p->dbg.plan_no_pairs == 2) continue;
      std::vector<Item3> its = level_items(lev, pr, rd);
      const int r = (rounds_of(its.size()) + nw - 1) / nw;
      if (r < best_r) { best_r = r; best = std::move(its); }
    }
    return best;
  };
  for (int lev = 0; lev < nlev; ++lev) {
    const std::vector<Item3> its = level_for(lev, 1, true);
    for (size_t o = 0; o < its.size(); o += 64)
      for (int lane = 0; lane < 64; ++lane) {
        if (o + lane < its.size()) { p->lp_b.push_back(its[o + lane].w0); p->lp_b.push_back(its[o + lane].w1); p->lp_b2.push_back(its[o + lane].w2); p->lp_b3.push_back(its[o + lane].w3); }
        else { p->lp_b.push_back(NONE | (NONE << 16)); p->lp_b.push_back(0); p->lp_b2.push_back(NO_RIDER); p->lp_b3.push_back(NO_SECOND); }
      }
    b_bounds.push_back((int32_t)(p->lp_b.size() / 128));
  }
  p->rb = (int32_t)(p->lp_b.size() / 128);
  // ---- C ---------------------------------------------------------------------
  // Back substitution in the SAME item form as the forward substitution of part B, column-oriented: once
  // the right-hand side y_j of the pivots j of a level is final, every earlier pivot k whose row holds a
  // block U_kj gets  y_k -= U_kj A_jj^-1 y_j  (an "rhs target" item: target k, block (k,j), pivot block
  // (j,j), source j).  One group of mutually independent items per level, walked from the last level to the
  // first; nothing is divided in place — the kernel's voltage update computes x_i = A_ii^-1 y_i for every
  // bus at the end.  (A row-oriented sweep needs a second group per level for pivots with many U-terms.)
  std::vector<std::array<uint32_t, 2>> items;
  std::vector<std::vector<std::array<int32_t, 2>>> col_terms(nb);      // j -> (k, block (k,j))
  for (size_t q = 0; q + 1 < p->piv_uptr.size(); ++q)
    for (int u = p->piv_uptr[q]; u < p->piv_uptr[q + 1]; ++u)
      col_terms[p->u_col[u]].push_back({p->piv_bus[q], p->u_blk[u]});
  auto flush_c = [&]() {
    for (size_t o = 0; o < items.size(); o += 64) {
      for (int lane = 0; lane < 64; ++lane) {
        if (o + lane < items.size()) { p->lp_c.push_back(items[o + lane][0]); p->lp_c.push_back(items[o + lane][1]); }
        else { p->lp_c.push_back(NONE | (NONE << 16)); p->lp_c.push_back(0); }
      }
    }
    items.clear();
  };
  for (int lev = nlev - 1; lev >= 0; --lev) {
    for (int q = p->lev_pptr[lev]; q < p->lev_pptr[lev + 1]; ++q) {
      const int j = p->piv_bus[q];
      for (auto& kt : col_terms[j])
        items.push_back({(0x8000u | (uint32_t)kt[0]) | ((uint32_t)kt[1] << 16),
                         (uint32_t)p->diag_blk[j] | ((uint32_t)j << 16)});
    }
    flush_c();
    c_bounds.push_back((int32_t)(p->lp_c.size() / 128));
  }
  p->rc = (int32_t)(p->lp_c.size() / 128);
  // The packed stream pads each part to a multiple of 4 rounds with empty items: the kernel
  // keeps 4 rounds in flight in a rotating register set and loads unconditionally.
  p->rb_pad = (p->rb + 3) & ~3;
  p->rc_pad = (p->rc + 3) & ~3;
  if (p->rb_pad + p->rc_pad < 4) p->rc_pad = 4;
  p->lp_groups = b_bounds;
  if (p->rb_pad > p->rb) p->lp_groups.push_back(p->rb_pad);
  for (int32_t cb : c_bounds) if (p->rb_pad + cb > p->lp_groups.back()) p->lp_groups.push_back(p->rb_pad + cb);
  if (p->rc_pad > p->rc) p->lp_groups.push_back(p->rb_pad + p->rc_pad);
  // ---- packed device forms ---------------------------------------------------------
  auto put_d = [](std::vector<uint32_t>& v, size_t at, double x) { std::memcpy(&v[at], &x, 8); };
  p->lp_bc.assign((size_t)(p->rb_pad + p->rc_pad) * 64 * 4, 0u);
  for (size_t q = 0; q < p->lp_bc.size(); q += 4) {
    p->lp_bc[q] = NONE | (NONE << 16); p->lp_bc[q + 1] = NONE | (NONE << 16); p->lp_bc[q + 2] = NONE | (NONE << 16); p->lp_bc[q + 3] = NO_RIDER_BITS;
  }
  // device words: w2 = the second column, w3 = round flags | rider bits
  for (int r = 0; r < p->rb; ++r)
    for (int l = 0; l < 64; ++l) {
      p->lp_bc[((size_t)r * 64 + l) * 4 + 0] = p->lp_b[((size_t)r * 64 + l) * 2];
      p->lp_bc[((size_t)r * 64 + l) * 4 + 1] = p->lp_b[((size_t)r * 64 + l) * 2 + 1];
      p->lp_bc[((size_t)r * 64 + l) * 4 + 2] = p->lp_b3[(size_t)r * 64 + l];
      p->lp_bc[((size_t)r * 64 + l) * 4 + 3] = rider_bits(p->lp_b2[(size_t)r * 64 + l]);
    }
  for (int r = 0; r < p->rc; ++r)
    for (int l = 0; l < 64; ++l)
      for (int w = 0; w < 2; ++w)
        p->lp_bc[((size_t)(p->rb_pad + r) * 64 + l) * 4 + w] = p->lp_c[((size_t)r * 64 + l) * 2 + w];
  // word 3 of a round's items: bit 0 = workgroup barrier after the round (wave teams only), bit 1 = the
  // next round of this wavefront belongs to the same group, i.e. does not depend on this one (the kernel
  // then issues its LDS reads before this round computes)
  for (size_t g = 0; g + 1 < p->lp_groups.size(); ++g)
    for (int r = p->lp_groups[g]; r + 1 < p->lp_groups[g + 1]; ++r)
      for (int l = 0; l < 64; ++l) p->lp_bc[((size_t)r * 64 + l) * 4 + 3] |= 2u;
  const bool spread = !p->dbg.plan_no_bank;
  if (spread)
    for (size_t g = 0; g + 1 < p->lp_groups.size(); ++g)
      if (p->lp_groups[g + 1] > p->lp_groups[g])
        spread_over_banks(&p->lp_bc[(size_t)p->lp_groups[g] * 256], p->lp_groups[g + 1] - p->lp_groups[g], p->n_full);
  // ---- the chord stream (opfx_solve_opts.jacobian_reuse_tol): an iteration that keeps the factorisation of an earlier
  // one runs the FORWARD SUBSTITUTION alone — the right-hand-side items of every level, which read A_ik and A_kk as the
  // factorisation left them — and then the same back substitution: [rf_pad rounds F | the C rounds of lp_bc] ----------
  {
    std::vector<uint32_t> f;
    std::vector<int32_t> f_bounds{0};
    for (int lev = 0; lev < nlev; ++lev) {
      const std::vector<Item3>& its = lev_rhs[lev];
      for (size_t o = 0; o < its.size(); o += 64)
        for (int lane = 0; lane < 64; ++lane) {
          if (o + lane < its.size()) { f.push_back(its[o + lane].w0); f.push_back(its[o + lane].w1); f.push_back(NO_SECOND); f.push_back(NO_RIDER_BITS); }
          else { f.push_back(NONE | (NONE << 16)); f.push_back(NONE | (NONE << 16)); f.push_back(NO_SECOND); f.push_back(NO_RIDER_BITS); }
        }
      if ((int32_t)(f.size() / 256) > f_bounds.back()) f_bounds.push_back((int32_t)(f.size() / 256));
    }
    p->rf = (int32_t)(f.size() / 256);
    p->rf_pad = (p->rf + 3) & ~3;
    if (p->rf_pad + p->rc_pad < 4) p->rf_pad = 4;
    if (spread)
      for (size_t g = 0; g + 1 < f_bounds.size(); ++g) spread_over_banks(&f[(size_t)f_bounds[g] * 256], f_bounds[g + 1] - f_bounds[g], p->n_full);
    while ((int32_t)(f.size() / 256) < p->rf_pad)
      for (int l = 0; l < 64; ++l) { f.push_back(NONE | (NONE << 16)); f.push_back(NONE | (NONE << 16)); f.push_back(NO_SECOND); f.push_back(NO_RIDER_BITS); }
    p->lp_bcc = f;
    p->lp_bcc.insert(p->lp_bcc.end(), p->lp_bc.begin() + (size_t)p->rb_pad * 256, p->lp_bc.end());
  }
  // ---- the dense tail ---------------------------------------------------------------------------
  // A meshed grid ends in a chain of levels with ONE pivot each (the last separator fills in completely:
  // 16 such levels on the 306-bus grid, 20 on the 372-bus one).  Their back substitution is a strictly serial
  // chain of tiny groups — one LDS round trip and one 2x2 inverse per level with a handful of live lanes.  The
  // wave teams run that chain in REGISTERS instead (opfx.hip: tail_solve): lane e holds y of tail pivot e and
  // its diagonal block, x_s travels by v_readlane, the U-blocks are plain LDS reads that do not depend on the
  // chain.  The plan hands over the tail pivots' buses and, per tail row, the ids of its U-blocks inside the
  // tail; the U-terms of tail COLUMNS in rows outside the tail follow as one ordinary group.
  const int tail_m = tail_length(p);
  p->tail_m = tail_m;
  constexpr int TAIL_MAX = opfx_plan::TAIL_MAX;
  const int tail_M = (tail_m + 7) & ~7;                  // steps of the unrolled chain the kernels instantiate (8, 16, 24, 32)
  p->tail_bus.assign(TAIL_MAX, 0u);
  p->tail_ids.assign(tail_m > 0 ? (size_t)(tail_m + 1) * tail_M : 0, (uint16_t)NONE);      // (+ a row of "none" for the idle lanes)
  std::vector<int> tail_pos(nb, -1);
  for (int e = 0; e < tail_m; ++e) {
    const int bus = p->piv_bus[p->lev_pptr[nlev - tail_m + e]];
    p->tail_bus[e] = (uint32_t)bus | ((uint32_t)p->diag_blk[bus] << 16);
    tail_pos[bus] = e;
  }
  for (int sx = 0; sx < tail_m; ++sx)
    for (auto& kt : col_terms[p->tail_bus[sx] & 0xFFFFu])
      if (tail_pos[kt[0]] >= 0) p->tail_ids[(size_t)tail_pos[kt[0]] * tail_M + sx] = (uint16_t)kt[1];
  // the LOWER triangle: entry [e][s], e > s, = L-block (row e, column s) — the forward substitution through the tail as a
  // register chain (chord iterations, opfx.hip tail_chain FWD); read off the right-hand-side items of the tail levels
  for (int lev = nlev - tail_m; lev < nlev; ++lev)
    for (const Item3& it : lev_rhs[lev]) {
      const int i = (int)(it.w0 & 0x7FFFu), k = (int)(it.w1 >> 16);
      if (tail_pos[i] > tail_pos[k] && tail_pos[k] >= 0) p->tail_ids[(size_t)tail_pos[i] * tail_M + tail_pos[k]] = (uint16_t)(it.w0 >> 16);
    }
  p->tail_ids32.assign(p->tail_ids.begin(), p->tail_ids.end());
  // back-substitution rounds of the team streams: (register chain) [tail columns -> outside rows] [levels below the tail]
  std::vector<std::vector<std::array<uint32_t, 2>>> tc_groups;      // item lists, one per group (materialised per team size below)
  auto tc_flush = [&]() {
    if (!items.empty()) tc_groups.push_back(items);
    items.clear();
  };
  if (tail_m > 0) {
    for (int sx = 0; sx < tail_m; ++sx) {
      const int j = (int)(p->tail_bus[sx] & 0xFFFFu);
      for (auto& kt : col_terms[j])
        if (tail_pos[kt[0]] < 0)
          items.push_back({(0x8000u | (uint32_t)kt[0]) | ((uint32_t)kt[1] << 16), (uint32_t)p->diag_blk[j] | ((uint32_t)j << 16)});
    }
    tc_flush();
    for (int lev = nlev - tail_m - 1; lev >= 0; --lev) {
      for (int q = p->lev_pptr[lev]; q < p->lev_pptr[lev + 1]; ++q) {
        const int j = p->piv_bus[q];
        for (auto& kt : col_terms[j])
          items.push_back({(0x8000u | (uint32_t)kt[0]) | ((uint32_t)kt[1] << 16), (uint32_t)p->diag_blk[j] | ((uint32_t)j << 16)});
      }
      tc_flush();
    }
  }
  for (int t = 0; t < 2; ++t) {
    const int NW = t == 0 ? 2 : 4;
    std::vector<uint32_t>& out = p->lp_team[t];
    // real groups (padding ranges dropped): (source, first round, end round); source 0 = lp_bc, 1 = tc,
    // 2 = tbk (this team's factorisation rounds).
    // With a tail the stream has two parts, each padded to a multiple of 4 rounds: factorisation + forward
    // substitution | back substitution below the tail; the kernels run the register chain between them
    // (a workgroup barrier follows it), so the last round of the first part needs a barrier only if another
    // wavefront than 0 took part in its group.
    struct Group { int src, r0, r1; };
    std::vector<Group> groups;
    // factorisation + forward substitution: per level the item form that needs fewer rounds per wavefront of THIS
    // team; items with a rider are dealt round-robin over the level's rounds so that no wavefront gets all of them
    // (Dealing a group's items over a whole number of rounds PER WAVEFRONT — 105 items as 27 + 26 + 26 + 26 on four
    // wavefronts instead of 64 + 41 on two — was measured and is slower, config 3 1.85 -> 1.89 ms: a round's fixed cost
    // outweighs its lanes' LDS passes.)
    std::vector<uint32_t> tbk;
    for (int lev = 0; lev < nlev; ++lev) {
      const std::vector<Item3> its = level_for(lev, NW, false);          // (the team kernels' items take no riders, opfx.hip item_factor)
      const int nr = rounds_of(its.size());
      if (nr == 0) continue;
      const int first = (int)(tbk.size() / 256);
      tbk.resize(tbk.size() + (size_t)nr * 256, NONE | (NONE << 16));
      for (size_t q = 0; q < its.size(); ++q) {
        uint32_t* at = &tbk[((size_t)(first + q / 64) * 64 + q % 64) * 4];
        at[0] = its[q].w0; at[1] = its[q].w1; at[2] = its[q].w3;          // (word 2 of a device item: the second column)
      }
      if (spread) spread_over_banks(&tbk[(size_t)first * 256], nr, p->n_full);
      // shared slots: the ids to be zeroed during this level ride on its items, one each (bits 2-16 of word 3, which a
      // team item does not use for a rider; plan.cpp share_slots made sure the level has enough items)
      for (size_t q = first * 64; q < (size_t)(first + nr) * 64; ++q) tbk[q * 4 + 3] = NO_RIDER_BITS;
      {
        size_t at = (size_t)first * 64;
        for (size_t z = 0; z < p->zero_lev.size(); ++z) {
          if (p->zero_lev[z] != lev) continue;
          while (at < (size_t)(first + nr) * 64 && (tbk[at * 4] & 0xFFFF) == NONE) ++at;
          if (at >= (size_t)(first + nr) * 64) { p->ra = p->rb = p->rc = -1; return; }     // (cannot happen: capacity was counted)
          tbk[at * 4 + 3] = ((uint32_t)p->zero_id[z] << 2) | (RIDER_NONE15 << 17);
          ++at;
        }
      }
      groups.push_back({2, first, first + nr});
    }
    std::vector<uint32_t> tc;                               // this team's back-substitution rounds below the tail
    for (const auto& its : tc_groups) {
      const int nr = rounds_of(its.size());
      const int first = (int)(tc.size() / 256);
      tc.resize(tc.size() + (size_t)nr * 256, NONE | (NONE << 16));
      for (size_t q = 0; q < its.size(); ++q) {
        uint32_t* at = &tc[((size_t)(first + q / 64) * 64 + q % 64) * 4];
        at[0] = its[q][0]; at[1] = its[q][1];
      }
      if (spread) spread_over_banks(&tc[(size_t)first * 256], nr, p->n_full);
    }
    if (tail_m == 0)
      for (size_t g = 0; g + 1 < p->lp_groups.size(); ++g) {     // back substitution: the rounds of lp_bc
        const int r0 = p->lp_groups[g], r1 = p->lp_groups[g + 1];
        if (r0 == r1 || r0 < p->rb_pad || r0 >= p->rb_pad + p->rc) continue;
        groups.push_back({0, r0, r1});
      }
    const size_t n_first = tail_m > 0 ? groups.size() : 0;
    if (tail_m > 0) {
      int first = 0;
      for (const auto& its : tc_groups) { const int nr = rounds_of(its.size()); groups.push_back({1, first, first + nr}); first += nr; }
    }
    // forward substitution alone, per level (chord steps): same dealing, same barrier rules
    std::vector<uint32_t> tfk;
    std::vector<Group> groups_c;
    for (int lev = 0; lev < nlev - tail_m; ++lev) {      // (the tail's own forward substitution: the register chain)
      const std::vector<Item3>& its = lev_rhs[lev];
      const int nr = rounds_of(its.size());
      if (nr == 0) continue;
      const int first = (int)(tfk.size() / 256);
      tfk.resize(tfk.size() + (size_t)nr * 256, NONE | (NONE << 16));
      for (size_t q = 0; q < its.size(); ++q) {
        uint32_t* at = &tfk[((size_t)(first + q / 64) * 64 + q % 64) * 4];
        at[0] = its[q].w0; at[1] = its[q].w1; at[2] = its[q].w3;
      }
      if (spread) spread_over_banks(&tfk[(size_t)first * 256], nr, p->n_full);
      groups_c.push_back({3, first, first + nr});
    }
    const size_t n_first_c = tail_m > 0 ? groups_c.size() : 0;
    for (size_t g = (tail_m > 0 ? n_first : 0); g < groups.size(); ++g)
      if (groups[g].src != 2) groups_c.push_back(groups[g]);          // the back-substitution groups of the main stream
    auto emit = [&](const std::vector<Group>& gs, size_t n_first_, std::vector<uint32_t>& out_, int32_t& K_out, int32_t& kb_out, int32_t& barriers_out) {
      int K = 0;
      auto empty_round = [&](uint32_t flags) {
        for (int w = 0; w < NW; ++w)
          for (int l = 0; l < 64; ++l) { out_.push_back(NONE | (NONE << 16)); out_.push_back(NONE | (NONE << 16)); out_.push_back(NONE | (NONE << 16)); out_.push_back(flags | NO_RIDER_BITS); }
        ++K;
      };
      kb_out = -1;
      for (size_t g = 0; g < gs.size(); ++g) {
        const int r0 = gs[g].r0, r1 = gs[g].r1;
        const int per = (r1 - r0 + NW - 1) / NW;
        // A group of ONE round runs on wavefront 0 alone; when the next group is such a group too, the same
        // wavefront carries on and its LDS operations execute in issue order: no workgroup barrier between them
        // (the dense tail of a meshed grid is a long chain of one-round groups).
        const bool single = r1 - r0 == 1;
        if (tail_m > 0 && g == n_first_) {                         // part boundary: the chain runs here
          while (K % 4 || K < 4) empty_round(0u);
          kb_out = K;
        }
        const bool last_of_part = tail_m > 0 && g + 1 == n_first_; // (the chain runs on wavefront 0, like a one-round group)
        const bool next_single = last_of_part || (g + 1 < gs.size() && gs[g + 1].r1 - gs[g + 1].r0 == 1);
        const uint32_t* src = gs[g].src == 0 ? p->lp_bc.data() : (gs[g].src == 1 ? tc.data() : (gs[g].src == 2 ? tbk.data() : tfk.data()));
        for (int j = 0; j < per; ++j) {
          const uint32_t flags = ((j == per - 1 && !(single && next_single)) ? 1u : 0u) | (j < per - 1 ? 2u : 0u);
          barriers_out += (int32_t)(flags & 1u);
          for (int w = 0; w < NW; ++w) {
            const int r = r0 + j * NW + w;
            for (int l = 0; l < 64; ++l) {
              if (r < r1) for (int q = 0; q < 3; ++q) out_.push_back(src[((size_t)r * 64 + l) * 4 + q]);
              else for (int q = 0; q < 3; ++q) out_.push_back(NONE | (NONE << 16));
              // (word 3: the round's flags | for a factorisation item of a plan with shared slots the id it zeroes)
              out_.push_back(flags | ((r < r1 && gs[g].src == 2) ? (src[((size_t)r * 64 + l) * 4 + 3] & ~3u) : NO_RIDER_BITS));
            }
          }
          ++K;
        }
      }
      if (tail_m > 0 && kb_out < 0) {                              // (no back-substitution group at all)
        while (K % 4 || K < 4) empty_round(0u);
        kb_out = K;
      }
      while (K % 4 || K < 4) empty_round(0u);
      K_out = K;
      if (tail_m == 0) kb_out = K;
    };
    emit(groups, n_first, out, p->team_rounds[t], p->team_kb[t], p->team_barriers[t]);
    emit(groups_c, n_first_c, p->lp_teamc[t], p->team_rounds_c[t], p->team_kb_c[t], p->team_barriers_c[t]);
    p->n_groups = (int32_t)groups.size();
  }
  static_assert(KA >= 1 && KA <= 3, "entries and the diagonal-block word share one 16-byte vector");
  constexpr int NV = opfx_plan::APK_VECS;
  p->lp_apk.assign((size_t)p->ra * NV * 64 * 4, 0u);
  for (int r = 0; r < p->ra; ++r)
    for (int l = 0; l < 64; ++l) {
      auto at = [&](int v) { return (((size_t)r * NV + v) * 64 + l) * 4; };
      for (int k = 0; k < KA; ++k) {
        const size_t o = ((size_t)r * KA + k) * 64 + l;
        p->lp_apk[at(0) + k] = p->lp_a_ent[o];
        put_d(p->lp_apk, at(1 + k), p->lp_a_y[o * 2]);
        put_d(p->lp_apk, at(1 + k) + 2, p->lp_a_y[o * 2 + 1]);
      }
      p->lp_apk[at(0) + 3] = p->lp_a_dblk[(size_t)r * 64 + l];
      put_d(p->lp_apk, at(1 + KA), p->lp_a_ydiag[((size_t)r * 64 + l) * 2]);
      put_d(p->lp_apk, at(1 + KA) + 2, p->lp_a_ydiag[((size_t)r * 64 + l) * 2 + 1]);
    }
  p->lp_hpk.assign((size_t)p->rh * 2 * 64 * 4, 0u);
  for (int h = 0; h < p->rh; ++h)
    for (int l = 0; l < 64; ++l) {
      const size_t a0 = (((size_t)h * 2 + 0) * 64 + l) * 4, a1 = (((size_t)h * 2 + 1) * 64 + l) * 4;
      put_d(p->lp_hpk, a0, p->lp_h_y[((size_t)h * 64 + l) * 2]);
      put_d(p->lp_hpk, a0 + 2, p->lp_h_y[((size_t)h * 64 + l) * 2 + 1]);
      p->lp_hpk[a1] = p->lp_h_ent[(size_t)h * 64 + l];
      p->lp_hpk[a1 + 1] = p->lp_h_row[(size_t)h * 64 + l];
    }
}

}  // namespace

// How the level-scheduled minimum-degree elimination breaks its ties: `slack` = a level takes every independent vertex of
// degree <= dmin + slack; `tie_seed` = order of candidates of equal degree (0: by bus number, else by a seeded hash).
struct PlanKnobs { int slack = 2; unsigned tie_seed = 0; };

static int plan_build(const opfx_case* c, const PlanKnobs& knobs, const opfx_debug_opts& dbg, opfx_plan** out) {
  if (!c || !out) { opfx_set_error("opfx_plan_create: null argument"); return OPFX_ERR_INVALID; }
  if (c->nb <= 0 || c->nbr < 0 || !c->bus_type || !c->vm_set || !c->va_set ||
      (c->nbr > 0 && (!c->br_f || !c->br_t || !c->br_y))) {
    opfx_set_error("opfx_plan_create: incomplete case descriptor");
    return OPFX_ERR_INVALID;
  }
  const int32_t nb = c->nb, nbr = c->nbr;
  auto* p = new opfx_plan();
  p->dbg = dbg;
  p->nb = nb; p->nbr = nbr; p->base_mva = c->base_mva;
  p->bus_type.assign(c->bus_type, c->bus_type + nb);
  p->vm_set.assign(c->vm_set, c->vm_set + nb);
  p->va_set.assign(c->va_set, c->va_set + nb);
  p->br_f.assign(c->br_f, c->br_f + nbr);
  p->br_t.assign(c->br_t, c->br_t + nbr);
  p->br_y.assign(c->br_y, c->br_y + (size_t)nbr * 8);
  p->br_kf.assign(nbr, 0.0); p->br_kt.assign(nbr, 0.0);
  if (c->br_kf) p->br_kf.assign(c->br_kf, c->br_kf + nbr);
  if (c->br_kt) p->br_kt.assign(c->br_kt, c->br_kt + nbr);
  if (c->br_bdc && c->br_pfinj) { p->br_bdc.assign(c->br_bdc, c->br_bdc + nbr); p->br_pfinj.assign(c->br_pfinj, c->br_pfinj + nbr); }
  p->ref_ord.assign(nb, -1);
  for (int32_t i = 0; i < nb; ++i) {
    int32_t t = p->bus_type[i];
    if (t == OPFX_REF) { p->ref_ord[i] = (int32_t)p->ref_bus.size(); p->ref_bus.push_back(i); }
    else if (t == OPFX_PV) p->npv++;
    else if (t == OPFX_PQ) p->npq++;
    else { delete p; opfx_set_error("opfx_plan_create: bad bus_type"); return OPFX_ERR_INVALID; }
  }
  p->nref = (int32_t)p->ref_bus.size();
  if (p->nref == 0) { delete p; opfx_set_error("opfx_plan_create: no REF bus"); return OPFX_ERR_SINGULAR; }
  for (int32_t k = 0; k < nbr; ++k)
    if (p->br_f[k] < 0 || p->br_f[k] >= nb || p->br_t[k] < 0 || p->br_t[k] >= nb ||
        p->br_f[k] == p->br_t[k]) {
      delete p; opfx_set_error("opfx_plan_create: bad branch end"); return OPFX_ERR_INVALID;
    }

  // ---- Ybus (makeYbus): block CSR with sorted columns -------------
  std::vector<std::map<int32_t, std::pair<double, double>>> rows(nb);
  for (int32_t i = 0; i < nb; ++i)
    rows[i][i] = {c->gs ? c->gs[i] : 0.0, c->bs ? c->bs[i] : 0.0};
  if (c->gs) p->gs_copy.assign(c->gs, c->gs + nb);
  // a branch without coupling terms (open-ended in the net itself: a shunt at its connected end, case.py
  // open_ended_stamps) adds to the diagonal only: no structural (f,t) entry, no edge for connectivity
  std::vector<char> coupled(nbr, 1);
  for (int32_t k = 0; k < nbr; ++k) {
    const double* y = &p->br_y[(size_t)k * 8];
    int32_t f = p->br_f[k], t = p->br_t[k];
    coupled[k] = (y[2] != 0.0 || y[3] != 0.0 || y[4] != 0.0 || y[5] != 0.0) ? 1 : 0;
    auto add = [&](int32_t i, int32_t j, double g, double b) {
      auto& e = rows[i][j]; e.first += g; e.second += b; };
    add(f, f, y[0], y[1]); add(t, t, y[6], y[7]);
    if (coupled[k]) { add(f, t, y[2], y[3]); add(t, f, y[4], y[5]); }
  }
  p->y_ptr.assign(nb + 1, 0);
  p->y_diag.assign(nb, -1);
  for (int32_t i = 0; i < nb; ++i) {
    p->y_ptr[i] = (int32_t)p->y_col.size();
    for (auto& kv : rows[i]) {
      if (kv.first == i) p->y_diag[i] = (int32_t)p->y_col.size();
      p->y_col.push_back(kv.first);
      p->y_g.push_back(kv.second.first);
      p->y_b.push_back(kv.second.second);
    }
  }
  p->y_ptr[nb] = (int32_t)p->y_col.size();
  auto ypos = [&](int32_t i, int32_t j) {
    auto b = p->y_col.begin() + p->y_ptr[i], e = p->y_col.begin() + p->y_ptr[i + 1];
    return (int32_t)(std::lower_bound(b, e, j) - p->y_col.begin());
  };
  p->br_pos.resize((size_t)nbr * 4);
  for (int32_t k = 0; k < nbr; ++k) {
    int32_t f = p->br_f[k], t = p->br_t[k];
    p->br_pos[k * 4 + 0] = ypos(f, f); p->br_pos[k * 4 + 1] = coupled[k] ? ypos(f, t) : -1;
    p->br_pos[k * 4 + 2] = coupled[k] ? ypos(t, f) : -1; p->br_pos[k * 4 + 3] = ypos(t, t);
  }

  // ---- islanding outages: the Newton matrix of such a case is singular; the solvers report
  // them as not converged at once (pandapower would de-energise the island instead) ----------
  {
    std::vector<std::vector<std::pair<int32_t, int32_t>>> nbrs(nb);      // (other bus, branch)
    for (int32_t k = 0; k < nbr; ++k) if (coupled[k]) { nbrs[p->br_f[k]].push_back({p->br_t[k], k}); nbrs[p->br_t[k]].push_back({p->br_f[k], k}); }
    p->br_island.assign(nbr, 0);
    p->isl_ptr.assign(1, 0);
    std::vector<char> seen(nb);
    std::vector<int32_t> stack;
    for (int32_t out = 0; out < nbr; ++out) {
      std::fill(seen.begin(), seen.end(), 0);
      stack.clear();
      for (int32_t r : p->ref_bus) { seen[r] = 1; stack.push_back(r); }
      int32_t reached = (int32_t)stack.size();
      while (!stack.empty()) {
        const int32_t u = stack.back(); stack.pop_back();
        for (auto& e : nbrs[u]) if (e.second != out && !seen[e.first]) { seen[e.first] = 1; ++reached; stack.push_back(e.first); }
      }
      p->br_island[out] = reached < nb ? 1 : 0;
      if (reached < nb) for (int32_t i = 0; i < nb; ++i) if (!seen[i]) p->isl_bus.push_back(i);
      p->isl_ptr.push_back((int32_t)p->isl_bus.size());
    }
  }

  // ---- Jacobian block pattern on non-REF buses -------------------------------
  BlockMap bm{nb, {}, &p->blk_row, &p->blk_col};
  std::vector<std::set<int32_t>> adj(nb);
  auto is_ref = [&](int32_t i) { return p->bus_type[i] == OPFX_REF; };
  p->y_blk.assign(p->y_col.size(), -1);
  p->diag_blk.assign(nb, -1);
  for (int32_t i = 0; i < nb; ++i) {
    if (is_ref(i)) continue;
    for (int32_t e = p->y_ptr[i]; e < p->y_ptr[i + 1]; ++e) {
      int32_t j = p->y_col[e];
      if (is_ref(j)) continue;
      p->y_blk[e] = bm.get_or_add(i, j, nullptr);
      if (i == j) p->diag_blk[i] = p->y_blk[e];
      else { adj[i].insert(j); adj[j].insert(i); }
      // pypower scalar Jacobian non-zeros (for the byte model)
      bool ipq = p->bus_type[i] == OPFX_PQ, jpq = p->bus_type[j] == OPFX_PQ;
      p->nnz_j += 1 + (jpq ? 1 : 0) + (ipq ? 1 : 0) + (ipq && jpq ? 1 : 0);
    }
  }
  // symmetric pattern is required by the block elimination
  for (int32_t i = 0; i < nb; ++i)
    for (int32_t j : adj[i])
      if (bm.get(i, j) < 0) {
        bool added; int32_t id = bm.get_or_add(i, j, &added);
        p->fill_blk.push_back(id);
      }

  // ---- level-scheduled multiple-minimum-degree elimination --------------------
  std::vector<char> alive(nb, 0), blocked(nb, 0), last(nb, 0);
  int32_t n_alive = 0, n_first = 0;            // n_first: alive buses that are not held back (opfx_case.elim_last)
  for (int32_t i = 0; i < nb; ++i) if (!is_ref(i)) { alive[i] = 1; ++n_alive; last[i] = (c->elim_last && c->elim_last[i]) ? 1 : 0; n_first += !last[i]; }
  p->lev_tptr.push_back(0);
  p->lev_pptr.push_back(0);
  p->tgt_sptr.push_back(0);
  p->piv_uptr.push_back(0);
  std::vector<int32_t> cand;
  while (n_alive > 0) {
    cand.clear();
    size_t dmin = SIZE_MAX;
    // (buses held back by opfx_case.elim_last are no candidates while any other bus is left)
    for (int32_t i = 0; i < nb; ++i) if (alive[i] && (n_first == 0 || !last[i])) { cand.push_back(i); dmin = std::min(dmin, adj[i].size()); }
    // a level takes every independent vertex of degree <= dmin + slack: on meshed grids a slack of 2 cuts the number
    // of levels by a quarter at no extra fill
    const int slack = knobs.slack;
    const unsigned tie_seed = knobs.tie_seed;
    size_t dcap = std::max<size_t>(2, dmin + slack);
    if (tie_seed) {
      auto h = [&](int32_t v) { unsigned x = (unsigned)v * 2654435761u + tie_seed * 40503u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; return x; };
      std::sort(cand.begin(), cand.end(), [&](int32_t a, int32_t b) {
        if (adj[a].size() != adj[b].size()) return adj[a].size() < adj[b].size();
        return h(a) < h(b); });
    } else
    std::stable_sort(cand.begin(), cand.end(), [&](int32_t a, int32_t b) {
      return adj[a].size() < adj[b].size(); });
    std::fill(blocked.begin(), blocked.end(), 0);
    std::vector<int32_t> piv;
    for (int32_t k : cand) {
      if (adj[k].size() > dcap) break;
      if (blocked[k]) continue;
      piv.push_back(k);
      blocked[k] = 1;
      for (int32_t j : adj[k]) blocked[j] = 1;
    }
    // targets of this level, grouped: key = block id (>=0) or -1-bus (rhs)
    std::map<int32_t, std::vector<std::array<int32_t, 3>>> tg;
    std::vector<std::pair<int32_t, int32_t>> new_edges;
    for (int32_t k : piv) {
      const int32_t kk = p->diag_blk[k];
      std::vector<int32_t> nbrs(adj[k].begin(), adj[k].end());
      p->piv_bus.push_back(k);
      for (int32_t j : nbrs) { p->u_blk.push_back(bm.get(k, j)); p->u_col.push_back(j); }
      p->piv_uptr.push_back((int32_t)p->u_blk.size());
      for (int32_t i : nbrs) {
        const int32_t ik = bm.get(i, k);
        tg[-1 - i].push_back({ik, kk, k});
        for (int32_t j : nbrs) {
          bool added = false;
          int32_t ij = bm.get_or_add(i, j, &added);
          if (added) { p->fill_blk.push_back(ij); if (i < j) new_edges.push_back({i, j}); }
          if (i == j && p->diag_blk[i] < 0) p->diag_blk[i] = ij;
          tg[ij].push_back({ik, kk, bm.get(k, j)});
        }
      }
    }
    // heavy targets first so that a wave round holds similar trip counts
    std::vector<std::pair<int32_t, const std::vector<std::array<int32_t, 3>>*>> order;
    for (auto& kv : tg) order.push_back({kv.first, &kv.second});
    std::stable_sort(order.begin(), order.end(), [](auto& a, auto& b) {
      return a.second->size() > b.second->size(); });
    for (auto& o : order) {
      p->tgt_blk.push_back(o.first);
      for (auto& s : *o.second) { p->src_ik.push_back(s[0]); p->src_kk.push_back(s[1]); p->src_kj.push_back(s[2]); }
      p->tgt_sptr.push_back((int32_t)p->src_ik.size());
    }
    p->lev_tptr.push_back((int32_t)p->tgt_blk.size());
    p->lev_pptr.push_back((int32_t)p->piv_bus.size());
    p->max_level_width = std::max<int32_t>(p->max_level_width, (int32_t)order.size());
    p->max_level_width = std::max<int32_t>(p->max_level_width, (int32_t)piv.size());
    for (auto& e : new_edges) { adj[e.first].insert(e.second); adj[e.second].insert(e.first); }
    for (int32_t k : piv) {
      for (int32_t j : adj[k]) adj[j].erase(k);
      adj[k].clear();
      alive[k] = 0; --n_alive;
      n_first -= !last[k];
    }
  }
  p->n_blk = (int32_t)p->blk_row.size();
  renumber_blocks(p);
  if (dbg.plan_share_slots > 0 && nb <= 0x7FFF) share_slots(p);
  build_lane_programs(p);
  *out = p;
  return OPFX_OK;
}

// What a Newton iteration of this plan costs, in rounds of the kernel that will run it: the wave teams walk
// team_rounds[1] rounds per wavefront with team_barriers[1] workgroup barriers between them (grids whose instance takes
// more than half a CU's LDS: two instances per CU, teams of four), the single-wave kernel rb + rc rounds.
static double plan_cost(const opfx_plan* p) {
  const size_t nbe = (size_t)(p->nb + 1) & ~(size_t)1;
  const size_t lds_doubles = 4 * nbe + 2 * (((size_t)p->n_blk + 1) & ~(size_t)1) + 2 * (((size_t)p->n_full + 1) & ~(size_t)1);
  const double items = 1e-4 * (double)p->src_ik.size();          // (equal rounds: the plan with fewer update terms)
  // a plan with shared slots is asked for to bring a THIRD instance into the CU, as teams of two: rounds of that stream, and a
  // plan that stays above a third of the LDS (less ~2.7 KB of an environment's own arrays) is no candidate
  if (p->dbg.plan_share_slots > 0 && p->n_shared > 0)
    return p->team_rounds[0] + 0.25 * p->team_barriers[0] + items + (lds_doubles * 8 > 50 * 1024 + 512 ? 1000.0 : 0.0);
  if (lds_doubles * 8 > 52 * 1024) return p->team_rounds[1] + 0.25 * p->team_barriers[1] + items;   // <= 2 instances per CU
  if (lds_doubles * 8 > 31 * 1024) return p->team_rounds[0] + 0.25 * p->team_barriers[0] + items;   // <= 4
  return p->rb + p->rc + items;
}

// The elimination order decides how many rounds and barriers an iteration takes, and on meshed grids the outcome of the
// minimum-degree heuristic moves by 5-10 % with the way it breaks ties (306-bus HV grid: 48 to 56 rounds per wavefront
// over 24 tie-breaking rules).  A plan is compiled once per grid and runs millions of times: for grids of the wave-team
// kernels several rules are tried and the cheapest plan is kept (15 to 63 extra plans by default, by grid size).  Radial MV / LV grids are item-bound, not
// level-bound, and every rule gives the same round count: no search below 200 buses; above 800 (a plan takes most of a
// second to build) only on request (opfx_debug_opts.plan_search / plan_dcap_slack / plan_seed, include/opfx_debug.h).
extern "C" int opfx_plan_create_debug(const opfx_case* c_in, const opfx_debug_opts* dbg_in, opfx_plan** out) {
  opfx_case cs;
  int rc = opfx_take(c_in, &cs, "opfx_plan_create(opfx_case)");
  if (rc != OPFX_OK) return rc;
  const opfx_case* c = &cs;
  opfx_debug_opts dbg{};
  if (dbg_in && (rc = opfx_take(dbg_in, &dbg, "opfx_plan_create_debug(opfx_debug_opts)")) != OPFX_OK) return rc;
  PlanKnobs base;
  const bool pinned = dbg.plan_dcap_slack > 0 || dbg.plan_seed > 0;
  if (dbg.plan_dcap_slack > 0) base.slack = dbg.plan_dcap_slack;
  if (dbg.plan_seed > 0) base.tie_seed = (unsigned)dbg.plan_seed;
  opfx_plan* best = nullptr;
  rc = plan_build(c, base, dbg, &best);
  if (rc != OPFX_OK) return rc;
  // (default: as many extra plans as about two seconds of host time buy — 63 up to ~400 buses, 30 at 800; round 6: the 372-bus grid
  //  of config 5 finds its 48-round plan among the first 31, 52 rounds among 15: 205.6 -> 201.0 ms)
  const int n_default = std::max(15, std::min(63, 24000 / std::max(1, c->nb)));
  const int n_search = dbg.plan_search > 0 ? dbg.plan_search : (dbg.plan_search < 0 ? 0 : n_default);
  if (!pinned && best->nb >= 200 && (best->nb <= 800 || dbg.plan_search > 0) && n_search > 0) {
    double best_cost = plan_cost(best);
    for (int t = 0; t < n_search; ++t) {
      PlanKnobs k;
      k.slack = 2 + (t & 1);                       // slack 2 and 3 in turn
      k.tie_seed = (unsigned)(t / 2 + (k.slack == 2 ? 1 : 0));   // (slack 2 / seed 0 is the base plan)
      opfx_plan* q = nullptr;
      if (plan_build(c, k, dbg, &q) != OPFX_OK) continue;
      const double cost = plan_cost(q);
      if (cost < best_cost) { delete best; best = q; best_cost = cost; } else delete q;
    }
  }
  *out = best;
  return OPFX_OK;
}

extern "C" int opfx_plan_create(const opfx_case* c, opfx_plan** out) { return opfx_plan_create_debug(c, nullptr, out); }

extern "C" void opfx_plan_destroy(opfx_plan* p) { delete p; }

extern "C" int opfx_plan_get_info(const opfx_plan* p, opfx_plan_info* o) {
  if (!p || !o) { opfx_set_error("opfx_plan_get_info: null argument"); return OPFX_ERR_INVALID; }
  opfx_plan_info* const caller = o;
  const uint32_t caller_size = caller->struct_size;
  if (caller_size < sizeof(uint32_t) || caller_size > sizeof(opfx_plan_info)) {
    opfx_set_error("opfx_plan_get_info: set opfx_plan_info.struct_size = sizeof(opfx_plan_info) before the call (got " + std::to_string(caller_size) + ")");
    return OPFX_ERR_INVALID;
  }
  opfx_plan_info full{};
  o = &full;
  o->nb = p->nb; o->nbr = p->nbr; o->nref = p->nref; o->npv = p->npv; o->npq = p->npq;
  o->nnz_y = (int32_t)p->y_col.size();
  o->nnz_j = p->nnz_j;
  o->n_blk = p->n_blk;
  o->n_fill = (int32_t)p->fill_blk.size();
  o->n_full = p->n_full;
  o->n_levels = p->n_levels();
  o->n_targets = (int32_t)p->tgt_blk.size();
  o->n_sources = (int32_t)p->src_ik.size();
  o->n_uterms = (int32_t)p->u_blk.size();
  o->max_level_width = p->max_level_width;
  {
    // solver state of the lane-programme kernels with two-value storage of the plain blocks
    // (vr, vi, rhs, rq + block values; scheduled P/Q live in global memory)
    const int32_t nbe = (p->nb + 1) & ~1, bs = (p->n_blk + 1) & ~1, nfs = (p->n_full + 1) & ~1;
    o->lds_doubles = 4 * nbe + 2 * bs + 2 * nfs;
  }
  o->lp_rounds_a = p->ra; o->lp_rounds_h = p->rh; o->lp_rounds_b = p->rb; o->lp_rounds_c = p->rc;
  for (int t = 0; t < 2; ++t) { o->team_rounds[t] = p->team_rounds[t]; o->team_barriers[t] = p->team_barriers[t]; }
  o->n_groups = p->n_groups;
  o->team_kb[0] = p->team_kb[0]; o->team_kb[1] = p->team_kb[1];
  o->tail_m = p->tail_m;
  o->lp_ell_width = opfx_plan::KA;
  o->has_dc = p->lp_dc.empty() ? 0 : 1;
  o->lp_rounds_f = p->rf;
  for (int t = 0; t < 2; ++t) { o->team_rounds_chord[t] = p->team_rounds_c[t]; o->team_barriers_chord[t] = p->team_barriers_c[t]; o->team_kb_chord[t] = p->team_kb_c[t]; }
  o->lp_rounds_f_pad = p->rf_pad;
  o->n_shared = p->n_shared;
  full.struct_size = caller_size;
  std::memcpy(caller, &full, caller_size);          // (a caller built against an older, shorter layout gets its prefix)
  return OPFX_OK;
}

extern "C" int64_t opfx_plan_get_array(const opfx_plan* p, int which, int32_t* out, int64_t cap) {
  if (!p) { opfx_set_error("opfx_plan_get_array: null plan"); return OPFX_ERR_INVALID; }
  const std::vector<int32_t>* v = nullptr;
  auto u32 = [](const std::vector<uint32_t>& x) { return reinterpret_cast<const std::vector<int32_t>*>(&x); };
  switch (which) {
    case OPFX_ARR_LP_A_ENT: v = u32(p->lp_a_ent); break;
    case OPFX_ARR_LP_A_DBLK: v = u32(p->lp_a_dblk); break;
    case OPFX_ARR_LP_H_ENT: v = u32(p->lp_h_ent); break;
    case OPFX_ARR_LP_H_ROW: v = u32(p->lp_h_row); break;
    case OPFX_ARR_LP_B: v = u32(p->lp_b); break;
    case OPFX_ARR_LP_B2: v = u32(p->lp_b2); break;
    case OPFX_ARR_LP_B3: v = u32(p->lp_b3); break;
    case OPFX_ARR_LP_C: v = u32(p->lp_c); break;
    case OPFX_ARR_LP_TEAM2: v = u32(p->lp_team[0]); break;
    case OPFX_ARR_LP_TEAM4: v = u32(p->lp_team[1]); break;
    case OPFX_ARR_LP_BCC: v = u32(p->lp_bcc); break;
    case OPFX_ARR_LP_TEAMC2: v = u32(p->lp_teamc[0]); break;
    case OPFX_ARR_LP_TEAMC4: v = u32(p->lp_teamc[1]); break;
    case OPFX_ARR_TAIL_BUS: v = u32(p->tail_bus); break;
    case OPFX_ARR_TAIL_IDS: v = &p->tail_ids32; break;
    case OPFX_ARR_BR_ISLAND: v = &p->br_island; break;
    case OPFX_ARR_ISL_PTR: v = &p->isl_ptr; break;
    case OPFX_ARR_ISL_BUS: v = &p->isl_bus; break;
    case OPFX_ARR_Y_PTR: v = &p->y_ptr; break;
    case OPFX_ARR_Y_COL: v = &p->y_col; break;
    case OPFX_ARR_Y_BLK: v = &p->y_blk; break;
    case OPFX_ARR_DIAG_BLK: v = &p->diag_blk; break;
    case OPFX_ARR_FILL_BLK: v = &p->fill_blk; break;
    case OPFX_ARR_LEV_TPTR: v = &p->lev_tptr; break;
    case OPFX_ARR_TGT_BLK: v = &p->tgt_blk; break;
    case OPFX_ARR_TGT_SPTR: v = &p->tgt_sptr; break;
    case OPFX_ARR_SRC_IK: v = &p->src_ik; break;
    case OPFX_ARR_SRC_KK: v = &p->src_kk; break;
    case OPFX_ARR_SRC_KJ: v = &p->src_kj; break;
    case OPFX_ARR_LEV_PPTR: v = &p->lev_pptr; break;
    case OPFX_ARR_PIV_BUS: v = &p->piv_bus; break;
    case OPFX_ARR_PIV_UPTR: v = &p->piv_uptr; break;
    case OPFX_ARR_U_BLK: v = &p->u_blk; break;
    case OPFX_ARR_U_COL: v = &p->u_col; break;
    case OPFX_ARR_BLK_ROW: v = &p->blk_row; break;
    case OPFX_ARR_BLK_COL: v = &p->blk_col; break;
    case OPFX_ARR_ZERO_LEV: v = &p->zero_lev; break;
    case OPFX_ARR_ZERO_BLK: v = &p->zero_id; break;
    default: opfx_set_error("opfx_plan_get_array: unknown array id"); return OPFX_ERR_INVALID;
  }
  const int64_t n_copy = std::min<int64_t>(cap, (int64_t)v->size());
  if (out && n_copy > 0) std::memcpy(out, v->data(), sizeof(int32_t) * (size_t)n_copy);      // (an empty vector's data() may be null)
  return (int64_t)v->size();
}

extern "C" int opfx_plan_get_ybus(const opfx_plan* p, double* out_g, double* out_b) {
  if (!p) { opfx_set_error("opfx_plan_get_ybus: null plan"); return OPFX_ERR_INVALID; }
  if (out_g) std::memcpy(out_g, p->y_g.data(), sizeof(double) * p->y_g.size());
  if (out_b) std::memcpy(out_b, p->y_b.data(), sizeof(double) * p->y_b.size());
  return OPFX_OK;
}

extern "C" int64_t opfx_plan_get_darray(const opfx_plan* p, int which, double* out, int64_t cap) {
  if (!p) { opfx_set_error("opfx_plan_get_darray: null plan"); return OPFX_ERR_INVALID; }
  const std::vector<double>* v = nullptr;
  switch (which) {
    case OPFX_DARR_LP_A_Y: v = &p->lp_a_y; break;
    case OPFX_DARR_LP_A_YDIAG: v = &p->lp_a_ydiag; break;
    case OPFX_DARR_LP_H_Y: v = &p->lp_h_y; break;
    case OPFX_DARR_LP_DC: v = &p->lp_dc; break;
    case OPFX_DARR_LP_H_DC: v = &p->lp_hdc; break;
    default: opfx_set_error("opfx_plan_get_darray: unknown array id"); return OPFX_ERR_INVALID;
  }
  const int64_t n_copy = std::min<int64_t>(cap, (int64_t)v->size());
  if (out && n_copy > 0) std::memcpy(out, v->data(), sizeof(double) * (size_t)n_copy);
  return (int64_t)v->size();
}
