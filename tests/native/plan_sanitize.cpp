// Host-only sanitizer harness for the plan compiler (csrc/plan.cpp): random radial / meshed / complete graphs
// through opfx_plan_create, every read-back array and opfx_plan_destroy under AddressSanitizer + UBSan
// (GPU sanitizers are not available on the pool; the plan compiler is plain host C++).
//   g++ -std=c++17 -g -O1 -fsanitize=address,undefined -fno-omit-frame-pointer -Iinclude -Iopfgym_amd/csrc \
//       tests/native/plan_sanitize.cpp opfgym_amd/csrc/plan.cpp -o /tmp/plan_sanitize && /tmp/plan_sanitize
#include <cstdio>
#include <cstdlib>
#include <random>
#include <set>
#include <utility>
#include <vector>

#include "opfx.h"
#include "opfx_debug.h"

static int run(int nb, int extra, unsigned seed, bool complete) {
  std::mt19937 rng(seed);
  std::uniform_real_distribution<double> u(0.0, 1.0);
  std::vector<int32_t> bt(nb, OPFX_PQ), f, t;
  bt[0] = OPFX_REF;
  if (nb > 6) { bt[3] = OPFX_PV; bt[nb - 2] = OPFX_PV; }
  if (nb > 40 && seed % 3 == 0) bt[nb / 2] = OPFX_REF;
  std::set<std::pair<int, int>> seen;
  auto add = [&](int a, int b) { if (a == b) return; if (a > b) std::swap(a, b); if (seen.insert({a, b}).second) { f.push_back(a); t.push_back(b); } };
  if (complete) { for (int a = 0; a < nb; ++a) for (int b = a + 1; b < nb; ++b) add(a, b); }
  else {
    for (int i = 1; i < nb; ++i) add(i, (int)(u(rng) * i));                 // random tree
    for (int k = 0; k < extra; ++k) add((int)(u(rng) * nb), (int)(u(rng) * nb));   // loops
    if (nb > 4) { f.push_back(f[0]); t.push_back(t[0]); }                    // a parallel branch
  }
  const int nbr = (int)f.size();
  std::vector<double> vm(nb, 1.0), va(nb, 0.0), gs(nb, 0.0), bs(nb, 0.0), y(8 * (size_t)nbr), kf(nbr, 1.0), kt(nbr, 1.0);
  for (int k = 0; k < nbr; ++k) {
    const double g = 1.0 + u(rng), b = -(4.0 + 4.0 * u(rng));
    const double s[8] = {g, b, -g, -b, -g, -b, g, b};
    for (int q = 0; q < 8; ++q) y[8 * (size_t)k + q] = s[q];
  }
  opfx_case c;
  OPFX_INIT(c);                                   // (zeroed, struct_size stamped: include/opfx.h VERSIONING)
  c.nb = nb; c.nbr = nbr; c.base_mva = 100.0; c.bus_type = bt.data(); c.vm_set = vm.data(); c.va_set = va.data();
  c.gs = gs.data(); c.bs = bs.data(); c.br_f = f.data(); c.br_t = t.data(); c.br_y = y.data(); c.br_kf = kf.data(); c.br_kt = kt.data();
  std::vector<int32_t> held(nb, 0);               // every third graph: a few buses held back to the end (opfx_case.elim_last)
  if (seed % 3 == 0) { for (int i = 1; i < nb; i += 7) held[i] = 1; c.elim_last = held.data(); }
  opfx_plan* p = nullptr;
  // (the plan search of grids with 200-800 buses builds 16 plans: three of them here, through the developer entry point,
  //  so that the sanitizer run stays a matter of seconds)
  opfx_debug_opts dbg;
  OPFX_INIT(dbg);
  dbg.plan_search = 3;
  const int rc = nb >= 200 ? opfx_plan_create_debug(&c, &dbg, &p) : opfx_plan_create(&c, &p);
  if (rc != OPFX_OK) { std::printf("nb=%d: plan_create -> %d (%s)\n", nb, rc, opfx_last_error()); return rc == OPFX_ERR_TOO_LARGE ? 0 : 1; }
  opfx_plan_info info;
  OPFX_INIT(info);
  if (opfx_plan_get_info(p, &info) != OPFX_OK) { std::printf("plan_get_info: %s\n", opfx_last_error()); return 1; }
  long long total = 0;
  for (int which = 0; which <= OPFX_ARR_LP_TEAMC4; ++which) {
    const int64_t n = opfx_plan_get_array(p, which, nullptr, 0);
    if (n < 0) { std::printf("array %d: %lld\n", which, (long long)n); return 1; }
    std::vector<int32_t> buf((size_t)n + 1);
    opfx_plan_get_array(p, which, buf.data(), n);
    total += n;
  }
  std::vector<double> yg(info.nnz_y), yb(info.nnz_y);
  opfx_plan_get_ybus(p, yg.data(), yb.data());
  std::printf("nb=%4d nbr=%5d levels=%3d blocks=%5d tail=%2d team rounds %d/%d  arrays %lld ints\n", nb, nbr, info.n_levels,
              info.n_blk, info.tail_m, info.team_rounds[0], info.team_rounds[1], total);
  opfx_plan_destroy(p);
  return 0;
}

int main() {
  int bad = 0;
  for (unsigned s = 0; s < 24; ++s) bad += run(5 + (int)(s * 37 % 400), (int)(s * 11 % 90), s, false);
  for (int n : {4, 9, 26, 33, 34, 41, 60}) bad += run(n, 0, 7, true);
  bad += run(2, 0, 1, false);
  std::printf(bad ? "FAILED\n" : "plan compiler: clean under ASan/UBSan\n");
  return bad ? 1 : 0;
}
