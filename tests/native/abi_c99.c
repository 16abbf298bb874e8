/* The boundary used from plain C99 exactly as a foreign-function binding would use it (no C++, no Python): OPFX_INIT'ed
 * structs, a two-bus case through opfx_plan_create / opfx_plan_get_info / opfx_plan_get_array, the versioning rules
 * (include/opfx.h, VERSIONING) and the developer entry point of include/opfx_debug.h.  Host side only: no GPU call. */
#include <stdio.h>
#include <string.h>

#include "opfx.h"
#include "opfx_debug.h"

#define CHECK(cond) do { if (!(cond)) { printf("FAILED line %d: %s (%s)\n", __LINE__, #cond, opfx_last_error()); return 1; } } while (0)

int main(void) {
  int major = -1, minor = -1, patch = -1;
  opfx_version(&major, &minor, &patch);
  CHECK(major == OPFX_VERSION_MAJOR && minor == OPFX_VERSION_MINOR && patch == OPFX_VERSION_PATCH);

  /* slack bus 0 -- line (r = 0.01, x = 0.1 p.u.) -- PQ bus 1 */
  int32_t bus_type[2] = {OPFX_REF, OPFX_PQ}, f[1] = {0}, t[1] = {1};
  double vm[2] = {1.0, 1.0}, va[2] = {0.0, 0.0}, gs[2] = {0.0, 0.0}, bs[2] = {0.0, 0.0}, kf[1] = {100.0}, kt[1] = {100.0};
  const double den = 0.01 * 0.01 + 0.1 * 0.1, g = 0.01 / den, b = -0.1 / den;
  double y[8];
  y[0] = g; y[1] = b; y[2] = -g; y[3] = -b; y[4] = -g; y[5] = -b; y[6] = g; y[7] = b;
  opfx_case c;
  OPFX_INIT(c);
  CHECK(c.struct_size == sizeof(opfx_case) && c.elim_last == NULL && c.br_bdc == NULL);
  c.nb = 2; c.nbr = 1; c.base_mva = 1.0; c.bus_type = bus_type; c.vm_set = vm; c.va_set = va; c.gs = gs; c.bs = bs;
  c.br_f = f; c.br_t = t; c.br_y = y; c.br_kf = kf; c.br_kt = kt;

  opfx_plan* plan = NULL;
  CHECK(opfx_plan_create(&c, &plan) == OPFX_OK && plan != NULL);
  opfx_plan_info info;
  OPFX_INIT(info);
  CHECK(opfx_plan_get_info(plan, &info) == OPFX_OK);
  CHECK(info.nb == 2 && info.nbr == 1 && info.nref == 1 && info.npq == 1 && info.npv == 0 && info.n_blk == 1 && info.has_dc == 0);
  int32_t piv[4];
  CHECK(opfx_plan_get_array(plan, OPFX_ARR_PIV_BUS, piv, 4) == 1 && piv[0] == 1);
  double yg[4], yb[4];
  CHECK(opfx_plan_get_ybus(plan, yg, yb) == OPFX_OK && info.nnz_y == 4);
  opfx_plan_destroy(plan);

  /* a caller built against another header: refused with text, nothing is read past the struct */
  c.struct_size = (uint32_t)sizeof(opfx_case) - 8u;
  plan = NULL;
  CHECK(opfx_plan_create(&c, &plan) == OPFX_ERR_INVALID && plan == NULL && strstr(opfx_last_error(), "struct_size") != NULL);
  c.struct_size = (uint32_t)sizeof(opfx_case);
  info.struct_size = 0;
  CHECK(opfx_plan_create(&c, &plan) == OPFX_OK);
  CHECK(opfx_plan_get_info(plan, &info) == OPFX_ERR_INVALID);
  opfx_plan_destroy(plan);

  /* developer switches travel in an explicit struct; a held-back bus ends the elimination order */
  opfx_debug_opts dbg;
  OPFX_INIT(dbg);
  dbg.plan_search = -1;
  plan = NULL;
  CHECK(opfx_plan_create_debug(&c, &dbg, &plan) == OPFX_OK);
  opfx_plan_destroy(plan);
  CHECK(opfx_plan_create(NULL, &plan) == OPFX_ERR_INVALID);

  /* no GPU here (or one: then the context works) — either way a status, never a crash */
  plan = NULL;
  CHECK(opfx_plan_create(&c, &plan) == OPFX_OK);
  opfx_ctx* ctx = NULL;
  const int rc = opfx_ctx_create(plan, 0, &ctx);
  CHECK(rc == OPFX_OK || rc == OPFX_ERR_NO_DEVICE || rc == OPFX_ERR_HIP);
  if (rc == OPFX_OK) opfx_ctx_destroy(ctx);
  opfx_plan_destroy(plan);
  printf("abi ok: libopfx %d.%d.%d from C99\n", major, minor, patch);
  return 0;
}
