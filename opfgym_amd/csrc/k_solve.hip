// Kernel translation unit: the pure power-flow kernels of opfx_solve, k_solve<V2, NW, DC, MEM, CHORD> (opfx_kernels.h).
#include "opfx_dev.h"
#include "opfx_kernels.h"

namespace {
template <bool DC, bool CHORD>
const void* solve_kernels(int v2, int team) {
  if (v2 == 2) return team == 4 ? kernel_handle(k_solve<2, 4, DC, false, CHORD>) : team == 2 ? kernel_handle(k_solve<2, 2, DC, false, CHORD>)
                    : team == 1 ? kernel_handle(k_solve<2, 1, DC, false, CHORD>) : nullptr;
  if (v2 == 1) return team == 4 ? kernel_handle(k_solve<1, 4, DC, false, CHORD>) : team == 2 ? kernel_handle(k_solve<1, 2, DC, false, CHORD>)
                    : team == 1 ? kernel_handle(k_solve<1, 1, DC, false, CHORD>) : nullptr;
  return nullptr;
}
}  // namespace

const void* opfx_k_solve(int kind, int v2, int team) {
  switch (kind) {
    case OPFX_KS_PLAIN: return solve_kernels<false, false>(v2, team);
    case OPFX_KS_DC: return solve_kernels<true, false>(v2, team);
    case OPFX_KS_CHORD: return solve_kernels<false, true>(v2, team);
    case OPFX_KS_MEM: return kernel_handle(k_solve<1, 4, false, true>);
    case OPFX_KS_V1: return kernel_handle(k_solve<0, 1>);
  }
  return nullptr;
}
