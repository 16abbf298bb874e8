// Kernel translation unit: the step kernels outside the plain / DC families (opfx_kernels.h) — chord steps
// (k_step<V2, NW, false, false, CHORD = true>), memory-resident blocks (k_step<1, 4, false, MEM = true>), and the
// first-generation kernel k_step<0, 1> that serves plans beyond the 16-bit descriptors.
#include "opfx_dev.h"
#include "opfx_kernels.h"

const void* opfx_k_step_other(int kind, int v2, int team) {
  if (kind == OPFX_K_MEM) return kernel_handle(k_step<1, 4, false, true>);
  if (kind == OPFX_K_V1) return kernel_handle(k_step<0, 1>);
  if (kind != OPFX_K_CHORD) return nullptr;
  if (v2 == 2) return team == 4 ? kernel_handle(k_step<2, 4, false, false, true>) : team == 2 ? kernel_handle(k_step<2, 2, false, false, true>)
                    : team == 1 ? kernel_handle(k_step<2, 1, false, false, true>) : nullptr;
  if (v2 == 1) return team == 4 ? kernel_handle(k_step<1, 4, false, false, true>) : team == 2 ? kernel_handle(k_step<1, 2, false, false, true>)
                    : team == 1 ? kernel_handle(k_step<1, 1, false, false, true>) : nullptr;
  return nullptr;
}
