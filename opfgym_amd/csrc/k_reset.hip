// Kernel translation unit: the reset kernels k_reset<rows per workgroup, FULL> (opfx_kernels.h).
#include "opfx_dev.h"
#include "opfx_kernels.h"

const void* opfx_k_reset(int rows, int full) {
  if (full) return rows == 4 ? kernel_handle(k_reset<4, true>) : rows == 2 ? kernel_handle(k_reset<2, true>) : rows == 1 ? kernel_handle(k_reset<1, true>) : nullptr;
  return rows == 4 ? kernel_handle(k_reset<4, false>) : rows == 2 ? kernel_handle(k_reset<2, false>) : rows == 1 ? kernel_handle(k_reset<1, false>) : nullptr;
}
