// Kernel translation unit: the step kernels compiled with the DC start, k_step<V2, NW, DC = true, ..., SPEC = 2, MINW>
// (opfx_kernels.h).
#include "opfx_dev.h"
#include "opfx_kernels.h"

const void* opfx_k_step_dc2(int v2, int team, int minw) { return step_kernels<true, 2>(v2, team, minw); }
