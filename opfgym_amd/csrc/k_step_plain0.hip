// Kernel translation unit: the plain Newton step kernels k_step<V2, NW, DC = false, ..., SPEC = 0, MINW> (opfx_kernels.h).
#include "opfx_dev.h"
#include "opfx_kernels.h"

const void* opfx_k_step_plain0(int v2, int team, int minw) { return step_kernels<false, 0>(v2, team, minw); }
