// Kernel translation units of libopfx and what each hands to the host side (opfx.hip).
//
// The kernels are templates in opfx_dev.h; ~100 instantiations of them exist (block storage x wavefronts per instance x
// DC start / chord steps / memory-resident blocks x SPEC x wavefronts per SIMD).  Compiled in one translation unit they
// took 100-170 s; they are spread over the files below, which __graft_entry__.build() compiles in parallel.  No device
// code crosses a translation unit (no -fgpu-rdc): every file includes opfx_dev.h and instantiates its own share.
//
// Each accessor returns the host-side handle of one instantiation (the pointer hipLaunchKernel / hipFuncSetAttribute /
// the occupancy query take) or nullptr when its arguments name none.  opfx.hip holds WEAK definitions returning nullptr,
// so a developer build may leave translation units out (scripts/ab_dup.sh builds the headline's alone); a launch that
// needs a kernel which is not linked in is refused with an error text, never redirected.
//
//   v2    block storage: 1 four-value, 2 two-value (0 = the first-generation kernel, k_other only)
//   team  wavefronts per instance (1, 2, 4)
//   minw  wavefronts per SIMD the kernel is compiled for (2; 3 = the 168-VGPR instantiations)
#ifndef OPFX_KERNELS_H
#define OPFX_KERNELS_H

const void* opfx_k_step_plain0(int v2, int team, int minw);    // k_step_plain0.hip: plain Newton step kernels, SPEC = 0
const void* opfx_k_step_plain1(int v2, int team, int minw);    // k_step_plain1.hip: SPEC = 1 (no PV bus)
const void* opfx_k_step_plain2(int v2, int team, int minw);    // k_step_plain2.hip: SPEC = 2 (no modifiers)
const void* opfx_k_step_plain3(int v2, int team, int minw);    // k_step_plain3.hip: SPEC = 3
const void* opfx_k_step_dc0(int v2, int team, int minw);       // k_step_dc0.hip: step kernels compiled with the DC start, SPEC = 0
const void* opfx_k_step_dc1(int v2, int team, int minw);       // k_step_dc1.hip
const void* opfx_k_step_dc2(int v2, int team, int minw);       // k_step_dc2.hip
const void* opfx_k_step_dc3(int v2, int team, int minw);       // k_step_dc3.hip
enum { OPFX_K_CHORD = 0, OPFX_K_MEM = 1, OPFX_K_V1 = 2 };
const void* opfx_k_step_other(int kind, int v2, int team);     // k_step_other.hip: chord steps; memory-resident blocks; first generation
enum { OPFX_KS_PLAIN = 0, OPFX_KS_DC = 1, OPFX_KS_CHORD = 2, OPFX_KS_MEM = 3, OPFX_KS_V1 = 4 };
const void* opfx_k_solve(int kind, int v2, int team);          // k_solve.hip: the pure power-flow kernels (opfx_solve)
const void* opfx_k_reset(int rows, int full);                  // k_reset.hip: reset kernels, 1 / 2 / 4 rows per workgroup

#endif  // OPFX_KERNELS_H
