// STRICT arithmetic build of the render kernel for scenes with leaves below groups or CSG objects: rays are taken through the
// levels above such a leaf one after the other, as the reference takes them (src/shape.rs:39-42 at every level), instead of
// through the product of the levels.  Compile with -ffp-contract=off.  Scenes without such leaves run the kernels of
// rtamd_kernels_strict.hip, which carry none of that code.
#define RTAMD_STRICT 1
#define RTAMD_LEVELS 1
#define RTAMD_KERNEL_NAME rtamd_render_levels
#define RTAMD_MODE_SUFFIX levels
#include "rtamd_kernel.inc"
