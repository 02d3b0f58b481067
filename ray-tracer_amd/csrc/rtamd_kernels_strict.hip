// STRICT arithmetic build of the render kernel: compile with -ffp-contract=off.
#define RTAMD_STRICT 1
#define RTAMD_KERNEL_NAME rtamd_render_strict
#define RTAMD_MODE_SUFFIX strict
#include "rtamd_kernel.inc"
