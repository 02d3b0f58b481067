// FAST arithmetic build of the render kernel: compile with -ffp-contract=fast.
#define RTAMD_STRICT 0
#define RTAMD_KERNEL_NAME rtamd_render_fast
#define RTAMD_MODE_SUFFIX fast
#include "rtamd_kernel.inc"
