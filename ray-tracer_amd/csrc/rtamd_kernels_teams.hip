// FAST arithmetic build of the render kernel with teams of lanes per pixel: compile with -ffp-contract=fast.
// Launched instead of the kernels of rtamd_kernels_fast.hip for frames (or parts of frames) so small that every group
// of pixels gets a wave at once; the pixels are the same bits.
#define RTAMD_STRICT 0
#define RTAMD_TEAMS 1
#define RTAMD_KERNEL_NAME rtamd_render_teams
#define RTAMD_MODE_SUFFIX teams
#include "rtamd_kernel.inc"
