// The kernels of rtamd_kernels_teams.hip once more, built for three waves per SIMD (168 registers per lane: nothing is spilled to
// memory).  Launched when the frame is so small that even at that occupancy every piece of work gets a wave at once --
// how many waves could be resident then makes no difference, how long one wave's iteration takes does.  Same bits.
// Compile with -ffp-contract=fast.
#define RTAMD_STRICT 0
#define RTAMD_TEAMS 1
#define RTAMD_WAVES 3
#define RTAMD_KERNEL_NAME rtamd_render_teams3
#define RTAMD_MODE_SUFFIX teams3
#include "rtamd_kernel.inc"
