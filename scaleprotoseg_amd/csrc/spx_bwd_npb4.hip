// Backward pixel-kernel instances for 4-block panels (see spx_bwd_impl.h).
#include "spx_bwd_impl.h"
hipError_t spx_launch_bwd_npb4(const SpxBwdArgs& a, int x_dtype, hipStream_t s) { return spx_launch_bwd_npb<4>(a, x_dtype, s); }
