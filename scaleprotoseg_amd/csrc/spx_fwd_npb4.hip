// Forward kernel instances for 4-block panels (see spx_fwd_impl.h).
#include "spx_fwd_impl.h"
hipError_t spx_launch_fwd_npb4(const SpxFwdArgs& a, int x_dtype, hipStream_t s) { return spx_launch_fwd_npb<4>(a, x_dtype, s); }
