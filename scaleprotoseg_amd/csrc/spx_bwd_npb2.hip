// Backward pixel-kernel instances for 2-block panels (see spx_bwd_impl.h).
#include "spx_bwd_impl.h"
hipError_t spx_launch_bwd_npb2(const SpxBwdArgs& a, int x_dtype, hipStream_t s) { return spx_launch_bwd_npb<2>(a, x_dtype, s); }

// bf16 elements of one G (or a) scratch: [panel][tile][wave][pb][s2] fragments of 512 elements
size_t spx_bwd_scratch_elems(const spx_plan& pl, int B, int HW) {
    const size_t tiles = (size_t)B * ((HW + SPX_TILE_PX - 1) / SPX_TILE_PX);
    // 16-bit elements: the fragment blobs + one float per (lane, block) of inverse scales (= blobs / 8) + 16 bytes for the
    // activation blob's format word
    const size_t blobs = (size_t)pl.npanels * tiles * 4 * pl.npb * 2 * 512;
    return blobs + blobs / 8 + 8 + 2 * (size_t)pl.npanels * tiles + 8;       // + the G blob's exponent per (panel, tile)
}
