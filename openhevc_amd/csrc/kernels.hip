/*
 * kernels.hip — CDNA4 (gfx950) kernels of the block-reconstruction passes: overview and one-time set-up.
 * One file per pass (device symbols stay file-local: no relocatable device code).
 *
 * Integer/byte work throughout (no MFMA): the passes are bounded by HBM/L2 traffic and by
 * per-block latency, so the design rules are wave64-sized work units, LDS staging of the
 * separable-filter windows and coefficient blocks, and coalesced row accesses.
 * Arithmetic follows the reference exactly (file:line cited per kernel, paths relative to
 * /root/reference/libavcodec/); tests/ check every kernel bit-for-bit against the CPU checker.
 *
 *   hand-over prep.hip   prep_*           raw lists -> MC jobs, TU size buckets, intra descriptors, level statistics (one launch set per 32 lists)
 *   pass 1  mc.hip        mc_kernel         a wave runs four <=8x8 blocks of one plane (16 lanes each): windows in LDS, v_dot2 h- and v-pass
 *   pass 2  residual.hip  residual_kernel   one launch per size; sixteen 4x4 / four 8x8 / one 16x16 / one 32x32 block per wave, two LDS matrix passes
 *   pass 3  intra.hip     intra_ctu_kernel  one workgroup per CTU of one wavefront level; <=8x8 blocks four per wave, prepared one sub-level ahead
 *                         intra_rows_kernel one workgroup per CTU row of an I picture, rows two CTUs apart like the reference's WPP threads (one launch)
 *   pass 4  deblock.hip   deblock_*_kernel  one lane per 4-line edge segment, V pass then H pass, in place
 *   pass 5  sao.hip       sao_kernel        eight samples per lane, CTB-uniform waves, half 0 -> half 1
 *   BS      bs.hip        bs_kernel         optional: both boundary-strength grids from the motion field, one lane per 4x4 cell
 *   SHVC    upsample.hip  upsample_tile_kernel  one workgroup per tile (CTB), window and horizontal rows in LDS
 *   output  md5.hip       md5_kernel        one wave per (picture, plane) MD5 chain
 */
#include "kernels_common.h"

/* per-file set-up: the interpolation taps and the transform bases are packed on the host once (g_mctab in
 * mc.hip, g_basis in residual.hip); the intra kernels raise their dynamic-LDS limit */
int ohk_init_mc(void);
int ohk_init_residual(void);
int ohk_init_intra(void);

extern "C" int ohk_init(void)
{
    if (ohk_init_mc() || ohk_init_residual() || ohk_init_intra())
        return -1;
    return 0;
}
