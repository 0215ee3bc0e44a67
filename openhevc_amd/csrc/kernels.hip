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
 *   pass 1  mc.hip        mc_kernel         one wave per <=16x16 luma tile (+ its chroma), LDS window + 2-stage filter
 *   pass 2  residual.hip  residual_kernel   one wave per transform block, two LDS matrix passes
 *   pass 3  intra.hip     intra_ctu_kernel  one workgroup per CTU of one wavefront level, waves take the blocks of a sub-level
 *   pass 4  deblock.hip   deblock_*_kernel  one lane per 4-line edge segment, V pass then H pass, in place
 *   pass 5  sao.hip       sao_kernel        one lane per sample, cur -> out
 *   SHVC    upsample.hip  upsample_*_kernel one lane per output sample, two launches per plane
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
