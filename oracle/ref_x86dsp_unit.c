/*
 * ref_x86dsp_unit.c — TEST INFRASTRUCTURE ONLY: the reference's x86 table initialiser (libavcodec/x86/hevcdsp_init.c) compiled
 * WITHOUT an assembler.  The tree's SSE4 path is two things: C files of intrinsics (x86/hevc_mc_sse.c, hevc_idct_sse.c,
 * hevc_sao_sse.c, hevc_intra_pred_sse.c, hevc_il_pred_sse.c — gcc compiles them where they lie) and yasm files
 * (x86/hevc_deblock.asm, hevc_idct.asm, hevc_mc.asm); there is no yasm / nasm in this image.  ff_hevcdsp_init_x86
 * (x86/hevcdsp_init.c:424-676) assigns the four deblocking slots from the yasm file.  Here the initialiser is compiled under
 * another name and called from a ff_hevcdsp_init_x86 of our own that saves the four deblocking slots before the call and puts
 * them back after it — so they keep the C template functions hevcdsp.c put there, and everything else the initialiser assigns
 * (MC, IDCT, transform_add, SAO, up-sampling: all intrinsics) is the reference's own.  The yasm entry points are weak
 * references (null, assigned and at once overwritten).  Nothing of the reference is edited or replaced.
 * This is SURVEY.md §6's "SSE4 intrinsics build, deblocking stays C" baseline (bench.py cpu_baseline.sse_*).
 */
#include "libavutil/cpu.h"
#include "libavutil/x86/asm.h"
#include "libavutil/x86/cpu.h"
#include "libavcodec/get_bits.h"
#include "libavcodec/hevcdsp.h"

/* libavutil/cpu.c is compiled with ARCH_X86 0 in this build (its x86 probe needs the assembler too) and reports no flags; ask
 * the compiler's cpuid helper instead */
static int oh_x86_flags(void)
{
    int f = 0;
    __builtin_cpu_init();
    if (__builtin_cpu_supports("mmx"))    f |= AV_CPU_FLAG_MMX | AV_CPU_FLAG_MMXEXT;
    if (__builtin_cpu_supports("sse"))    f |= AV_CPU_FLAG_SSE;
    if (__builtin_cpu_supports("sse2"))   f |= AV_CPU_FLAG_SSE2;
    if (__builtin_cpu_supports("sse3"))   f |= AV_CPU_FLAG_SSE3;
    if (__builtin_cpu_supports("ssse3"))  f |= AV_CPU_FLAG_SSSE3;
    if (__builtin_cpu_supports("sse4.1")) f |= AV_CPU_FLAG_SSE4;
    if (__builtin_cpu_supports("sse4.2")) f |= AV_CPU_FLAG_SSE42;
    return f;
}
#define av_get_cpu_flags() oh_x86_flags()

#pragma weak ff_hevc_v_loop_filter_chroma_8_sse2
#pragma weak ff_hevc_h_loop_filter_chroma_8_sse2
#pragma weak ff_hevc_v_loop_filter_chroma_10_sse2
#pragma weak ff_hevc_h_loop_filter_chroma_10_sse2
#pragma weak ff_hevc_v_loop_filter_chroma_12_sse2
#pragma weak ff_hevc_h_loop_filter_chroma_12_sse2
#pragma weak ff_hevc_v_loop_filter_luma_8_ssse3
#pragma weak ff_hevc_h_loop_filter_luma_8_ssse3
#pragma weak ff_hevc_v_loop_filter_luma_10_ssse3
#pragma weak ff_hevc_h_loop_filter_luma_10_ssse3
#pragma weak ff_hevc_v_loop_filter_luma_12_ssse3
#pragma weak ff_hevc_h_loop_filter_luma_12_ssse3

#define ff_hevcdsp_init_x86 oh_ref_hevcdsp_init_x86
#include "libavcodec/x86/hevcdsp_init.c"
#undef ff_hevcdsp_init_x86

void ff_hevcdsp_init_x86(HEVCDSPContext *c, const int bit_depth)
{
    HEVCDSPContext keep = *c;
    oh_ref_hevcdsp_init_x86(c, bit_depth);
    c->hevc_v_loop_filter_luma   = keep.hevc_v_loop_filter_luma;
    c->hevc_h_loop_filter_luma   = keep.hevc_h_loop_filter_luma;
    c->hevc_v_loop_filter_chroma = keep.hevc_v_loop_filter_chroma;
    c->hevc_h_loop_filter_chroma = keep.hevc_h_loop_filter_chroma;
}
