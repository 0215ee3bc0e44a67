/*
 * ref_prelude.h — forced include (-include) for every /root/reference source compiled into
 * oracle/_ref.  TEST INFRASTRUCTURE ONLY.
 *
 * The reference's build generates config.h with cmake (platform/x86/config.h.in); we do not run
 * that build system and do not write a config.h of our own.  The tree also carries a real,
 * checked-in config.h (platform/arm/config.h, generated for an ARM/NEON target).  It is used
 * as is, and the ARM-only switches are turned off HERE so that the portable C templates — the
 * code the survey's oracle is defined by — are what gets compiled on this x86 host:
 * with ARCH_ARM 0 the `if (ARCH_ARM) ff_hevcdsp_init_arm(...)` hooks (hevcdsp.c:1327,
 * hevcpred.c:84) are dead code and the tables keep the C slots.
 */
#include "config.h"            /* -I/root/reference/platform/arm */
#undef  ARCH_ARM
#define ARCH_ARM 0
#undef  HAVE_NEON
#define HAVE_NEON 0
#undef  HAVE_ARMV6
#define HAVE_ARMV6 0
#undef  HAVE_ARMV6T2
#define HAVE_ARMV6T2 0
#undef  HAVE_VFP
#define HAVE_VFP 0
#undef  HAVE_VFPV3
#define HAVE_VFPV3 0
#undef  HAVE_INLINE_ASM
#define HAVE_INLINE_ASM 0
#undef  HAVE_NEON_INLINE
#define HAVE_NEON_INLINE 0
#undef  HAVE_ARMV6T2_INLINE
#define HAVE_ARMV6T2_INLINE 0

/* platform/arm/config.h:174 still carries the placeholder of the template it was generated from (`#define HAVE_GMTIME_R
 * @GMTIME_R_FOUND@`); libavutil/parseutils.c tests the macro in an #if.  glibc has gmtime_r: say so. */
#undef  HAVE_GMTIME_R
#define HAVE_GMTIME_R 1

/* the SSE4 variant (oracle/Makefile `refdec_sse`): only hevcdsp.c, hevcpred.c, hevc_filter.c and the x86/ intrinsics files are
 * compiled with -DOH_REF_X86, which turns on the `if (ARCH_X86) ff_hevcdsp_init_x86(...)` hooks (hevcdsp.c:1326, hevcpred.c:84)
 * and the HAVE_SSE* blocks; every other file keeps the portable configuration above.  No assembler: *_EXTERNAL only gates the
 * EXTERNAL_SSE2(flags)-style tests of x86/hevcdsp_init.c (libavutil/x86/cpu.h), see ref_x86dsp_unit.c for the yasm symbols. */
#ifdef OH_REF_X86
#undef  ARCH_X86
#define ARCH_X86 1
#undef  ARCH_X86_64
#define ARCH_X86_64 1
#undef  ARCH_X86_32
#define ARCH_X86_32 0
#undef  HAVE_MMX
#define HAVE_MMX 1
#undef  HAVE_MMX_EXTERNAL
#define HAVE_MMX_EXTERNAL 1
#undef  HAVE_MMXEXT
#define HAVE_MMXEXT 1
#undef  HAVE_MMXEXT_EXTERNAL
#define HAVE_MMXEXT_EXTERNAL 1
#undef  HAVE_SSE
#define HAVE_SSE 1
#undef  HAVE_SSE2
#define HAVE_SSE2 1
#undef  HAVE_SSE2_EXTERNAL
#define HAVE_SSE2_EXTERNAL 1
#undef  HAVE_SSE3
#define HAVE_SSE3 1
#undef  HAVE_SSSE3
#define HAVE_SSSE3 1
#undef  HAVE_SSSE3_EXTERNAL
#define HAVE_SSSE3_EXTERNAL 1
#undef  HAVE_SSE4
#define HAVE_SSE4 1
#undef  HAVE_SSE4_EXTERNAL
#define HAVE_SSE4_EXTERNAL 1
#undef  HAVE_SSE42
#define HAVE_SSE42 1
#undef  HAVE_SSE42_EXTERNAL
#define HAVE_SSE42_EXTERNAL 1
#endif

/* hevc_filter.c reports row progress to frame threads (hevc_filter.c:1040-1050); the harness
 * runs with threads_type == 0, so these are never called.  Declaring the references weak lets
 * the shared object load without pthread_frame.c — no replacement definition is provided. */
#pragma weak ff_thread_report_progress
#pragma weak ff_thread_await_progress
