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

/* hevc_filter.c reports row progress to frame threads (hevc_filter.c:1040-1050); the harness
 * runs with threads_type == 0, so these are never called.  Declaring the references weak lets
 * the shared object load without pthread_frame.c — no replacement definition is provided. */
#pragma weak ff_thread_report_progress
#pragma weak ff_thread_await_progress
