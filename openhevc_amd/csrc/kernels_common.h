/*
 * kernels_common.h — what every pass's kernel file shares: address-space helpers, packed sample
 * accesses, clips.  Device code only (included by the .hip files of this directory).
 */
#ifndef OHEVC_KERNELS_COMMON_H
#define OHEVC_KERNELS_COMMON_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dev_frame.h"
#include "kernels.h"

/* Pointers read out of DevFrame are generic to the compiler, which then emits flat_* accesses;
 * those count on lgkmcnt as well as vmcnt, so every LDS wait would also wait for stores in flight.
 * All of them point to HBM: say so. */
#define GLOBAL __attribute__((address_space(1)))
#define G_CONST(T, p) ((const GLOBAL T *)(p))
#define G_MUT(T, p)   ((GLOBAL T *)(p))

/* struct load from HBM (C++ cannot copy-construct from an address-space-qualified lvalue) */
template <typename T>
static __device__ __forceinline__ T gload(const T *p)
{
    static_assert(sizeof(T) % 4 == 0, "dword-sized structs only");
    T out;
    const GLOBAL uint32_t *s = (const GLOBAL uint32_t *)p;
    uint32_t w[sizeof(T) / 4];
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; i++)
        w[i] = s[i];
    __builtin_memcpy(&out, w, sizeof(T));                     /* not through a uint32_t alias of `out`: its members have other types */
    return out;
}

typedef short short4v __attribute__((ext_vector_type(4)));
typedef unsigned int uint2v __attribute__((ext_vector_type(2)));
typedef unsigned int uint4v __attribute__((ext_vector_type(4)));

static __device__ __forceinline__ int clip3(int v, int lo, int hi) { return min(max(v, lo), hi); }
static __device__ __forceinline__ int clip_px(int v, int bd) { return min(max(v, 0), (1 << bd) - 1); }
static __device__ __forceinline__ int clip16(int v) { return min(max(v, -32768), 32767); }
static __device__ __forceinline__ int hsh(const OhPicParams &p, int c) { return c && (p.chroma_format_idc == 1 || p.chroma_format_idc == 2); }
static __device__ __forceinline__ int vsh(const OhPicParams &p, int c) { return c && p.chroma_format_idc == 1; }

/* four consecutive samples as one 4-byte (8 bit) or 8-byte (>8 bit) access */
template <typename PX>
static __device__ __forceinline__ void load4(const GLOBAL PX *p, int v[4])
{
    if (sizeof(PX) == 1) {
        unsigned r = *(const GLOBAL unsigned *)p;
        v[0] = r & 0xff; v[1] = (r >> 8) & 0xff; v[2] = (r >> 16) & 0xff; v[3] = r >> 24;
    } else {
        uint2v r = *(const GLOBAL uint2v *)p;
        v[0] = r[0] & 0xffff; v[1] = r[0] >> 16; v[2] = r[1] & 0xffff; v[3] = r[1] >> 16;
    }
}
template <typename PX>
static __device__ __forceinline__ void store4(GLOBAL PX *p, int a, int b, int c, int d)
{
    if (sizeof(PX) == 1) {
        *(GLOBAL unsigned *)p = (unsigned)(a | (b << 8) | (c << 16) | (d << 24));
    } else {
        uint2v r = { (unsigned)(a | (b << 16)), (unsigned)(c | (d << 16)) };
        *(GLOBAL uint2v *)p = r;
    }
}
typedef uint2v uint2v_a2 __attribute__((aligned(2)));
typedef unsigned unsigned_a1 __attribute__((aligned(1)));
/* four consecutive samples at any sample address -> two dwords of 16-bit pairs */
static __device__ __forceinline__ uint2v load4_pairs(const GLOBAL uint8_t *p)
{
    const unsigned b = *(const GLOBAL unsigned_a1 *)p;
    return uint2v{ __builtin_amdgcn_perm(0, b, 0x0c010c00), __builtin_amdgcn_perm(0, b, 0x0c030c02) };
}
static __device__ __forceinline__ uint2v load4_pairs(const GLOBAL uint16_t *p) { return *(const GLOBAL uint2v_a2 *)p; }

#endif
