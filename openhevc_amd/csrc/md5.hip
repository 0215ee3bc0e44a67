/*
 * md5.hip — picture hash on the GPU: the MD5 of each plane of finished pictures, as the reference computes it on the host for the
 * decoded-picture-hash SEI check (libavcodec/hevc.c:4623-4638 calc_md5: the rows of the plane, width << pixel_shift bytes each,
 * without the padding of the line pitch, little-endian samples; hevc.c:4146-4162 compares with the SEI's digests).
 *
 * MD5 is a serial chain over 64-byte blocks, so ONE plane cannot be spread over lanes; what the GPU offers is one chain per
 * (picture, plane) — a batch of 32 pictures is 96 chains — and keeping the 25 MB of a 4K picture off PCIe: 48 bytes per picture come
 * back instead.  One workgroup (one wave) per chain: the 64 lanes fetch the next 64 blocks (4 KB, coalesced dwords, rows of the
 * plane stitched together) into LDS while the wave runs the 64 x 64 steps of the current ones out of LDS (every lane computes the
 * same chain: no divergence, lane 0 writes the digest).  About 2.0 k cycles per block (4 dependent VALU instructions per step):
 * ~75 MB/s per chain, 1024 SIMDs.
 */
#include "kernels_common.h"

namespace {

typedef OhMd5Job Md5Job;

__device__ __forceinline__ uint32_t rotl(uint32_t x, int s) { return __builtin_amdgcn_alignbit(x, x, 32 - s); }

/* word w (4-byte little-endian) of the padded message: data, 0x80 terminator, zeros, bit length */
__device__ __forceinline__ uint32_t msg_word(const Md5Job &j, uint64_t total, uint64_t n_words, uint64_t w)
{
    const uint64_t o = w * 4;
    if (o + 4 <= total) {
        if ((j.row_bytes & 3) == 0) {
            const uint32_t row = (uint32_t)(o / j.row_bytes), col = (uint32_t)(o - (uint64_t)row * j.row_bytes);
            return *(const GLOBAL uint32_t *)((const GLOBAL uint8_t *)j.base + (size_t)row * j.pitch + col);
        }
        uint32_t v = 0;
        for (int k = 0; k < 4; k++) {
            const uint64_t ob = o + k;
            const uint32_t row = (uint32_t)(ob / j.row_bytes), col = (uint32_t)(ob - (uint64_t)row * j.row_bytes);
            v |= (uint32_t)((const GLOBAL uint8_t *)j.base)[(size_t)row * j.pitch + col] << (8 * k);
        }
        return v;
    }
    uint32_t v = 0;
    for (int k = 0; k < 4; k++) {                             /* the tail: last data bytes (row_bytes not a multiple of 4), 0x80 */
        const uint64_t ob = o + k;
        uint32_t b = 0;
        if (ob < total) {
            const uint32_t row = (uint32_t)(ob / j.row_bytes), col = (uint32_t)(ob - (uint64_t)row * j.row_bytes);
            b = ((const GLOBAL uint8_t *)j.base)[(size_t)row * j.pitch + col];
        } else if (ob == total)
            b = 0x80;
        v |= b << (8 * k);
    }
    if (w == n_words - 2) v = (uint32_t)(total << 3);
    if (w == n_words - 1) v = (uint32_t)(total >> 29);
    return v;
}

constexpr uint32_t K[64] = {
    0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af, 0xffff5bb1,
    0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821, 0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453,
    0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a, 0xfffa3942,
    0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70, 0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05,
    0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665, 0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d,
    0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391 };
constexpr int S[16] = { 7, 12, 17, 22, 5, 9, 14, 20, 4, 11, 16, 23, 6, 10, 15, 21 };

__device__ __forceinline__ void md5_block(uint32_t st[4], const uint32_t *m)
{
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3];
#pragma unroll
    for (int i = 0; i < 64; i++) {
        uint32_t f;
        int g;
        if (i < 16)      { f = d ^ (b & (c ^ d)); g = i; }
        else if (i < 32) { f = c ^ (d & (b ^ c)); g = (5 * i + 1) & 15; }
        else if (i < 48) { f = b ^ c ^ d;         g = (3 * i + 5) & 15; }
        else             { f = c ^ (b | ~d);      g = (7 * i) & 15; }
        const uint32_t t = a + f + (m[g] + K[i]);
        a = d; d = c; c = b;
        b = b + rotl(t, S[(i >> 4) * 4 + (i & 3)]);
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d;
}

constexpr int CHUNK = 64;                                     /* blocks fetched per round: one per lane's 16 dwords */

__global__ __launch_bounds__(64) void md5_kernel(const Md5Job *jobs, uint8_t *digests)
{
    __shared__ uint32_t buf[2][CHUNK * 16];
    const Md5Job j = gload(jobs + blockIdx.x);
    const int lane = threadIdx.x;
    const uint64_t total = (uint64_t)j.row_bytes * j.rows;
    const uint64_t n_blocks = (total + 8) / 64 + 1, n_words = n_blocks * 16;
    uint32_t st[4] = { 0x67452301u, 0xefcdab89u, 0x98badcfeu, 0x10325476u };
    uint32_t nxt[16];

    auto fetch = [&](uint64_t first_block) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const uint64_t w = first_block * 16 + (uint64_t)i * 64 + lane;
            nxt[i] = w < n_words ? msg_word(j, total, n_words, w) : 0;
        }
    };
    fetch(0);
    int cur = 0;
    for (uint64_t b0 = 0; b0 < n_blocks; b0 += CHUNK, cur ^= 1) {
#pragma unroll
        for (int i = 0; i < 16; i++)
            buf[cur][i * 64 + lane] = nxt[i];
        __syncthreads();
        if (b0 + CHUNK < n_blocks)
            fetch(b0 + CHUNK);                                /* in flight while the chain below runs */
        const int nb = (int)(n_blocks - b0 < CHUNK ? n_blocks - b0 : CHUNK);
        for (int k = 0; k < nb; k++)
            md5_block(st, &buf[cur][k * 16]);
    }
    if (lane < 4)
        ((GLOBAL uint32_t *)digests)[blockIdx.x * 4 + lane] = lane == 0 ? st[0] : lane == 1 ? st[1] : lane == 2 ? st[2] : st[3];
}

} // namespace

/* jobs / digests: device memory; n chains */
extern "C" void ohk_md5(const OhMd5Job *jobs, int n, void *digests, hipStream_t st)
{
    if (n > 0)
        hipLaunchKernelGGL(md5_kernel, dim3(n), dim3(64), 0, st, jobs, (uint8_t *)digests);
}
