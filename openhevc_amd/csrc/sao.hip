/*
 * sao.hip — pass 5: sample adaptive offset (band / edge), cur -> out
 * (gfx950; overview of the passes: kernels.hip; bit-exactness: tests/test_gpu_parity.py)
 */
#include "kernels_common.h"

/* =========================================================================================
 * pass 5: SAO — hevcdsp_template.c:340-567 driven per CTB by sao_filter_CTB (hevc_filter.c:197-322),
 * here one whole-picture pass from the deblocked planes (cur) into the output planes (out).
 * One lane owns 8 consecutive samples of a row (8- or 16-byte accesses); a group never straddles a
 * CTB (CTB widths are multiples of 8 samples in every plane).  Rows are 256-byte aligned and padded,
 * so whole-vector accesses past the picture width stay inside the row.
 * ======================================================================================= */
template <typename PX> struct Vec8;
template <> struct Vec8<uint8_t>  { typedef unsigned int  T __attribute__((ext_vector_type(2))); };
template <> struct Vec8<uint16_t> { typedef unsigned int  T __attribute__((ext_vector_type(4))); };

template <typename PX>
static __device__ __forceinline__ void load8(const GLOBAL PX *p, int v[8])
{
    typename Vec8<PX>::T r = *(const GLOBAL typename Vec8<PX>::T *)p;
    if (sizeof(PX) == 1) {
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = (r[j >> 2] >> (8 * (j & 3))) & 0xff;
    } else {
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = (r[j >> 1] >> (16 * (j & 1))) & 0xffff;
    }
}
template <typename PX>
static __device__ __forceinline__ void store8(GLOBAL PX *p, const int v[8])
{
    typename Vec8<PX>::T r;
    if (sizeof(PX) == 1) {
        r[0] = v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24);
        r[1] = v[4] | (v[5] << 8) | (v[6] << 16) | (v[7] << 24);
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++) r[j] = v[2 * j] | (v[2 * j + 1] << 16);
    }
    __builtin_nontemporal_store(r, (GLOBAL typename Vec8<PX>::T *)p);     /* the output is next read by another picture's MC: stream it past the L2 */
}

struct SaoEdgeCtx { int x, y, x0, y0, w, h, pw, ph, sstride, cx, cy, ctbw, ctbh, flags, bd; const GLOBAL uint16_t *stale; int stale_r, vs; };

/* 16x16 CTBs with subsampled chroma (DevFrame.sao_stale): sample (x0 + 8, yr) — first column of the right neighbour — as
 * sao_filter_CTB of this CTB saw it: rows touched by a horizontal chroma edge of a pending CTB row (stale_r bit 0: row cy, bit 1: row cy + 1) were not filtered yet */
static __device__ __forceinline__ int sao_stale_or(const SaoEdgeCtx &e, const OhPicParams &pp, int c, int yr, int v)
{
    const int p0 = (yr & 7) == 7, ye = p0 ? yr + 1 : yr;
    if ((ye & 7) == 0 && ye > 0 && ye < e.ph && ((e.stale_r >> ((((ye << e.vs) >> 4) > e.cy) ? 1 : 0)) & 1))
        return e.stale[oh_sao_stale_index(&pp, c, ye >> 3, e.cx + 1) + (p0 ? 0 : 1)];
    return v;
}

/* first neighbour a = (x+DX, y+DY), second b = (x-DX, y-DY) (pos[][] of hevcdsp_template.c:379-384) */
template <typename PX, int DX, int DY>
static __device__ __forceinline__ void sao_edge8(const GLOBAL PX *__restrict__ src, const SaoEdgeCtx &e, const int off[5], const int v[8], int r[8],
                                                 const OhPicParams &pp, const int c)
{
    int a[10], b[10];                                       /* samples x-1..x+8 of rows y+DY and y-DY */
    const int ya = min(max(e.y + DY, 0), e.ph - 1), yb = min(max(e.y - DY, 0), e.ph - 1);
    load8<PX>(src + (size_t)ya * e.sstride + e.x, a + 1);
    load8<PX>(src + (size_t)yb * e.sstride + e.x, b + 1);
    a[0] = b[0] = a[9] = b[9] = 0;
    if (DX != 0) {
        if (e.x > 0)        { a[0] = src[(size_t)ya * e.sstride + e.x - 1]; b[0] = src[(size_t)yb * e.sstride + e.x - 1]; }
        if (e.x + 8 < e.pw) { a[9] = src[(size_t)ya * e.sstride + e.x + 8]; b[9] = src[(size_t)yb * e.sstride + e.x + 8]; }
        if (e.stale && e.cx + 2 < e.ctbw) { a[9] = sao_stale_or(e, pp, c, ya, a[9]); b[9] = sao_stale_or(e, pp, c, yb, b[9]); }
    }
    const int ly = e.y - e.y0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int lx = e.x + j - e.x0;
        bool keep = e.x + j >= e.pw;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int nx = lx + (k ? -DX : DX), ny = ly + (k ? -DY : DY);
            const int rx = nx < 0 ? -1 : (nx >= e.w ? 1 : 0), ry = ny < 0 ? -1 : (ny >= e.h ? 1 : 0);
            if ((rx < 0 && e.cx == 0) || (rx > 0 && e.cx == e.ctbw - 1) || (ry < 0 && e.cy == 0) || (ry > 0 && e.cy == e.ctbh - 1))
                keep = true;
            else if (e.flags && (rx || ry)) {
                int bit;
                if (rx && ry) bit = 4 + (ry < 0 ? (rx < 0 ? 0 : 1) : (rx > 0 ? 2 : 3));
                else if (rx)  bit = rx > 0 ? 1 : 0;
                else          bit = ry > 0 ? 3 : 2;
                keep = keep || ((e.flags >> bit) & 1);
            }
        }
        if (!keep) {
            const int na = a[1 + j + DX], nb = b[1 + j - DX];
            const int sum = (v[j] > na) - (v[j] < na) + (v[j] > nb) - (v[j] < nb);
            const int o = sum == 0 ? off[0] : (sum == -2 ? off[1] : (sum == -1 ? off[2] : (sum == 1 ? off[3] : off[4])));
            r[j] = clip_px(v[j] + o, e.bd);
        }
    }
}

/* Lane -> sample mapping: a wave covers ONE CTB-wide strip (wc samples x 512/wc rows) so that the
 * SAO type / class is the same for all its lanes (no divergent band/edge paths); a workgroup of
 * 4 waves covers 4 such strips stacked vertically.  grid = (CTB columns, strips of rows of every plane, pictures). */
template <typename PX>
__global__ __launch_bounds__(256) void sao_kernel(const OhBatch B, const int strips_luma, const int strips_chroma)
{
    const DevFrame *__restrict__ f = B.f[blockIdx.z];
    const OhPicParams &pp = f->pp;
    /* grid.y: the luma strips, then the strips of Cb, then those of Cr (a chroma plane of 4:2:0 needs a quarter of the
     * luma plane's workgroups: no workgroup is launched just to exit) */
    int strip = blockIdx.y, c = 0;
    if (strip >= strips_luma) {
        strip -= strips_luma;
        c = strip >= strips_chroma ? 2 : 1;
        strip -= (c - 1) * strips_chroma;
    }
    const int pw = f->cur.w[c], ph = f->cur.h[c];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int log2_gx = pp.log2_ctb_size - hsh(pp, c) - 3;            /* 8-sample groups per CTB row: 1 << log2_gx */
    const int rows_per_wave = 64 >> log2_gx;
    const int x = (blockIdx.x << (log2_gx + 3)) + ((lane & ((1 << log2_gx) - 1)) << 3);
    const int y = (strip * 4 + wave) * rows_per_wave + (lane >> log2_gx);
    if (x >= pw || y >= ph)
        return;
    const int bd = pp.bit_depth, hs = hsh(pp, c), vs = vsh(pp, c), lc = pp.log2_ctb_size;
    const int ctbw = (pp.width + (1 << lc) - 1) >> lc, ctbh = (pp.height + (1 << lc) - 1) >> lc;
    const int sstride = f->cur.stride[c];
    const GLOBAL PX *__restrict__ src = G_CONST(PX, f->cur.p[c]);
    GLOBAL PX *__restrict__ dst = G_MUT(PX, f->out.p[c]) + (size_t)y * f->out.stride[c] + x;
    const int cx = (x << hs) >> lc, cy = (y << vs) >> lc;
    const GLOBAL OhSaoCtb *s = G_CONST(OhSaoCtb, f->sao) + cy * ctbw + cx;
    const int type = s->type_idx[c];
    int v[8], r[8];
    load8<PX>(src + (size_t)y * sstride + x, v);
#pragma unroll
    for (int j = 0; j < 8; j++) r[j] = v[j];
    if (type == 1) {                                        /* band, :340-365 */
        const int bp = s->band_position[c];
        int off[4];
#pragma unroll
        for (int k = 0; k < 4; k++) off[k] = s->offset_val[c][k + 1];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            int k = ((v[j] >> (bd - 5)) - bp) & 31;
            int o = k == 0 ? off[0] : (k == 1 ? off[1] : (k == 2 ? off[2] : off[3]));
            if (k < 4) r[j] = clip_px(v[j] + o, bd);
        }
    } else if (type == 2) {                                 /* edge, :372-567 (per-sample form) */
        const int eo = s->eo_class[c];
        const int x0 = (cx << lc) >> hs, y0 = (cy << lc) >> vs;
        const int w = min((1 << lc) >> hs, pw - x0), h = min((1 << lc) >> vs, ph - y0);
        const int flags = s->edge_flags;
        int off[5];
#pragma unroll
        for (int k = 0; k < 5; k++) off[k] = s->offset_val[c][k];
        /* the chroma CTB is one 8-sample group wide in that configuration: x + 8 is the neighbour's first column */
        const GLOBAL uint16_t *stale = c && pp.deblock_enabled ? G_CONST(uint16_t, f->sao_stale) : nullptr;
        /* rows pending when the reference's driver ran this CTB's SAO: from min(cy + 1, ctbh - 2) on — from min(cy + 1, ctbh - 1) on when
         * the CTB after the neighbour is the last of its row (its deblocking call comes a CTB earlier, hevc_filter.c:1058-1059) */
        const int r_lag = cx + 2 == ctbw - 1 ? ctbh - 1 : (ctbh >= 2 ? ctbh - 2 : 0), r_pending = min(cy + 1, r_lag);
        /* as bits: 0 = the edges of CTB row cy pending, 1 = those of row cy + 1; handed over per CTB when the picture was not decoded
         * in raster order (tiles: DevFrame.sao_pending, OhFrame.sao_pending) */
        const int pend = f->sao_pending ? G_CONST(uint8_t, f->sao_pending)[cy * ctbw + cx] : (cy >= r_pending ? 1 : 0) | (cy + 1 >= r_pending ? 2 : 0);
        const SaoEdgeCtx ec = { x, y, x0, y0, w, h, pw, ph, sstride, cx, cy, ctbw, ctbh, flags, bd, stale, pend, vs };
        switch (eo) {                                       /* compile-time neighbour offsets: no indexed registers */
        case 0:  sao_edge8<PX, -1, 0>(src, ec, off, v, r, pp, c); break;
        case 1:  sao_edge8<PX, 0, -1>(src, ec, off, v, r, pp, c); break;
        case 2:  sao_edge8<PX, -1, -1>(src, ec, off, v, r, pp, c); break;
        default: sao_edge8<PX, 1, -1>(src, ec, off, v, r, pp, c); break;
        }
    }
    if (type && f->is_pcm && (pp.transquant_bypass_enable || pp.pcm_loop_filter_disable)) {
        /* restore_tqb_pixels (hevc_filter.c:163-193) with its geometry quirks: the min-PU range is
         * derived from the CTB's LUMA origin plus the COMPONENT's size, and the row copy length
         * is (min_pu >> hshift) BYTES whatever the sample size. */
        const int l = pp.log2_min_pu_size, mpw = pp.width >> l;
        const int X0 = cx << lc, Y0 = cy << lc;
        const int wc = min((1 << lc) >> hs, pw - (X0 >> hs)), hc = min((1 << lc) >> vs, ph - (Y0 >> vs));
        const int py = (y << vs) >> l;
        const GLOBAL uint8_t *pcm = G_CONST(uint8_t, f->is_pcm);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int px = ((x + j) << hs) >> l;
            if (px >= (X0 >> l) && px < ((X0 + wc) >> l) && py >= (Y0 >> l) && py < ((Y0 + hc) >> l) && pcm[py * mpw + px]) {
                int sx = (px << l) >> hs;
                int len_samples = ((1 << l) >> hs) / (int)sizeof(PX);
                if (x + j - sx < len_samples)
                    r[j] = v[j];
            }
        }
    }
    store8<PX>(dst, r);
}

/* =========================================================================================
 * launcher
 * ======================================================================================= */
extern "C" void ohk_sao(const OhBatch *B, int n, const OhPicParams *p, hipStream_t st)
{
    const int lc = p->log2_ctb_size, ctbw = (p->width + (1 << lc) - 1) >> lc;
    const int hs = p->chroma_format_idc == 1 || p->chroma_format_idc == 2, vs = p->chroma_format_idc == 1;
    const int rows_luma = 4 * (64 >> (lc - 3)), rows_chroma = 4 * (64 >> (lc - hs - 3));     /* rows per workgroup: 4 waves x (512 / CTB width) */
    const int sl = (p->height + rows_luma - 1) / rows_luma;
    const int sc = p->chroma_format_idc ? ((p->height >> vs) + rows_chroma - 1) / rows_chroma : 0;
    dim3 grid(ctbw, sl + 2 * sc, n);
    if (p->bit_depth == 8) hipLaunchKernelGGL(HIP_KERNEL_NAME(sao_kernel<uint8_t>), grid, dim3(256), 0, st, *B, sl, sc);
    else                   hipLaunchKernelGGL(HIP_KERNEL_NAME(sao_kernel<uint16_t>), grid, dim3(256), 0, st, *B, sl, sc);
}
