/*
 * tables.c — recording implementations of the reference's HEVCDSPContext / HEVCPredContext /
 * VideoDSPContext slots (include/ohevc_tables.h).  Plain C, host only.
 *
 * State is per decoding thread (the reference gives every slice/frame thread its own
 * HEVCContext copy, hevc.c:3067-3075, but the table slots carry no context argument,
 * SURVEY.md §8b "Threading"): a __thread binding set by oh_tables_bind().  A FRAME thread binds its own picture on its own
 * thread.  The reference's SLICE / WAVEFRONT workers (pthread_slice.c: hls_decode_entry_wpp on worker threads, one CTU row each)
 * never bind: a thread without a binding of its own adopts the latest binding made by any thread (the picture in flight) — the
 * transient state (PU being assembled, last intra block, emulation buffers) stays per thread, the recorder takes its lock.
 * Frame threads AND slice threads at once (thread_type 4) would need the slots to tell pictures apart: not supported.
 */
#include <string.h>
#include "../../include/ohevc_tables.h"

typedef struct Planes { const uint8_t *data[3]; ptrdiff_t linesize[3]; int bound; } Planes;

typedef struct Tables {
    OhRecorder *rec;
    OhPicParams p;
    int bpp;
    Planes cur, refs[OH_MAX_REFS];
    oh_intra_accessor intra_fn;
    int untranslated;
    int why[8];                       /* untranslated calls by slot family: 0 transform_add, 1 put_pcm, 2 intra (no accessor), 3 intra (recorder), 4 edge emulation, 5 luma MC, 6 list-0 half, 7 PU record */
    /* residual: transform remembered for a coefficient buffer until its transform_add */
    const int16_t *tr_coeffs; int tr_kind, tr_flags;
    /* last intra block, waiting for its residual */
    int li_valid, li_c, li_x, li_y, li_log2; uint32_t li_index;
    /* edge emulation buffers (lc->edge_emu_buffer / edge_emu_buffer2) standing for picture rectangles */
    struct { const uint8_t *buf; ptrdiff_t linesize; int slot, plane, src_x, src_y, bw, bh, valid; } emu[2];
    int emu_next;
    /* list-0 half of a bi-predicted block: put_hevc_{q,e}pel into a stack int16 array */
    struct { const int16_t *tmp; int slot, mvx, mvy, valid; } l0;
    /* cross-component prediction: the transform unit's luma block, and the scale announced for the next chroma block (oh_tables_cross) */
    uint32_t last_y_tu; int last_y_valid, cross_pending, cross_scale;
    /* PU being assembled: recorded by the luma call, chroma calls only complete the weights */
    struct { int valid, x, y, w, h, ref[2], mv[2][2], weighted; OhWeights wp; } pu;
} Tables;

static __thread Tables T;
static __thread unsigned T_adopted;               /* generation of the shared binding this thread copied; 0: bound by itself or never */
static Tables G;                                  /* the latest binding (binding part only), for threads that never bind */
static volatile unsigned G_gen;
static volatile int G_untranslated, G_why[8];     /* what workers could not translate (oh_tables_finish runs on another thread) */

/* every slot starts here: a thread that never called oh_tables_bind() works on the picture in flight */
static inline void adopt(void)
{
    if ((T.rec && !T_adopted) || T_adopted == G_gen)
        return;
    oh_intra_accessor fn = G.intra_fn;
    memset(&T, 0, sizeof(T));
    T.rec = G.rec; T.p = G.p; T.bpp = G.bpp; T.cur = G.cur; T.intra_fn = fn;
    memcpy(T.refs, G.refs, sizeof(T.refs));
    T_adopted = G_gen;
}
#define UNTRANSLATED(k) do { T.untranslated++; T.why[k]++; if (T_adopted) { __sync_fetch_and_add(&G_untranslated, 1); __sync_fetch_and_add(&G_why[k], 1); } } while (0)

/* ---- pointer resolution ---- */
static int resolve(const Planes *pl, int plane, const uint8_t *ptr, int bpp, int *x, int *y)
{
    if (!pl->bound || !pl->data[plane] || pl->linesize[plane] <= 0)
        return 0;
    ptrdiff_t d = ptr - pl->data[plane], ls = pl->linesize[plane];
    ptrdiff_t yy = d >= 0 ? d / ls : -((-d + ls - 1) / ls);
    ptrdiff_t xx = (d - yy * ls) / bpp;
    /* a position left of column 0 shows up at the end of the row above: the row's padding (everything beyond the plane's width)
     * is split in the middle between "right of the picture" and "left of the next row" */
    const ptrdiff_t pw = T.p.width >> oh_hshift(&T.p, plane), row = ls / bpp;
    if (xx >= pw + (row - pw) / 2) { xx -= row; yy += 1; }
    *x = (int)xx; *y = (int)yy;
    return 1;
}

/* destination pointers always lie inside the current picture */
static int resolve_dst(const uint8_t *dst, ptrdiff_t stride, int *c, int *x, int *y)
{
    for (int k = 0; k < (T.p.chroma_format_idc ? 3 : 1); k++) {
        if (T.cur.linesize[k] != stride || !resolve(&T.cur, k, dst, T.bpp, x, y))
            continue;
        int pw = T.p.width >> oh_hshift(&T.p, k), ph = T.p.height >> oh_vshift(&T.p, k);
        if (*x >= 0 && *y >= 0 && *x < pw && *y < ph) { *c = k; return 1; }
    }
    return 0;
}

/* source pointers lie inside a reference plane or in an emulation buffer */
static int resolve_src(const uint8_t *src, ptrdiff_t stride, int plane, int *slot, int *x, int *y)
{
    for (int e = 0; e < 2; e++)
        if (T.emu[e].valid && T.emu[e].plane == plane && T.emu[e].linesize == stride) {
            ptrdiff_t d = src - T.emu[e].buf;
            if (d >= 0 && d / stride < T.emu[e].bh && (d % stride) / T.bpp < T.emu[e].bw) {
                *slot = T.emu[e].slot;
                *x = T.emu[e].src_x + (int)((d % stride) / T.bpp);
                *y = T.emu[e].src_y + (int)(d / stride);
                return 1;
            }
        }
    int best = -1;
    for (int s = 0; s < OH_MAX_REFS; s++) {
        int xx, yy;
        if (T.refs[s].linesize[plane] != stride || !resolve(&T.refs[s], plane, src, T.bpp, &xx, &yy))
            continue;
        int pw = T.p.width >> oh_hshift(&T.p, plane), ph = T.p.height >> oh_vshift(&T.p, plane);
        /* INSIDE the plane, nothing looser: a block whose taps leave the picture goes through emulated_edge_mc first
         * (hevc.c:1660-1675 and siblings), so a pointer that is not in an emulation buffer is the block's top-left sample inside
         * its reference plane.  A tolerance around the plane would let the pointer match a NEIGHBOURING allocation — the frames of
         * the DPB often lie back to back on the heap — and pick the wrong reference depending on where malloc put them. */
        if (xx >= 0 && yy >= 0 && xx < pw && yy < ph) { best = s; *x = xx; *y = yy; break; }
    }
    if (best < 0)
        return 0;
    *slot = best;
    return 1;
}

static void flush_pu(void)
{
    if (!T.pu.valid || !T.rec)
        return;
    if (oh_rec_pu(T.rec, T.pu.x, T.pu.y, T.pu.w, T.pu.h, T.pu.ref[0], T.pu.mv[0][0], T.pu.mv[0][1],
                  T.pu.ref[1], T.pu.mv[1][0], T.pu.mv[1][1], T.pu.weighted ? &T.pu.wp : NULL) != 0) {
        UNTRANSLATED(7);
    }
    T.pu.valid = 0;
}

/* ---- binding ---- */
void oh_tables_bind(OhRecorder *rec, uint8_t *const cur_data[3], const int cur_linesize[3])
{
    oh_intra_accessor keep = T.intra_fn;
    memset(&T, 0, sizeof(T));
    T.intra_fn = keep;
    T.rec = rec;
    T.p = *oh_rec_params(rec);
    T.bpp = T.p.bit_depth > 8 ? 2 : 1;
    for (int c = 0; c < 3; c++) { T.cur.data[c] = cur_data[c]; T.cur.linesize[c] = cur_linesize[c]; }
    T.cur.bound = 1;
    T_adopted = 0;
    G = T;                                               /* the picture in flight, for threads that never bind (slice / wavefront workers) */
    G_untranslated = 0; memset((void *)G_why, 0, sizeof(G_why));
    __sync_synchronize();
    G_gen = G_gen + 1 ? G_gen + 1 : 1;
}

void oh_tables_bind_ref(int slot, uint8_t *const data[3], const int linesize[3])
{
    if (slot < 0 || slot >= OH_MAX_REFS)
        return;
    for (int c = 0; c < 3; c++) { T.refs[slot].data[c] = data[c]; T.refs[slot].linesize[c] = linesize[c]; }
    T.refs[slot].bound = 1;
    G.refs[slot] = T.refs[slot];                         /* bound before the first slot call of the picture: workers adopt afterwards */
}

void oh_tables_set_intra_accessor(oh_intra_accessor fn) { T.intra_fn = fn; G.intra_fn = fn; }

int oh_tables_finish(void)
{
    flush_pu();
    return T.untranslated + G_untranslated;              /* this thread's and the adopting workers' */
}

void oh_tables_untranslated_by_family(int out[8])
{
    for (int i = 0; i < 8; i++) out[i] = T.why[i] + G_why[i];
}

/* ---- residual slots (hevc_cabac.c:1868-1949) ---- */
static void note_transform(const int16_t *coeffs, int kind, int flags)
{
    adopt();
    if (T.tr_coeffs != coeffs) { T.tr_coeffs = coeffs; T.tr_kind = OH_TU_BYPASS; T.tr_flags = 0; }
    if (kind >= 0) T.tr_kind = kind;
    T.tr_flags |= flags;
}
static void s_transform_skip(int16_t *coeffs, int16_t log2_size) { (void)log2_size; note_transform(coeffs, OH_TU_SKIP, 0); }
static void s_transform_rdpcm(int16_t *coeffs, int16_t log2_size, int mode)
{ (void)log2_size; note_transform(coeffs, -1, OH_TUF_RDPCM | (mode ? OH_TUF_RDPCM_VER : 0)); }
static void s_idct_4x4_luma(int16_t *coeffs) { note_transform(coeffs, OH_TU_DST4, 0); }
static void s_idct(int16_t *coeffs, int col_limit) { (void)col_limit; note_transform(coeffs, OH_TU_IDCT, 0); }
static void s_idct_dc(int16_t *coeffs) { note_transform(coeffs, OH_TU_IDCT, 0); }

static void transform_add_n(uint8_t *dst, int16_t *coeffs, ptrdiff_t stride, int log2)
{
    int c, x, y;
    adopt();
    flush_pu();
    if (!T.rec || !resolve_dst(dst, stride, &c, &x, &y)) { UNTRANSLATED(0); return; }
    int kind = OH_TU_BYPASS, flags = 0;
    if (T.tr_coeffs == coeffs) { kind = T.tr_kind; flags = T.tr_flags; }
    T.tr_coeffs = NULL;
    int is_intra = T.li_valid && T.li_c == c && T.li_x == x && T.li_y == y && T.li_log2 == log2;
    uint32_t tu = oh_rec_tu(T.rec, c, x, y, log2, kind, flags | (is_intra ? 0 : OH_TUF_ADD_NOW), coeffs);
    if (is_intra) {
        oh_rec_intra_attach_tu(T.rec, T.li_index, tu);
        T.li_valid = 0;
    }
    if (c == 0) {
        T.last_y_tu = tu; T.last_y_valid = tu != OH_NO_COEFF;
    } else if (T.cross_pending) {
        /* the host decoder left the chroma block WITHOUT the scaled luma residual: link it, the engine adds
         * (res_scale_val * luma residual) >> 3 once the luma residual exists */
        if (!T.last_y_valid || tu == OH_NO_COEFF || oh_rec_tu_cross(T.rec, tu, T.last_y_tu, T.cross_scale) != 0) { UNTRANSLATED(0); }
        T.cross_pending = 0;
    }
}

void oh_tables_cross(int res_scale_val)
{
    adopt();
    T.cross_pending = res_scale_val != 0;                 /* scale 0 adds nothing */
    T.cross_scale = res_scale_val;
}
static void s_transform_add0(uint8_t *d, int16_t *c, ptrdiff_t s) { transform_add_n(d, c, s, 2); }
static void s_transform_add1(uint8_t *d, int16_t *c, ptrdiff_t s) { transform_add_n(d, c, s, 3); }
static void s_transform_add2(uint8_t *d, int16_t *c, ptrdiff_t s) { transform_add_n(d, c, s, 4); }
static void s_transform_add3(uint8_t *d, int16_t *c, ptrdiff_t s) { transform_add_n(d, c, s, 5); }

/* ---- put_pcm (hevcdsp_template.c:30-43, called by hls_pcm_sample hevc.c:1587-1640): the samples are read from the
 * bitstream exactly like the C template does (get_bits, MSB first) and recorded as PCM blocks ---- */
static unsigned gb_read(struct GetBitContext *gb, int n)
{
    unsigned v = 0;
    for (int i = 0; i < n; i++, gb->index++)
        v = (v << 1) | ((gb->buffer[gb->index >> 3] >> (7 - (gb->index & 7))) & 1u);
    return v;
}
static void s_put_pcm(uint8_t *dst, ptrdiff_t stride, int width, int height, struct GetBitContext *gb, int pcm_bit_depth)
{
    int c, x, y;
    flush_pu();
    adopt();
    if (!T.rec || !gb || width < 4 || height < 4 || width > 32 || height > 32 || (width & (width - 1)) || (height & (height - 1)) ||
        !resolve_dst(dst, stride, &c, &x, &y)) {
        UNTRANSLATED(1);
        return;
    }
    int16_t rect[32 * 32], blk[32 * 32];
    for (int i = 0; i < width * height; i++)
        rect[i] = (int16_t)(gb_read(gb, pcm_bit_depth) << (T.p.bit_depth - pcm_bit_depth));
    /* the work list holds square blocks: a 4:2:2 chroma rectangle (w x 2w) goes as two squares */
    int s = width < height ? width : height, log2 = 0;
    while ((1 << log2) < s) log2++;
    for (int oy = 0; oy < height; oy += s)
        for (int ox = 0; ox < width; ox += s) {
            for (int yy = 0; yy < s; yy++)
                memcpy(blk + yy * s, rect + (oy + yy) * width + ox, sizeof(int16_t) * (size_t)s);
            (void)oh_rec_tu(T.rec, c, x + ox, y + oy, log2, OH_TU_PCM, OH_TUF_ADD_NOW, blk);
        }
}

/* ---- intra slots (hevc.c:1215-1417) ---- */
static void intra_pred_n(struct HEVCContext *s, int x0, int y0, int c_idx, int log2)
{
    int mode = 1, avail = 0;
    adopt();
    flush_pu();
    if (!T.rec || !T.intra_fn) { UNTRANSLATED(2); return; }
    T.intra_fn(s, x0, y0, c_idx, log2, &mode, &avail);
    int x = x0 >> oh_hshift(&T.p, c_idx), y = y0 >> oh_vshift(&T.p, c_idx);   /* x0,y0 are luma units, hevcpred_template.c:86-87 */
    const uint32_t idx = oh_rec_intra_idx(T.rec, c_idx, x, y, log2, mode, avail, OH_NO_COEFF);
    if (idx == OH_NO_COEFF) { UNTRANSLATED(3); return; }
    T.li_valid = 1; T.li_c = c_idx; T.li_x = x; T.li_y = y; T.li_log2 = log2; T.li_index = idx;
}
static void s_intra_pred2(struct HEVCContext *s, int x0, int y0, int c) { intra_pred_n(s, x0, y0, c, 2); }
static void s_intra_pred3(struct HEVCContext *s, int x0, int y0, int c) { intra_pred_n(s, x0, y0, c, 3); }
static void s_intra_pred4(struct HEVCContext *s, int x0, int y0, int c) { intra_pred_n(s, x0, y0, c, 4); }
static void s_intra_pred5(struct HEVCContext *s, int x0, int y0, int c) { intra_pred_n(s, x0, y0, c, 5); }

/* ---- edge emulation (hevc.c:1660-1675 and siblings) ---- */
static void s_emulated_edge_mc(uint8_t *buf, const uint8_t *src, ptrdiff_t buf_linesize, ptrdiff_t src_linesize,
                               int block_w, int block_h, int src_x, int src_y, int w, int h)
{
    (void)h;
    adopt();
    /* which reference plane?  The pointer may be far outside the allocation, so it is matched by
     * stride and by being consistent with (src_x, src_y): base + src_y*linesize + src_x*bpp == src */
    int slot = -1, plane = -1;
    for (int s = 0; s < OH_MAX_REFS && slot < 0; s++)
        for (int c = 0; c < (T.p.chroma_format_idc ? 3 : 1); c++) {
            if (!T.refs[s].bound || T.refs[s].linesize[c] != src_linesize)
                continue;
            if (T.refs[s].data[c] + (ptrdiff_t)src_y * src_linesize + (ptrdiff_t)src_x * T.bpp == src &&
                (T.p.width >> oh_hshift(&T.p, c)) == w) { slot = s; plane = c; break; }
        }
    /* a buffer stands for whatever its LATEST call put there: reuse its entry (the reference owns two
     * such buffers per thread, lc->edge_emu_buffer and edge_emu_buffer2, hevc.h:1162-1163) */
    int e;
    if (T.emu[0].valid && T.emu[0].buf == buf) e = 0;
    else if (T.emu[1].valid && T.emu[1].buf == buf) e = 1;
    else { e = T.emu_next; T.emu_next ^= 1; }
    T.emu[e].valid = slot >= 0;
    if (slot < 0) { UNTRANSLATED(4); return; }
    T.emu[e].buf = buf; T.emu[e].linesize = buf_linesize; T.emu[e].slot = slot; T.emu[e].plane = plane;
    T.emu[e].src_x = src_x; T.emu[e].src_y = src_y; T.emu[e].bw = block_w; T.emu[e].bh = block_h;
}

/* ---- interpolation slots (hevc.c:1641-1949) ---- */
static void mc_luma(uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride, const int16_t *src2,
                    int h, int mx, int my, int w, int weighted, int denom, int wa, int wb, int oa, int ob)
{
    int c, x, y, slot, sx, sy;
    adopt();
    flush_pu();
    if (!T.rec || !resolve_dst(dst, dststride, &c, &x, &y) || c != 0 || !resolve_src(src, srcstride, 0, &slot, &sx, &sy)) {
        UNTRANSLATED(5);
        return;
    }
    int mvx = ((sx - x) << 2) + mx, mvy = ((sy - y) << 2) + my;
    memset(&T.pu, 0, sizeof(T.pu));
    T.pu.valid = 1; T.pu.x = x; T.pu.y = y; T.pu.w = w; T.pu.h = h;
    T.pu.ref[0] = T.pu.ref[1] = -1;
    if (src2) {                                          /* second half of a bi-predicted block */
        if (!T.l0.valid || T.l0.tmp != src2) { UNTRANSLATED(6); T.pu.valid = 0; return; }
        T.pu.ref[0] = T.l0.slot; T.pu.mv[0][0] = T.l0.mvx; T.pu.mv[0][1] = T.l0.mvy;
        T.pu.ref[1] = slot; T.pu.mv[1][0] = mvx; T.pu.mv[1][1] = mvy;
        T.l0.valid = 0;
        if (weighted) {                                  /* (denom, w_l0, w_l1, o_l0, o_l1), hevc.c:1767-1773 */
            T.pu.weighted = 1; T.pu.wp.log2_denom[0] = (uint8_t)denom;
            T.pu.wp.w[0][0] = (int16_t)wa; T.pu.wp.w[1][0] = (int16_t)wb; T.pu.wp.o[0][0] = (int16_t)oa; T.pu.wp.o[1][0] = (int16_t)ob;
        }
    } else {
        /* which list a uni-predicted block uses is irrelevant to the arithmetic: record it as list 0 */
        T.pu.ref[0] = slot; T.pu.mv[0][0] = mvx; T.pu.mv[0][1] = mvy;
        if (weighted) {
            T.pu.weighted = 1; T.pu.wp.log2_denom[0] = (uint8_t)denom;
            T.pu.wp.w[0][0] = (int16_t)wa; T.pu.wp.o[0][0] = (int16_t)oa;
        }
    }
    /* complete unless explicit chroma weights are still to come (the Cr call closes it, chroma_weights): a worker thread's last PU
     * must not wait for a "next call" that never comes on that thread */
    if (!T.pu.weighted || !T.p.chroma_format_idc)
        flush_pu();
}

static void mc_luma_put(int16_t *dst, const uint8_t *src, ptrdiff_t srcstride, int mx, int my, int cur_x_unknown)
{
    (void)cur_x_unknown;
    /* list-0 half: the destination is a stack array, so the block position is not known yet; keep
     * the SOURCE position and turn it into an MV when the bi call names the destination */
    int slot, sx, sy;
    adopt();
    flush_pu();
    if (!resolve_src(src, srcstride, 0, &slot, &sx, &sy)) { UNTRANSLATED(6); T.l0.valid = 0; return; }
    T.l0.tmp = dst; T.l0.slot = slot; T.l0.mvx = (sx << 2) + mx; T.l0.mvy = (sy << 2) + my; T.l0.valid = 2;   /* absolute, fixed up below */
}

static void bi_fixup(const uint8_t *dst, ptrdiff_t ds);
#define QPEL_SLOTS(NAME) \
static void NAME##_put(int16_t *dst, ptrdiff_t ds, uint8_t *src, ptrdiff_t ss, int h, intptr_t mx, intptr_t my, int w) \
{ (void)ds; (void)h; (void)w; mc_luma_put(dst, src, ss, (int)mx, (int)my, 0); } \
static void NAME##_uni(uint8_t *dst, ptrdiff_t ds, uint8_t *src, ptrdiff_t ss, int h, intptr_t mx, intptr_t my, int w) \
{ mc_luma(dst, ds, src, ss, NULL, h, (int)mx, (int)my, w, 0, 0, 0, 0, 0, 0); } \
static void NAME##_uni_w(uint8_t *dst, ptrdiff_t ds, uint8_t *src, ptrdiff_t ss, int h, int denom, int wx, int ox, intptr_t mx, intptr_t my, int w) \
{ mc_luma(dst, ds, src, ss, NULL, h, (int)mx, (int)my, w, 1, denom, wx, 0, ox, 0); } \
static void NAME##_bi(uint8_t *dst, ptrdiff_t ds, uint8_t *src, ptrdiff_t ss, int16_t *src2, ptrdiff_t s2, int h, intptr_t mx, intptr_t my, int w) \
{ (void)s2; bi_fixup(dst, ds); mc_luma(dst, ds, src, ss, src2, h, (int)mx, (int)my, w, 0, 0, 0, 0, 0, 0); } \
static void NAME##_bi_w(uint8_t *dst, ptrdiff_t ds, uint8_t *src, ptrdiff_t ss, int16_t *src2, ptrdiff_t s2, int h, int denom, int w0, int w1, \
                        int o0, int o1, intptr_t mx, intptr_t my, int w) \
{ (void)s2; bi_fixup(dst, ds); mc_luma(dst, ds, src, ss, src2, h, (int)mx, (int)my, w, 1, denom, w0, w1, o0, o1); }

/* the list-0 half was stored with an absolute source position: make it relative to the block */
static void bi_fixup(const uint8_t *dst, ptrdiff_t ds)
{
    int c, x, y;
    if (T.l0.valid == 2 && resolve_dst(dst, ds, &c, &x, &y)) {
        T.l0.mvx -= x << 2; T.l0.mvy -= y << 2; T.l0.valid = 1;
    }
}
QPEL_SLOTS(q)

/* chroma calls: the geometry is implied by the luma call; only explicit weights are new */
static void chroma_weights(const uint8_t *dst, ptrdiff_t ds, int bi, int denom, int wa, int wb, int oa, int ob)
{
    int c, x, y;
    if (!T.pu.valid || !T.pu.weighted || !resolve_dst(dst, ds, &c, &x, &y) || c == 0)
        return;
    T.pu.wp.log2_denom[1] = (uint8_t)denom;
    T.pu.wp.w[0][c] = (int16_t)wa; T.pu.wp.o[0][c] = (int16_t)oa;
    if (bi) { T.pu.wp.w[1][c] = (int16_t)wb; T.pu.wp.o[1][c] = (int16_t)ob; }
    if (c == 2)
        flush_pu();
}
static void e_put(int16_t *dst, ptrdiff_t ds, uint8_t *src, ptrdiff_t ss, int h, intptr_t mx, intptr_t my, int w)
{ (void)dst; (void)ds; (void)src; (void)ss; (void)h; (void)mx; (void)my; (void)w; }
static void e_uni(uint8_t *dst, ptrdiff_t ds, uint8_t *src, ptrdiff_t ss, int h, intptr_t mx, intptr_t my, int w)
{ (void)dst; (void)ds; (void)src; (void)ss; (void)h; (void)mx; (void)my; (void)w; }
static void e_uni_w(uint8_t *dst, ptrdiff_t ds, uint8_t *src, ptrdiff_t ss, int h, int denom, int wx, int ox, intptr_t mx, intptr_t my, int w)
{ (void)src; (void)ss; (void)h; (void)mx; (void)my; (void)w; chroma_weights(dst, ds, 0, denom, wx, 0, ox, 0); }
static void e_bi(uint8_t *dst, ptrdiff_t ds, uint8_t *src, ptrdiff_t ss, int16_t *src2, ptrdiff_t s2, int h, intptr_t mx, intptr_t my, int w)
{ (void)dst; (void)ds; (void)src; (void)ss; (void)src2; (void)s2; (void)h; (void)mx; (void)my; (void)w; }
static void e_bi_w(uint8_t *dst, ptrdiff_t ds, uint8_t *src, ptrdiff_t ss, int16_t *src2, ptrdiff_t s2, int h, int denom, int w0, int w1,
                   int o0, int o1, intptr_t mx, intptr_t my, int w)
{ (void)src; (void)ss; (void)src2; (void)s2; (void)h; (void)mx; (void)my; (void)w; chroma_weights(dst, ds, 1, denom, w0, w1, o0, o1); }

/* ---- in-loop filter slots: whole-picture GPU passes replace the per-edge / per-CTB calls ---- */
static void s_lf_luma(uint8_t *p, ptrdiff_t s, int beta, int *tc, uint8_t *np, uint8_t *nq) { (void)p; (void)s; (void)beta; (void)tc; (void)np; (void)nq; }
static void s_lf_chroma(uint8_t *p, ptrdiff_t s, int *tc, uint8_t *np, uint8_t *nq) { (void)p; (void)s; (void)tc; (void)np; (void)nq; }
static void s_sao_band(uint8_t *d, uint8_t *s, ptrdiff_t sd, ptrdiff_t ss, struct SAOParams *sao, int *b, int w, int h, int c)
{ (void)d; (void)s; (void)sd; (void)ss; (void)sao; (void)b; (void)w; (void)h; (void)c; }
static void s_sao_edge(uint8_t *d, uint8_t *s, ptrdiff_t sd, ptrdiff_t ss, struct SAOParams *sao, int *b, int w, int h, int c,
                       uint8_t *ve, uint8_t *he, uint8_t *de)
{ (void)d; (void)s; (void)sd; (void)ss; (void)sao; (void)b; (void)w; (void)h; (void)c; (void)ve; (void)he; (void)de; }

static void s_up_h(int16_t *dst, ptrdiff_t dststride, uint8_t *src, ptrdiff_t srcstride, int x_EL, int x_BL, int block_w, int block_h, int widthEL,
                   const struct HEVCWindow *Enhscal, struct UpsamplInf *up_info)
{ (void)dst; (void)dststride; (void)src; (void)srcstride; (void)x_EL; (void)x_BL; (void)block_w; (void)block_h; (void)widthEL; (void)Enhscal; (void)up_info; }
static void s_up_v(uint8_t *dst, ptrdiff_t dststride, int16_t *src, ptrdiff_t srcstride, int y_BL, int x_EL, int y_EL, int block_w, int block_h, int widthEL,
                   int heightEL, const struct HEVCWindow *Enhscal, struct UpsamplInf *up_info)
{ (void)dst; (void)dststride; (void)src; (void)srcstride; (void)y_BL; (void)x_EL; (void)y_EL; (void)block_w; (void)block_h; (void)widthEL; (void)heightEL; (void)Enhscal; (void)up_info; }
static void s_up_frame(struct AVFrame *FrameEL, struct AVFrame *FrameBL, short *Buffer[3], const struct HEVCWindow *Enhscal, struct UpsamplInf *up_info, int channel)
{ (void)FrameEL; (void)FrameBL; (void)Buffer; (void)Enhscal; (void)up_info; (void)channel; }

/* ---- the hooks ---- */
void ff_hevcdsp_init_hip(HEVCDSPContext *c, const int bit_depth)
{
    (void)bit_depth;                                     /* the recorder carries the bit depth (OhPicParams) */
    c->transform_add[0] = s_transform_add0; c->transform_add[1] = s_transform_add1;
    c->transform_add[2] = s_transform_add2; c->transform_add[3] = s_transform_add3;
    c->transform_skip = s_transform_skip;
    c->transform_rdpcm = s_transform_rdpcm;
    c->idct_4x4_luma = s_idct_4x4_luma;
    for (int i = 0; i < 4; i++) { c->idct[i] = s_idct; c->idct_dc[i] = s_idct_dc; }
    c->sao_band_filter = s_sao_band;
    c->sao_edge_filter[0] = c->sao_edge_filter[1] = s_sao_edge;
    for (int i = 0; i < 10; i++)
        for (int a = 0; a < 2; a++)
            for (int b = 0; b < 2; b++) {
                c->put_hevc_qpel[i][a][b] = q_put; c->put_hevc_qpel_uni[i][a][b] = q_uni; c->put_hevc_qpel_uni_w[i][a][b] = q_uni_w;
                c->put_hevc_qpel_bi[i][a][b] = q_bi; c->put_hevc_qpel_bi_w[i][a][b] = q_bi_w;
                c->put_hevc_epel[i][a][b] = e_put; c->put_hevc_epel_uni[i][a][b] = e_uni; c->put_hevc_epel_uni_w[i][a][b] = e_uni_w;
                c->put_hevc_epel_bi[i][a][b] = e_bi; c->put_hevc_epel_bi_w[i][a][b] = e_bi_w;
            }
    c->hevc_h_loop_filter_luma = c->hevc_v_loop_filter_luma = s_lf_luma;
    c->hevc_h_loop_filter_luma_c = c->hevc_v_loop_filter_luma_c = s_lf_luma;
    c->hevc_h_loop_filter_chroma = c->hevc_v_loop_filter_chroma = s_lf_chroma;
    c->hevc_h_loop_filter_chroma_c = c->hevc_v_loop_filter_chroma_c = s_lf_chroma;
    c->put_pcm = s_put_pcm;
    /* SHVC: the enhancement-layer decoder up-samples its inter-layer reference picture CTB by CTB where prediction units read it
     * (ff_upsample_block, hevc_filter.c:1370-1426 -> upsample_block_luma / upsample_block_mc -> these four slot families).  With the
     * engine that picture is resampled in HBM (oh_pic_upsample, issued by the decoder's hand-over before the picture's work list,
     * INTEGRATION.md section 8); the slots do nothing, and ff_upsample_block keeps the part that is host work: is_upsampled[] and the
     * scaled motion field (ff_upscale_mv_block).  upsample_base_layer_frame (the whole-picture variant of builds without
     * ACTIVE_PU_UPSAMPLING, hevc.c:3241) is replaced the same way. */
    for (int i = 0; i < 3; i++) {
        c->upsample_filter_block_luma_h[i] = s_up_h; c->upsample_filter_block_cr_h[i] = s_up_h;
        c->upsample_filter_block_luma_v[i] = s_up_v; c->upsample_filter_block_cr_v[i] = s_up_v;
    }
    c->upsample_base_layer_frame = s_up_frame;
}

void ff_hevcpred_init_hip(HEVCPredContext *c, const int bit_depth)
{
    (void)bit_depth;
    c->intra_pred[0] = s_intra_pred2; c->intra_pred[1] = s_intra_pred3;
    c->intra_pred[2] = s_intra_pred4; c->intra_pred[3] = s_intra_pred5;
    /* pred_planar / pred_dc / pred_angular are only reached through intra_pred[] (hevcpred_template.c:329-343) */
}

void ff_videodsp_init_hip(VideoDSPContext *c, int bit_depth)
{
    (void)bit_depth;
    c->emulated_edge_mc = s_emulated_edge_mc;
}
