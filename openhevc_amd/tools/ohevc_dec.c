/*
 * ohevc_dec — SDL-free counterpart of the reference's `hevc` test harness (main_hm/main.c:134-311, options main_hm/getopt.c:47-62)
 * for the MI355X engine:
 *
 *     ohevc_dec -i stream.bin -F <front-end.so> [-c] [-n] [-o out.yuv] [-s frames] [-p threads -f 2]
 *
 * reads a raw Annex-B file, splits it into access units (oh_annexb_split: what the reference's harness gets from libavformat's raw
 * HEVC demuxer through hevc_parser.c), hands every access unit to the FRONT END — the host decoder that does what stays on the host
 * (parameter sets, slice headers, CABAC, motion-vector and mode derivation: SURVEY.md §8 "out of scope") and records the picture's
 * work list through the table slots of libohevc_hip.so (include/ohevc_tables.h) — runs the work list on the GPU
 * (oh_frame_submit), and checks the result against the stream's decoded-picture-hash SEI with MD5s computed on the GPU
 * (oh_pics_md5): the reference's `decode_checksum_sei` check (hevc.c:4146-4162), its verdict lines included.  It ends with the
 * reference's summary line (main.c:304-306):
 *
 *     frame= N fps= F time= T video_size= WxH
 *
 * The front end is named on the command line and loaded with dlopen(); it exports the libOpenHevc* API of the reference's wrapper
 * (gpac/modules/openhevc_dec/openHevcWrapper.h:79-98: libOpenHevcInit / StartDecoder / Decode / Close) plus
 *     const OhFrame *ref_hooked_finish(int *cur_id, int *poc, int *untranslated);
 * which returns the work list of the access unit just decoded (picture ids = the front end's DPB slots).  -c: do not check MD5
 * (as the reference's flag), -n: accepted and ignored (there is no display), -o: write the decoded pictures (whole coded planes,
 * decode order), -s: stop after so many pictures, -p / -f: threads of the front end as in the reference's harness — -f 2 (slice /
 * wavefront threads) is what the recording slots support with more than one thread.
 */
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../../include/ohevc_annexb.h"
#include "../../include/ohevc_frame.h"
#include "../../include/ohevc_hip.h"

typedef void *(*fe_init_fn)(int nb_pthreads, int thread_type);
typedef int (*fe_start_fn)(void *h);
typedef int (*fe_decode_fn)(void *h, const unsigned char *buf, int len, int64_t pts);
typedef void (*fe_close_fn)(void *h);
typedef const OhFrame *(*fe_finish_fn)(int *cur_id, int *poc, int *untranslated);
/* the output side of the front end's public API (openHevcWrapper.h:38-66, :86): the frame libOpenHevcDecode just released for
 * output — in OUTPUT order, inside its conformance window — as plane pointers into the front end's picture buffers; the harness
 * only uses them to learn WHICH picture that is (fe_locate) and takes the samples from the engine's copy */
typedef struct FeRational { int num, den; } FeRational;
typedef struct FeFrameInfo {
    int nYPitch, nUPitch, nVPitch, nBitDepth, nWidth, nHeight, chromat_format;
    FeRational sample_aspect_ratio, frameRate;
    int display_picture_number, flag;
    int64_t nTimeStamp;
} FeFrameInfo;
typedef struct FeFrame { void **pvY, **pvU, **pvV; FeFrameInfo frameInfo; } FeFrame;
typedef int (*fe_output_fn)(void *h, int got_picture, FeFrame *frame);
typedef int (*fe_locate_fn)(const void *luma, int *x, int *y);     /* DPB slot (= picture id of the work lists) and the window's origin */

static void usage(const char *prog)
{
    printf("%s: -i <file> -F <front end> [-b] [-c] [-n] [-o <output file>] [-s <num>]\n", prog);
    printf("     -b : boundary strengths on the GPU (the front end hands over its motion field instead of finished grids)\n");
    printf("     -c : no check md5\n");
    printf("     -F <shared object of the host decoder with the recording table slots linked in>\n");
    printf("     -i <input file>\n");
    printf("     -n : no display (there is none)\n");
    printf("     -o <output file>\n");
    printf("     -s <num> Stop after num frames \n");
    printf("     -p <number of threads of the front end> \n");
    printf("     -f <thread type> (2: slice; with -p 1 anything)\n");
}

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}

#define MAX_DPB 64

int main(int argc, char **argv)
{
    const char *input = NULL, *front = NULL, *output = NULL;
    int check_md5 = 1, num_frames = 0, nb_pthreads = 1, thread_type = 1, bs_on_gpu = 0;
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
        if (a[0] != '-' || !a[1] || a[2]) { usage(argv[0]); return 2; }
        const int need = strchr("iFospf", a[1]) != NULL;
        if (need && i + 1 >= argc) { usage(argv[0]); return 2; }
        switch (a[1]) {
        case 'b': bs_on_gpu = 1; break;
        case 'c': check_md5 = 0; break;
        case 'n': break;
        case 'i': input = argv[++i]; break;
        case 'F': front = argv[++i]; break;
        case 'o': output = argv[++i]; break;
        case 's': num_frames = atoi(argv[++i]); break;
        case 'p': nb_pthreads = atoi(argv[++i]); break;
        case 'f': thread_type = atoi(argv[++i]); break;
        default: usage(argv[0]); return 2;
        }
    }
    if (!input) { printf("No input file specified.\nSpecify it with: -i <filename>\n"); return 1; }
    if (!front) { printf("No front end specified.\nSpecify it with: -F <shared object>\n"); return 1; }
    if (nb_pthreads < 1 || (nb_pthreads > 1 && thread_type != 2)) {
        printf("the recording table slots support several front-end threads as slice / wavefront threads only (-f 2)\n");
        return 2;
    }

    FILE *fi = fopen(input, "rb");
    if (!fi) { printf("%s", input); return 1; }
    fseek(fi, 0, SEEK_END);
    const long fsize = ftell(fi);
    fseek(fi, 0, SEEK_SET);
    uint8_t *data = (uint8_t *)calloc((size_t)(fsize > 0 ? fsize : 0) + 64, 1);          /* zero padding behind the last packet (FF_INPUT_BUFFER_PADDING_SIZE) */
    if (!data || fread(data, 1, (size_t)fsize, fi) != (size_t)fsize) { fprintf(stderr, "could not read %s\n", input); return 1; }
    fclose(fi);

    long n_au = oh_annexb_split(data, (size_t)fsize, NULL, 0);
    size_t cap = (size_t)(-n_au) + 1;
    size_t *au = (size_t *)malloc(cap * sizeof(*au));
    n_au = oh_annexb_split(data, (size_t)fsize, au, cap);
    if (n_au < 0) { fprintf(stderr, "access-unit split failed\n"); return 1; }

    void *so = dlopen(front, RTLD_NOW | RTLD_LOCAL);
    if (!so) { fprintf(stderr, "could not open the front end: %s\n", dlerror()); return 1; }
    fe_init_fn fe_init = (fe_init_fn)dlsym(so, "libOpenHevcInit");
    fe_start_fn fe_start = (fe_start_fn)dlsym(so, "libOpenHevcStartDecoder");
    fe_decode_fn fe_decode = (fe_decode_fn)dlsym(so, "libOpenHevcDecode");
    fe_close_fn fe_close = (fe_close_fn)dlsym(so, "libOpenHevcClose");
    fe_finish_fn fe_finish = (fe_finish_fn)dlsym(so, "ref_hooked_finish");
    fe_output_fn fe_output = (fe_output_fn)dlsym(so, "libOpenHevcGetOutput");
    fe_locate_fn fe_locate = (fe_locate_fn)dlsym(so, "ref_hooked_locate");
    if (!fe_init || !fe_start || !fe_decode || !fe_close || !fe_finish || !fe_output || !fe_locate) {
        fprintf(stderr, "%s does not export libOpenHevcInit / StartDecoder / Decode / GetOutput / Close / ref_hooked_finish / ref_hooked_locate\n", front);
        return 1;
    }
    if (bs_on_gpu) {
        void (*fe_bs)(int) = (void (*)(int))dlsym(so, "ref_hooked_bs_from_motion");
        if (!fe_bs) { fprintf(stderr, "%s cannot hand over its motion field (no ref_hooked_bs_from_motion)\n", front); return 1; }
        fe_bs(1);                                             /* work lists with OhFrame.bs_in: bs_kernel derives both grids */
    }
    void *h = fe_init(nb_pthreads, thread_type);
    if (!h || fe_start(h) != 1) { fprintf(stderr, "could not open OpenHevc\n"); return 1; }

    OhEngine *e = NULL;
    if (oh_engine_create(&e, 0) != OH_OK) { fprintf(stderr, "could not create the engine (no HIP device?)\n"); return 1; }

    int engine_id[MAX_DPB];
    OhPicParams engine_p[MAX_DPB];
    for (int i = 0; i < MAX_DPB; i++) engine_id[i] = -1;
    FILE *fo = NULL;
    int nb_frame = 0, nb_decoded = 0, width = 0, height = 0, bad_planes = 0, hashed = 0, unverified = 0, rc = 0;
    uint8_t *planes[3] = { NULL, NULL, NULL };
    const double t0 = now_s();

    for (long k = 0; !rc; k++) {                             /* k >= n_au: flushing, one released picture per call */
        const uint8_t *buf = k < n_au ? data + au[k] : NULL;
        const size_t len = k < n_au ? au[k + 1] - au[k] : 0;
        OhPictureHash want;
        memset(&want, 0, sizeof(want));
        if (check_md5 && k < n_au) {
            OhNal units[256];
            long nu = oh_annexb_nal_units(buf, len, units, 256);
            for (long u = 0; u < nu && u < 256; u++)
                if (units[u].type == 39 || units[u].type == 40) {
                    OhPictureHash ph;
                    if (oh_sei_picture_hash(buf + units[u].offset, units[u].size, &ph) == 1)
                        want = ph;
                }
        }
        const int got_picture = k < n_au ? fe_decode(h, buf, (int)len, k) : fe_decode(h, NULL, 0, k);      /* past the last access unit: flush (main.c:225) */
        if (got_picture < 0) { fprintf(stderr, "front end failed on access unit %ld\n", k); rc = 1; break; }
        int cur = -1, poc = 0, untranslated = 0;
        const OhFrame *f = k < n_au ? fe_finish(&cur, &poc, &untranslated) : NULL;
        if (f) {
            if (untranslated) { fprintf(stderr, "%d table-slot calls of picture %d could not be turned into work-list items\n", untranslated, nb_decoded); rc = 1; break; }
            /* engine pictures for the DPB slots this work list names */
            OhFrame g = *f;
            int ids[1 + OH_MAX_REFS], n_ids = 0;
            ids[n_ids++] = cur;
            for (int r = 0; r < OH_MAX_REFS; r++)
                if (f->ref_pics[r] >= 0) ids[n_ids++] = f->ref_pics[r];
            for (int q = 0; q < n_ids && !rc; q++) {
                const int d = ids[q];
                if (d < 0 || d >= MAX_DPB) { fprintf(stderr, "DPB slot %d out of range\n", d); rc = 1; break; }
                const int same = engine_id[d] >= 0 && engine_p[d].width == f->p.width && engine_p[d].height == f->p.height &&
                                 engine_p[d].bit_depth == f->p.bit_depth && engine_p[d].chroma_format_idc == f->p.chroma_format_idc;
                if (!same) {
                    if (engine_id[d] >= 0) oh_pic_free(e, engine_id[d]);
                    if (oh_pic_alloc(e, &f->p, &engine_id[d]) != OH_OK) { fprintf(stderr, "%s\n", oh_engine_last_error(e)); rc = 1; break; }
                    engine_p[d] = f->p;
                }
            }
            if (rc) break;
            g.cur_pic = engine_id[cur];
            for (int r = 0; r < OH_MAX_REFS; r++)
                g.ref_pics[r] = f->ref_pics[r] >= 0 ? engine_id[f->ref_pics[r]] : -1;
            if (oh_frame_submit(e, &g) != OH_OK) { fprintf(stderr, "picture %d: %s\n", nb_decoded, oh_engine_last_error(e)); rc = 1; break; }
            if (check_md5 && want.present && want.hash_type == 0) {
                uint8_t got[48];
                if (oh_pics_md5(e, &g.cur_pic, 1, got) != OH_OK) { fprintf(stderr, "%s\n", oh_engine_last_error(e)); rc = 1; break; }
                for (int c = 0; c < (f->p.chroma_format_idc ? 3 : 1); c++) {
                    if (memcmp(got + 16 * c, want.md5[c], 16)) {
                        printf("Incorrect MD5 (poc: %d, plane: %d)\n", poc, c);
                        bad_planes++;
                    } else
                        printf("Correct MD5 (poc: %d, plane: %d)\n", poc, c);
                }
                hashed++;
            } else if (check_md5 && want.present) {
                printf("picture hash of type %d (CRC / checksum) present but not verified (poc: %d)\n", want.hash_type, poc);
                unverified++;
            }
            nb_decoded++;
        }
        if (got_picture == 0) {
            if (k >= n_au)
                break;                                        /* flushed */
            continue;                                         /* nothing released for output yet (reordering), or no picture in the access unit */
        }
        /* the picture the front end released for output (bumping and cropping are its host logic: hevc_refs.c:182-290) */
        FeFrame fr;
        memset(&fr, 0, sizeof(fr));
        fe_output(h, 1, &fr);
        int ox = 0, oy = 0;
        const int slot = fe_locate((const void *)fr.pvY, &ox, &oy);
        if (slot < 0 || slot >= MAX_DPB || engine_id[slot] < 0) { fprintf(stderr, "output picture %d is not a picture of the engine (slot %d)\n", nb_frame, slot); rc = 1; break; }
        const OhPicParams *op = &engine_p[slot];
        if (width != fr.frameInfo.nWidth || height != fr.frameInfo.nHeight) {
            width = fr.frameInfo.nWidth; height = fr.frameInfo.nHeight;
            if (fo) fclose(fo);
            fo = NULL;
            if (output) {
                char name[1024];
                char stem[900];
                snprintf(stem, sizeof(stem), "%s", output);
                const size_t sl = strlen(stem);
                if (sl > 4 && stem[sl - 4] == '.') stem[sl - 4] = 0;                  /* getopt.c:174-175 */
                snprintf(name, sizeof(name), "%s_%dx%d.yuv", stem, width, height);   /* main.c:231 */
                fo = fopen(name, "wb");
                if (!fo) { fprintf(stderr, "could not open %s\n", name); rc = 1; break; }
            }
        }
        if (fo) {
            const size_t bpp = op->bit_depth > 8 ? 2 : 1;
            const int cf = op->chroma_format_idc;
            const int hs = cf == 1 || cf == 2, vs = cf == 1;
            ptrdiff_t strides[3];
            size_t bytes[3];
            for (int c = 0; c < 3; c++) {
                const size_t w = c ? (size_t)(width >> hs) : (size_t)width, hh = c ? (size_t)(height >> vs) : (size_t)height;
                strides[c] = (ptrdiff_t)(w * bpp);
                bytes[c] = cf || !c ? w * bpp * hh : 0;
                planes[c] = (uint8_t *)realloc(planes[c], bytes[c] ? bytes[c] : 1);
            }
            const OhWindow win = { ox, op->width - ox - width, oy, op->height - oy - height };
            if (win.right < 0 || win.bottom < 0) { fprintf(stderr, "output window %dx%d+%d+%d leaves the %dx%d picture\n", width, height, ox, oy, op->width, op->height); rc = 1; break; }
            if (oh_pic_download_window(e, engine_id[slot], &win, planes, strides) != OH_OK) { fprintf(stderr, "%s\n", oh_engine_last_error(e)); rc = 1; break; }
            for (int c = 0; c < 3; c++)
                fwrite(planes[c], 1, bytes[c], fo);
        }
        nb_frame++;
        if (nb_frame == num_frames)
            break;
    }
    if (oh_engine_sync(e) != OH_OK) { fprintf(stderr, "%s\n", oh_engine_last_error(e)); rc = 1; }
    const double t = now_s() - t0;
    if (fo) fclose(fo);
    for (int c = 0; c < 3; c++) free(planes[c]);
    fe_close(h);
    oh_engine_destroy(e);
    free(au);
    free(data);
    if (check_md5)
        printf("md5: %d pictures checked, %d planes differ%s\n", hashed, bad_planes, unverified ? " (some pictures carry CRC / checksum hashes: not verified)" : "");
    printf("frame= %d fps= %.0f time= %.2f video_size= %dx%d\n", nb_frame, t > 0 ? nb_frame / t : 0.0, t, width, height);
    return rc ? rc : bad_planes ? 3 : 0;
}
