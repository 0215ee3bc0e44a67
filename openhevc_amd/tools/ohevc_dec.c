/*
 * ohevc_dec — SDL-free, libavformat-free counterpart of the reference's `hevc` test harness (main_hm/main.c:115-311, options
 * main_hm/getopt.c:47-62): a client of the PUBLIC API only.
 *
 *     ohevc_dec -i stream.bin -F <library> [-c] [-n] [-g] [-o out.yuv] [-s frames] [-p threads] [-f thread type] [-b]
 *
 * <library> is any shared object that exports the 18 libOpenHevc* functions of gpac/modules/openhevc_dec/openHevcWrapper.h:79-98:
 * the drop-in library with the MI355X engine inside (libopenhevc_hip.so: INTEGRATION.md) or the reference's own (libopenhevc_ref.so,
 * libopenhevc_ref_sse.so) — the same loop, the same clock, so `frame= N fps= F` of the two are comparable line by line.
 * The harness reads a raw Annex-B file, splits it into access units (oh_annexb_split: what the reference's harness gets from
 * libavformat's raw HEVC demuxer through hevc_parser.c:40-87) and runs main.c's loop: libOpenHevcInit, SetCheckMD5, StartDecoder,
 * then libOpenHevcDecode per access unit; for every released picture GetPictureInfo, and — with -o — GetOutputCpy into packed planes
 * written to <out>_WxH.yuv exactly as main.c:246-274 does; -g fetches every released picture with libOpenHevcGetOutput (the
 * display path of main.c:268-272 without a display) — with the drop-in library that is the device-to-host copy of the picture;
 * without -o / -g a picture is decoded and never looked at, as with the reference's `hevc -n`.  After the last access unit it
 * flushes with empty packets (main.c:225) and prints the reference's summary line (main.c:304-306):
 *
 *     frame= N fps= F time= T video_size= WxH
 *
 * The MD5 check (-c switches it off, as in the reference) is the DECODER's (hevc.c:4146-4169): its "Correct MD5 (poc, plane)" /
 * "Incorrect MD5" verdicts arrive on the library's log (stderr); the harness passes that log through to stdout, counts the verdicts
 * and exits with 3 when a plane differed.  -b: boundary strengths derived on the GPU (drop-in library: OHEVC_BS_FROM_MOTION).
 */
#include <dlfcn.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "../../include/ohevc_annexb.h"

/* openHevcWrapper.h:38-77, re-declared: the harness is compiled without the reference tree */
typedef void *OpenHevc_Handle;
typedef struct OpenHevc_Rational { int num, den; } OpenHevc_Rational;
typedef struct OpenHevc_FrameInfo {
    int nYPitch, nUPitch, nVPitch, nBitDepth, nWidth, nHeight, chromat_format;
    OpenHevc_Rational sample_aspect_ratio, frameRate;
    int display_picture_number, flag;
    int64_t nTimeStamp;
} OpenHevc_FrameInfo;
typedef struct OpenHevc_Frame { void **pvY, **pvU, **pvV; OpenHevc_FrameInfo frameInfo; } OpenHevc_Frame;
typedef struct OpenHevc_Frame_cpy { void *pvY, *pvU, *pvV; OpenHevc_FrameInfo frameInfo; } OpenHevc_Frame_cpy;
enum { YUV420 = 0, YUV422, YUV444 };

static struct {
    OpenHevc_Handle (*Init)(int nb_pthreads, int thread_type);
    int  (*StartDecoder)(OpenHevc_Handle);
    int  (*Decode)(OpenHevc_Handle, const unsigned char *buff, int nal_len, int64_t pts);
    void (*GetPictureInfo)(OpenHevc_Handle, OpenHevc_FrameInfo *);
    int  (*GetOutput)(OpenHevc_Handle, int got_picture, OpenHevc_Frame *);
    int  (*GetOutputCpy)(OpenHevc_Handle, int got_picture, OpenHevc_Frame_cpy *);
    void (*SetCheckMD5)(OpenHevc_Handle, int val);
    void (*SetDebugMode)(OpenHevc_Handle, int val);
    void (*Close)(OpenHevc_Handle);
} api;

static void usage(const char *prog)
{
    printf("%s: -i <file> -F <library> [-b] [-c] [-f <type>] [-g] [-l <passes>] [-n] [-o <output file>] [-p <threads>] [-s <num>]\n", prog);
    printf("     -b : boundary strengths on the GPU (drop-in library)\n");
    printf("     -c : no check md5\n");
    printf("     -f <thread type> (1: frame, 2: slice, 4: frameslice)\n");
    printf("     -F <shared object exporting the libOpenHevc* API (openHevcWrapper.h)>\n");
    printf("     -g : get every released picture (libOpenHevcGetOutput)\n");
    printf("     -i <input file>\n");
    printf("     -l <passes> : decode the file that many times in a row (fps per pass)\n");
    printf("     -n : no display (there is none)\n");
    printf("     -o <output file>\n");
    printf("     -p <number of threads> \n");
    printf("     -s <num> Stop after num frames \n");
}

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}

/* the library's log (file descriptor 2) -> stdout, the MD5 verdicts counted on the way */
static struct { int rd, saved_err, correct, incorrect; pthread_t th; } logp;
static void *log_reader(void *arg)
{
    (void)arg;
    FILE *in = fdopen(logp.rd, "r");
    char line[4096];
    while (in && fgets(line, sizeof(line), in)) {
        if (strstr(line, "Incorrect MD5")) logp.incorrect++;
        else if (strstr(line, "Correct MD5")) logp.correct++;
        fputs(line, stdout);
    }
    if (in) fclose(in);
    return NULL;
}
static void log_capture_begin(void)
{
    int fds[2];
    fflush(stderr);
    if (pipe(fds) != 0) return;
    logp.saved_err = dup(2);
    dup2(fds[1], 2);
    close(fds[1]);
    logp.rd = fds[0];
    pthread_create(&logp.th, NULL, log_reader, NULL);
}
static void log_capture_end(void)
{
    if (!logp.saved_err) return;
    fflush(stderr);
    dup2(logp.saved_err, 2);                                  /* closes the pipe's write end: the reader sees end of file */
    close(logp.saved_err);
    pthread_join(logp.th, NULL);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const char *input = NULL, *library = NULL, *output = NULL;
    int check_md5 = 1, num_frames = 0, nb_pthreads = 1, thread_type = 1, bs_on_gpu = 0, get_output = 0, loops = 1;
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
        if (a[0] != '-' || !a[1] || a[2]) { usage(argv[0]); return 2; }
        const int need = strchr("iFospfl", a[1]) != NULL;
        if (need && i + 1 >= argc) { usage(argv[0]); return 2; }
        switch (a[1]) {
        case 'b': bs_on_gpu = 1; break;
        case 'c': check_md5 = 0; break;
        case 'g': get_output = 1; break;
        case 'n': break;
        case 'i': input = argv[++i]; break;
        case 'F': library = argv[++i]; break;
        case 'o': output = argv[++i]; break;
        case 's': num_frames = atoi(argv[++i]); break;
        case 'p': nb_pthreads = atoi(argv[++i]); break;
        case 'f': thread_type = atoi(argv[++i]); break;
        case 'l': loops = atoi(argv[++i]); if (loops < 1) loops = 1; break;
        default: usage(argv[0]); return 2;
        }
    }
    if (!input) { printf("No input file specified.\nSpecify it with: -i <filename>\n"); return 1; }
    if (!library) { printf("No decoder library specified.\nSpecify it with: -F <shared object>\n"); return 1; }
    if (bs_on_gpu) setenv("OHEVC_BS_FROM_MOTION", "1", 1);

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

    void *so = dlopen(library, RTLD_NOW | RTLD_LOCAL);
    if (!so) { fprintf(stderr, "could not open the decoder library: %s\n", dlerror()); return 1; }
#define SYM(field, name) (*(void **)&api.field = dlsym(so, name))
    SYM(Init, "libOpenHevcInit"); SYM(StartDecoder, "libOpenHevcStartDecoder"); SYM(Decode, "libOpenHevcDecode");
    SYM(GetPictureInfo, "libOpenHevcGetPictureInfo"); SYM(GetOutput, "libOpenHevcGetOutput"); SYM(GetOutputCpy, "libOpenHevcGetOutputCpy");
    SYM(SetCheckMD5, "libOpenHevcSetCheckMD5"); SYM(SetDebugMode, "libOpenHevcSetDebugMode"); SYM(Close, "libOpenHevcClose");
#undef SYM
    if (!api.Init || !api.StartDecoder || !api.Decode || !api.GetPictureInfo || !api.GetOutput || !api.GetOutputCpy || !api.SetCheckMD5 || !api.Close) {
        fprintf(stderr, "%s does not export the libOpenHevc* API (openHevcWrapper.h)\n", library);
        return 1;
    }

    log_capture_begin();
    int rc = 0;
    OpenHevc_Handle h = api.Init(nb_pthreads, thread_type);
    if (!h) { fprintf(stderr, "could not open OpenHevc\n"); log_capture_end(); return 1; }
    api.SetCheckMD5(h, check_md5);
    if (api.SetDebugMode) api.SetDebugMode(h, 0);
    if (api.StartDecoder(h) != 1) { fprintf(stderr, "could not start the decoder (threads %d, thread type %d)\n", nb_pthreads, thread_type); log_capture_end(); return 2; }

    FILE *fo = NULL;
    OpenHevc_Frame frame;
    OpenHevc_Frame_cpy cpy;
    memset(&frame, 0, sizeof(frame));
    memset(&cpy, 0, sizeof(cpy));
    int nb_frame = 0, width = -1, height = -1, stop = 0;
    const double t0 = now_s();
    /* -l N: the file is decoded N times in a row through the same decoder, as if concatenated (a stream opens with an IDR picture and its
     * parameter sets): the passes after the first show the decoder with its buffer pools filled */
    const long n_total = n_au * loops;
    double t_pass = t0;
    int frames_pass = 0;
    for (long k = 0; !stop; k++) {                           /* k >= n_total: flushing with empty packets (main.c:225) */
        const int flushing = k >= n_total;
        const long ka = flushing ? 0 : k % n_au;
        if (loops > 1 && k > 0 && k <= n_total && k % n_au == 0) {
            const double tn = now_s();
            printf("pass %ld: %d pictures released in %.3f s = %.1f fps\n", k / n_au, nb_frame - frames_pass, tn - t_pass, (nb_frame - frames_pass) / (tn - t_pass));
            t_pass = tn; frames_pass = nb_frame;
        }
        const int got_picture = api.Decode(h, flushing ? NULL : data + au[ka], flushing ? 0 : (int)(au[ka + 1] - au[ka]), k);
        if (got_picture < 0) { fprintf(stderr, "decoder failed on access unit %ld\n", k); rc = 1; break; }
        if (got_picture > 0) {
            api.GetPictureInfo(h, &frame.frameInfo);
            if (width != frame.frameInfo.nWidth || height != frame.frameInfo.nHeight) {
                width = frame.frameInfo.nWidth; height = frame.frameInfo.nHeight;
                if (fo) fclose(fo);
                fo = NULL;
                if (output) {
                    char name[1024], stem[900];
                    snprintf(stem, sizeof(stem), "%s", output);
                    const size_t sl = strlen(stem);
                    if (sl > 4 && stem[sl - 4] == '.') stem[sl - 4] = 0;                  /* getopt.c:174-175 */
                    snprintf(name, sizeof(name), "%s_%dx%d.yuv", stem, width, height);   /* main.c:231 */
                    fo = fopen(name, "wb");
                    if (!fo) { fprintf(stderr, "could not open %s\n", name); rc = 1; break; }
                }
            }
            if (get_output)
                api.GetOutput(h, 1, &frame);                                              /* main.c:268-272 without the display */
            if (fo) {
                /* packed planes: rows of width << pixel_shift bytes (libOpenHevcGetPictureInfoCpy's pitches); the reference's harness sizes
                 * its buffers from GetPictureInfo's (main.c:246-251), which are never smaller */
                const int ps = frame.frameInfo.nBitDepth > 8, cf = frame.frameInfo.chromat_format;
                const int hs = cf != YUV444, vs = cf == YUV420;
                const size_t yb = ((size_t)width << ps) * (size_t)height, cb = ((size_t)(width >> hs) << ps) * (size_t)(height >> vs);
                cpy.pvY = realloc(cpy.pvY, yb ? yb : 1); cpy.pvU = realloc(cpy.pvU, cb ? cb : 1); cpy.pvV = realloc(cpy.pvV, cb ? cb : 1);
                api.GetOutputCpy(h, 1, &cpy);
                fwrite(cpy.pvY, 1, yb, fo);
                fwrite(cpy.pvU, 1, cb, fo);
                fwrite(cpy.pvV, 1, cb, fo);
            }
            nb_frame++;
            if (nb_frame == num_frames)
                stop = 1;
        } else if (flushing) {
            stop = 1;
        }
    }
    const double t = now_s() - t0;                           /* the decode loop, as main.c's SDL_GetTime() around it; Close (which waits for the GPU) below */
    if (fo) fclose(fo);
    free(cpy.pvY); free(cpy.pvU); free(cpy.pvV);
    api.Close(h);
    const double t_closed = now_s() - t0;
    log_capture_end();
    free(au);
    free(data);
    if (check_md5)
        printf("md5: %d pictures checked, %d planes differ\n", (logp.correct + logp.incorrect + 2) / 3, logp.incorrect);
    printf("time until the decoder was closed (everything enqueued has run)= %.3f\n", t_closed);
    printf("frame= %d fps= %.0f time= %.2f video_size= %dx%d\n", nb_frame, t > 0 ? nb_frame / t : 0.0, t, width, height);
    return rc ? rc : logp.incorrect ? 3 : 0;
}
