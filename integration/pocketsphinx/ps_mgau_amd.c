/*
 * oracle/ref_ps_shim.c -- TEST INFRASTRUCTURE: the SECONDARY drop-in boundary demonstrated.
 *
 * The unmodified pocketsphinx decoder (oracle/_ref/libpsref.so) with the acoustic scorer behind its
 * ps_mgaufuncs_t vtable (pocketsphinx/src/libpocketsphinx/acmod.h:97-115) replaced by
 * libcmusphinx_amd through the C ABI: after ps_init() the acmod's `mgau` object is swapped for one
 * whose vtable forwards frame_eval to s3a_ps_ms_cont_mgau_frame_eval.  Everything else -- feature
 * computation, senone activation lists, fwdtree/fwdflat/bestpath search, LM -- stays pocketsphinx's.
 *
 *   ref_ps_shim ref|gpu MDEF MEAN VAR MIXW TMAT DICT FDICT LM CTL CEPDIR OUT
 *
 * decodes the cepstrum files of CTL and writes "<hyp> (<uttid> <score>)" lines; tests diff the two
 * modes (tests/test_gpu_dropin.py).  This mirrors what INTEGRATION.md tells a pocketsphinx maintainer
 * to add to acmod_init_am.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sphinxbase/ckd_alloc.h>
#include <sphinxbase/cmd_ln.h>
#include <sphinxbase/err.h>
#include <sphinxbase/feat.h>
#include "pocketsphinx.h"
#include "pocketsphinx_internal.h"
#include "cmusphinx_amd.h"

typedef struct {
    ps_mgau_t base;             /* must be first: {vt, frame_idx} */
    s3a_ps_mgau_t *h;
    feat_t *fcb;
    float32 *cat;
    long calls;
} amd_mgau_t;

static int
amd_frame_eval(ps_mgau_t *mg, int16 *senscr, uint8 *senone_active, int32 n_senone_active, mfcc_t **feat,
               int32 frame, int32 compallsen)
{
    amd_mgau_t *a = (amd_mgau_t *)mg;
    float32 *x = feat[0];
    if (feat_dimension1(a->fcb) > 1) {          /* streams are separate rows: concatenate */
        int32 f, o;
        for (f = 0, o = 0; f < feat_dimension1(a->fcb); o += feat_dimension2(a->fcb, f), f++)
            memcpy(a->cat + o, feat[f], feat_dimension2(a->fcb, f) * sizeof(float32));
        x = a->cat;
    }
    if (s3a_ps_ms_cont_mgau_frame_eval(a->h, senscr, senone_active, n_senone_active, x, frame, compallsen) != S3A_OK)
        E_FATAL("ps shim: %s\n", s3a_last_error());
    a->calls++;
    return 0;
}

static int amd_transform(ps_mgau_t *mg, ps_mllr_t *mllr) { (void)mg; (void)mllr; E_ERROR("ps shim: MLLR not supported\n"); return -1; }
static void amd_free(ps_mgau_t *mg) { amd_mgau_t *a = (amd_mgau_t *)mg; s3a_ps_ms_mgau_free(a->h); ckd_free(a->cat); ckd_free(a); }
static ps_mgaufuncs_t amd_funcs = { "cmusphinx_amd", amd_frame_eval, amd_transform, amd_free };

static mfcc_t **
read_mfc(const char *path, int32 *nfr)
{
    FILE *fp = fopen(path, "rb");
    int32 n, i, swap = 0;
    long sz;
    float32 *buf;
    mfcc_t **cep;
    if (!fp) E_FATAL("cannot open %s\n", path);
    fseek(fp, 0, SEEK_END); sz = ftell(fp); fseek(fp, 0, SEEK_SET);
    if (fread(&n, 4, 1, fp) != 1) E_FATAL("%s: empty\n", path);
    if ((long)n * 4 + 4 != sz) { n = (int32)__builtin_bswap32((uint32)n); swap = 1; }
    if ((long)n * 4 + 4 != sz) E_FATAL("%s: header does not match the file size\n", path);
    buf = ckd_calloc(n, 4);
    if (fread(buf, 4, n, fp) != (size_t)n) E_FATAL("%s: short read\n", path);
    fclose(fp);
    if (swap) for (i = 0; i < n; i++) { uint32 *w = (uint32 *)&buf[i]; *w = __builtin_bswap32(*w); }
    *nfr = n / 13;
    cep = (mfcc_t **)ckd_calloc_2d(*nfr, 13, sizeof(mfcc_t));
    memcpy(cep[0], buf, (size_t)(*nfr) * 13 * 4);
    ckd_free(buf);
    return cep;
}

int
main(int argc, char **argv)
{
    cmd_ln_t *config;
    ps_decoder_t *ps;
    amd_mgau_t *a = NULL;
    FILE *ctl, *out;
    char line[1024], path[4096];
    int gpu;
    if (argc != 13) { fprintf(stderr, "usage: see the header of oracle/ref_ps_shim.c\n"); return 2; }
    gpu = strcmp(argv[1], "gpu") == 0;
    /* PS_SENLOGDIR: pocketsphinx's own -senlogdir (acmod_write_scores writes <dir>/<uttid>.sen) */
    config = cmd_ln_init(NULL, ps_args(), TRUE, "-mdef", argv[2], "-mean", argv[3], "-var", argv[4], "-mixw", argv[5],
                         "-tmat", argv[6], "-dict", argv[7], "-fdict", argv[8], "-lm", argv[9], "-senmgau", ".cont.",
                         "-topn", "4", getenv("PS_SENLOGDIR") ? "-senlogdir" : NULL, getenv("PS_SENLOGDIR"), NULL);
    if ((ps = ps_init(config)) == NULL) E_FATAL("ps_init failed\n");
    if (strcmp(ps->acmod->mgau->vt->name, "ms") != 0) E_FATAL("ps shim: expected the multi-stream scorer\n");
    if (gpu) {
        a = ckd_calloc(1, sizeof(*a));
        a->base.vt = &amd_funcs;
        a->fcb = ps->acmod->fcb;
        /* the same values ms_mgau_init reads from the configuration (ms_mgau.c:88-103) */
        a->h = s3a_ps_ms_mgau_init(cmd_ln_str_r(config, "-mean"), cmd_ln_str_r(config, "-var"),
                                   cmd_ln_float32_r(config, "-varfloor"), cmd_ln_str_r(config, "-mixw"),
                                   cmd_ln_float32_r(config, "-mixwfloor"), cmd_ln_str_r(config, "-senmgau"),
                                   cmd_ln_int32_r(config, "-topn"), cmd_ln_int32_r(config, "-aw"),
                                   cmd_ln_float32_r(config, "-logbase"));
        if (!a->h) E_FATAL("ps shim: %s\n", s3a_last_error());
        a->cat = ckd_calloc(s3a_ps_ms_mgau_veclen(a->h), sizeof(float32));
        ps_mgau_free(ps->acmod->mgau);
        ps->acmod->mgau = (ps_mgau_t *)a;
    }
    if ((ctl = fopen(argv[10], "r")) == NULL || (out = fopen(argv[12], "w")) == NULL) E_FATAL("ctl/out\n");
    while (fgets(line, sizeof line, ctl)) {
        char uttid[1024];
        const char *hyp, *id;
        int32 nfr, score;
        mfcc_t **cep;
        if (sscanf(line, "%1023s", uttid) != 1) continue;
        snprintf(path, sizeof path, "%s/%s.mfc", argv[11], uttid);
        cep = read_mfc(path, &nfr);
        ps_start_utt(ps, uttid);
        ps_process_cep(ps, cep, nfr, FALSE, TRUE);
        ps_end_utt(ps);
        hyp = ps_get_hyp(ps, &score, &id);
        fprintf(out, "%s (%s %d)\n", hyp ? hyp : "", uttid, score);
        ckd_free_2d((void **)cep);
    }
    fclose(out); fclose(ctl);
    if (a) E_INFO("ps shim: %ld frame_eval calls served by %s\n", a->calls, s3a_version());
    ps_free(ps);
    return 0;
}
