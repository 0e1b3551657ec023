/*
 * oracle/ref_ps_dump.c -- TEST INFRASTRUCTURE ONLY.  A driver of OUR authorship over the UNMODIFIED
 * pocketsphinx reference (oracle/_ref/libpsref.so, built from /root/reference/pocketsphinx by
 * oracle/Makefile): ms_mgau_init + ps_mgau_frame_eval (= ms_cont_mgau_frame_eval through the
 * ps_mgaufuncs_t vtable, acmod.h:97-110), the continuous scorer behind pocketsphinx's acmod_score.
 *
 *   ref_ps_dump MEAN VAR MIXW SENMGAU(.cont.|.semi.) TOPN AW LOGBASE FEAT.f32 T ACTIVE.u8|all OUTDIR
 *
 * FEAT = T frames of the concatenated streams; ACTIVE = [T][S] flags (turned into pocketsphinx's
 * delta-encoded list, acmod.c:1220-1271) or "all" (compallsen).  Dumps int16 scores, the
 * precomputed float32 determinants / log-domain precisions and the 8-bit mixture weights.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <sphinxbase/ckd_alloc.h>
#include <sphinxbase/cmd_ln.h>
#include <sphinxbase/logmath.h>
#include <sphinxbase/err.h>
#include "ms_mgau.h"
#include "acmod.h"

static const arg_t defn[] = {
    { "-mean", ARG_STRING, NULL, "" }, { "-var", ARG_STRING, NULL, "" }, { "-mixw", ARG_STRING, NULL, "" },
    { "-senmgau", ARG_STRING, NULL, "" }, { "-varfloor", ARG_FLOAT32, "0.0001", "" },
    { "-mixwfloor", ARG_FLOAT32, "0.0000001", "" }, { "-topn", ARG_INT32, "4", "" }, { "-aw", ARG_INT32, "1", "" },
    { NULL, 0, NULL, NULL }
};

static void
dump(const char *dir, const char *name, const char *dt, const void *p, size_t elsz, size_t n, const char *dims)
{
    char path[4096];
    FILE *fp;
    snprintf(path, sizeof path, "%s/%s.%s.%s.bin", dir, name, dt, dims);
    if ((fp = fopen(path, "wb")) == NULL) { perror(path); exit(2); }
    if (n && fwrite(p, elsz, n, fp) != n) { perror("fwrite"); exit(2); }
    fclose(fp);
}

static void *
slurp(const char *path, size_t *nb)
{
    FILE *fp = fopen(path, "rb");
    long sz;
    void *b;
    if (!fp) { perror(path); exit(2); }
    fseek(fp, 0, SEEK_END); sz = ftell(fp); fseek(fp, 0, SEEK_SET);
    b = malloc(sz > 0 ? sz : 1);
    if (fread(b, 1, sz, fp) != (size_t)sz) { perror("fread"); exit(2); }
    fclose(fp);
    if (nb) *nb = sz;
    return b;
}

int
main(int argc, char **argv)
{
    cmd_ln_t *config;
    logmath_t *lmath;
    ps_mgau_t *mg;
    ms_mgau_model_t *msg;
    gauden_t *g;
    senone_t *sn;
    int32 T, S, D = 0, f, t, s, m, d, i, k, c;
    size_t nb;
    float *feat;
    uint8 *active_in, *lst;
    int16 *senscr, *all;
    float *det, *prec;
    uint8 *pdf;
    mfcc_t **fv;
    char dims[128];

    if (argc != 12) { fprintf(stderr, "usage: see the header of oracle/ref_ps_dump.c\n"); return 2; }
    config = cmd_ln_init(NULL, defn, TRUE, "-mean", argv[1], "-var", argv[2], "-mixw", argv[3], "-senmgau", argv[4],
                         "-topn", argv[5], "-aw", argv[6], NULL);
    lmath = logmath_init(atof(argv[7]), 0, FALSE);      /* acmod.c: logmath_init(-logbase, 0, FALSE) */
    mg = ms_mgau_init(config, lmath, NULL);
    msg = (ms_mgau_model_t *)mg;
    g = ms_mgau_gauden(msg); sn = ms_mgau_senone(msg);
    S = sn->n_sen; T = atoi(argv[9]);
    for (f = 0; f < g->n_feat; f++) D += g->featlen[f];
    feat = slurp(argv[8], &nb);
    if ((size_t)T * D * 4 > nb) { fprintf(stderr, "feature file too short\n"); return 2; }
    active_in = strcmp(argv[10], "all") ? slurp(argv[10], NULL) : NULL;
    fv = ckd_calloc(g->n_feat, sizeof(*fv));
    senscr = malloc(sizeof(int16) * S); all = calloc((size_t)T * S, sizeof(int16)); lst = malloc(S + 8);
    /* the model as precomputed, flattened in file order [m][f][d][featlen] */
    det = malloc(sizeof(float) * (size_t)g->n_mgau * g->n_feat * g->n_density);
    prec = malloc(sizeof(float) * (size_t)g->n_mgau * g->n_density * D);
    for (m = 0, k = 0, c = 0; m < g->n_mgau; m++)
        for (f = 0; f < g->n_feat; f++)
            for (d = 0; d < g->n_density; d++) {
                det[k++] = g->det[m][f][d];
                for (i = 0; i < g->featlen[f]; i++) prec[c++] = g->var[m][f][d][i];
            }
    pdf = malloc((size_t)S * sn->n_feat * sn->n_cw);
    for (s = 0, k = 0; s < S; s++)
        for (f = 0; f < (int32)sn->n_feat; f++)
            for (c = 0; c < (int32)sn->n_cw; c++)
                pdf[k++] = (sn->n_gauden > 1) ? sn->pdf[s][f][c] : sn->pdf[f][c][s];
    for (t = 0; t < T; t++) {
        int32 off = 0, n = 0, last = 0;
        for (f = 0; f < g->n_feat; f++) { fv[f] = feat + (size_t)t * D + off; off += g->featlen[f]; }
        memset(senscr, 0, sizeof(int16) * S);
        if (active_in) {
            /* delta-encoded ascending senone list; gaps over 255 are bridged with 255-steps, which the
             * scorer sees as extra active senones (acmod_flags2list does exactly this) */
            for (s = 0; s < S; s++)
                if (active_in[(size_t)t * S + s]) {
                    int32 delta = s - last;
                    while (delta > 255) { lst[n++] = 255; delta -= 255; }
                    lst[n++] = (uint8)delta; last = s;
                }
            ps_mgau_frame_eval(mg, senscr, lst, n, fv, t, 0);
        }
        else
            ps_mgau_frame_eval(mg, senscr, NULL, 0, fv, t, 1);
        memcpy(all + (size_t)t * S, senscr, sizeof(int16) * S);
    }
    snprintf(dims, sizeof dims, "%dx%dx%d", g->n_mgau, g->n_feat, g->n_density);
    dump(argv[11], "det", "f32", det, 4, (size_t)g->n_mgau * g->n_feat * g->n_density, dims);
    snprintf(dims, sizeof dims, "%d", g->n_mgau * g->n_density * D);
    dump(argv[11], "prec", "f32", prec, 4, (size_t)g->n_mgau * g->n_density * D, dims);
    snprintf(dims, sizeof dims, "%dx%dx%d", S, sn->n_feat, sn->n_cw);
    dump(argv[11], "pdf", "u8", pdf, 1, (size_t)S * sn->n_feat * sn->n_cw, dims);
    snprintf(dims, sizeof dims, "%dx%d", T, S);
    dump(argv[11], "senscr", "i16", all, 2, (size_t)T * S, dims);
    return 0;
}
