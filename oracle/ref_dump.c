/*
 * oracle/ref_dump.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A driver of OUR authorship that links the UNMODIFIED reference
 * (oracle/_ref/libs3ref.so, built from /root/reference by oracle/Makefile) and
 * calls the reference's own entry points for the hot path, dumping raw arrays
 * that tests/golden/make_golden.py packs into committed fixtures and that the
 * not-gpu tests diff against the oracle restatement (oracle/s3o_*.c).
 *
 * Every array goes to  OUTDIR/<name>.<dtype>.<d0>[x<d1>...].bin  (little endian).
 *
 * Sub-commands:
 *   logmath BASE SHIFT OUTDIR
 *   mgau    MEAN VAR MIXW VARFLOOR MIXWFLOOR LOGBASE FEAT.f32 T OUTDIR
 *   frame   CD2CISEN.i16 NCISEN MEAN VAR MIXW LOGBASE FEAT.f32 T ACTIVE.u8|all
 *           CIPBEAM DS TIGHTEN MAXCD OUTDIR
 *   tmat    TMATFILE TPFLOOR LOGBASE SHIFT OUTDIR
 *   feat    MFCFILE [CMN VARNORM AGC] OUTDIR   (1s_c_d_dd; default -cmn current, -agc none, -varnorm no)
 *   bench_mgau MEAN VAR MIXW LOGBASE FEAT.f32 T   (times approx_cont_mgau_frame_eval with every
 *           senone active; prints "frames T seconds S" -- the CPU baseline of bench.py)
 *   hmm     NEMIT TP.i32 NTMAT SSEQ.i16 NSSEQ SENSCR.i32 NSEN T SPEC.i32 NHMM ENTER.i32 OUTDIR
 *   fe      RAWFILE(int16 LE) OUTDIR [fe options, e.g. -samprate 11025 -nfilt 36 ...]
 *           (fe_init_auto_r + fe_process_utt + fe_end_utt: the MFCC front end; also times it)
 *   ms      MEAN VAR MIXW SENMGAU(.s3cont.|.semi.) TOPN LOGBASE FEAT.f32 T ACTIVE.u8|all OUTDIR
 *           (ms_mgau_init + ms_cont_mgau_frame_eval: the -senmgau .s3cont./.semi. scorer)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <math.h>

#include <sphinxbase/ckd_alloc.h>
#include <sphinxbase/logmath.h>
#include <sphinxbase/err.h>
#include "s3types.h"
#include "logs3.h"
#include "cont_mgau.h"
#include "approx_cont_mgau.h"
#include "fast_algo_struct.h"
#include "ascr.h"
#include "mdef.h"
#include "tmat.h"
#include "hmm.h"
#include "ms_mgau.h"
#include <sphinxbase/feat.h>
#include <sphinxbase/cmn.h>
#include <sphinxbase/agc.h>
#include <sphinxbase/fe.h>
#include <sphinxbase/cmd_ln.h>
#include <time.h>

static void
dump(const char *outdir, const char *name, const char *dtype, const void *p,
     size_t elsz, int nd, ...)
{
    char path[4096], dims[256] = "";
    size_t n = 1;
    va_list ap;
    FILE *fp;
    int i;

    va_start(ap, nd);
    for (i = 0; i < nd; i++) {
        long d = va_arg(ap, long);
        char t[32];
        snprintf(t, sizeof t, i ? "x%ld" : "%ld", d);
        strcat(dims, t);
        n *= (size_t)d;
    }
    va_end(ap);
    snprintf(path, sizeof path, "%s/%s.%s.%s.bin", outdir, name, dtype, dims);
    if ((fp = fopen(path, "wb")) == NULL) { perror(path); exit(2); }
    if (n && fwrite(p, elsz, n, fp) != n) { perror("fwrite"); exit(2); }
    fclose(fp);
}

static void *
slurp(const char *path, size_t *nbytes)
{
    FILE *fp = fopen(path, "rb");
    long sz;
    void *buf;
    if (!fp) { perror(path); exit(2); }
    fseek(fp, 0, SEEK_END);
    sz = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    buf = malloc(sz ? sz : 1);
    if (sz && fread(buf, 1, sz, fp) != (size_t)sz) { perror("fread"); exit(2); }
    fclose(fp);
    if (nbytes) *nbytes = sz;
    return buf;
}

static int
cmd_logmath(int argc, char **argv)
{
    double base = atof(argv[0]);
    int shift = atoi(argv[1]);
    const char *out = argv[2];
    logmath_t *lm = logmath_init(base, shift, 1);
    uint32 size, width, sh, i;
    int32 *tab, ka[16];
    double kd[4];

    logmath_get_table_shape(lm, &size, &width, &sh);
    tab = malloc(sizeof(int32) * size);
    /* recover the table through the public API: add(0, -d) == table[d] */
    for (i = 0; i < size; i++)
        tab[i] = logmath_add(lm, 0, -(int32)i);
    dump(out, "table", "i32", tab, 4, 1, (long)size);
    ka[0] = size; ka[1] = width; ka[2] = sh;
    ka[3] = logmath_get_zero(lm);
    ka[4] = logmath_log(lm, 1e-150);
    ka[5] = logmath_log(lm, 42.0);
    ka[6] = logmath_log(lm, 1e-48);
    ka[7] = logmath_add(lm, logmath_log(lm, 1e-48), logmath_log(lm, 5e-48));
    ka[8] = logmath_add(lm, logmath_log(lm, 1e-48), logmath_log(lm, 42.0));
    ka[9] = logmath_log10_to_log(lm, -7.0);
    ka[10] = logmath_ln_to_log(lm, -123.456);
    ka[11] = logs3(lm, 1e-80);
    ka[12] = logs3(lm, 0.5);
    ka[13] = logmath_add(lm, S3_LOGPROB_ZERO, -12345);
    ka[14] = logmath_add(lm, -12345, S3_LOGPROB_ZERO);
    ka[15] = logmath_add(lm, -100, -100 - (int32)size);
    dump(out, "known", "i32", ka, 4, 1, 16L);
    kd[0] = logmath_log_to_ln(lm, S3_LOGPROB_ZERO);
    kd[1] = logmath_exp(lm, -5000);
    kd[2] = logmath_log_to_ln(lm, -79150);
    kd[3] = logmath_get_base(lm);
    dump(out, "knownf", "f64", kd, 8, 1, 4L);
    return 0;
}

static mgau_model_t *
load_mgau(const char *mean, const char *var, const char *mixw, double varfloor,
          double mixwfloor, logmath_t *lm)
{
    return mgau_init(mean, var, varfloor, mixw, mixwfloor, 1, ".cont.",
                     MIX_INT_FLOAT_COMP, lm);
}

static void
dump_model(const char *out, mgau_model_t *g)
{
    int32 S = g->n_mgau, C = g->max_comp, D = g->veclen, m, c;
    float *mean = calloc((size_t)S * C * D, 4), *var = calloc((size_t)S * C * D, 4);
    float *lrd = calloc((size_t)S * C, 4);
    int32 *mixw = calloc((size_t)S * C, 4), *ncomp = calloc(S, 4);
    for (m = 0; m < S; m++) {
        ncomp[m] = g->mgau[m].n_comp;
        for (c = 0; c < g->mgau[m].n_comp; c++) {
            memcpy(mean + ((size_t)m * C + c) * D, g->mgau[m].mean[c], 4 * D);
            memcpy(var + ((size_t)m * C + c) * D, g->mgau[m].var[c], 4 * D);
            lrd[m * C + c] = g->mgau[m].lrd[c];
            mixw[m * C + c] = g->mgau[m].mixw[c];
        }
    }
    dump(out, "n_comp", "i32", ncomp, 4, 1, (long)S);
    dump(out, "mean", "f32", mean, 4, 3, (long)S, (long)C, (long)D);
    dump(out, "prec", "f32", var, 4, 3, (long)S, (long)C, (long)D);
    dump(out, "lrd", "f32", lrd, 4, 2, (long)S, (long)C);
    dump(out, "mixw", "i32", mixw, 4, 2, (long)S, (long)C);
    dump(out, "distfloor", "f64", &g->distfloor, 8, 1, 1L);
    free(mean); free(var); free(lrd); free(mixw); free(ncomp);
}

static int
cmd_mgau(int argc, char **argv)
{
    logmath_t *lm = logs3_init(atof(argv[5]), 0, 1);
    mgau_model_t *g = load_mgau(argv[0], argv[1], argv[2], atof(argv[3]), atof(argv[4]), lm);
    size_t nb;
    float *feat = slurp(argv[6], &nb);
    int32 T = atoi(argv[7]), S = g->n_mgau, D = g->veclen, t, s;
    const char *out = argv[8];
    int32 *scr = malloc(sizeof(int32) * (size_t)T * S);
    int32 *bidx = malloc(sizeof(int32) * (size_t)T * S);
    int32 *bscr = malloc(sizeof(int32) * (size_t)T * S);

    if (nb != (size_t)T * D * 4) { fprintf(stderr, "feat size mismatch\n"); return 2; }
    dump_model(out, g);
    for (t = 0; t < T; t++)
        for (s = 0; s < S; s++) {
            scr[(size_t)t * S + s] = mgau_eval(g, s, NULL, feat + (size_t)t * D, t, 1);
            bidx[(size_t)t * S + s] = g->mgau[s].bstidx;
            bscr[(size_t)t * S + s] = g->mgau[s].bstscr;
        }
    dump(out, "score", "i32", scr, 4, 2, (long)T, (long)S);
    dump(out, "bstidx", "i32", bidx, 4, 2, (long)T, (long)S);
    dump(out, "bstscr", "i32", bscr, 4, 2, (long)T, (long)S);
    return 0;
}

static int
cmd_frame(int argc, char **argv)
{
    size_t nb;
    s3senid_t *cd2cisen = slurp(argv[0], &nb);
    int32 S = (int32)(nb / sizeof(s3senid_t));
    int32 n_ci_sen = atoi(argv[1]);
    logmath_t *lm = logs3_init(atof(argv[5]), 0, 1);
    mgau_model_t *g = load_mgau(argv[2], argv[3], argv[4], 0.0001, 0.0000001, lm);
    float *feat = slurp(argv[6], &nb);
    int32 T = atoi(argv[7]), D = g->veclen, t, s;
    uint8 *active_in = strcmp(argv[8], "all") ? slurp(argv[8], NULL) : NULL;
    double cipbeam = atof(argv[9]);
    int32 ds = atoi(argv[10]);
    float tighten = (float)atof(argv[11]);
    int32 maxcd = atoi(argv[12]);
    const char *out = argv[13];
    fast_gmm_t *fg = fast_gmm_init(ds, 0, 0, 1, 0, 3.2e-5 /* unused */, cipbeam, tighten,
                                   maxcd, n_ci_sen, lm);
    ascr_t *a = ascr_init(S, 0, 1, 0, 1, n_ci_sen);
    mdef_t md;          /* only the fields approx_cont_mgau_* read are filled */
    ptmr_t tm;
    int32 *senscr = malloc(sizeof(int32) * (size_t)T * S);
    int32 *bidx = malloc(sizeof(int32) * (size_t)T * S);
    int32 *upd = malloc(sizeof(int32) * (size_t)T * S);
    uint8 *act = malloc((size_t)T * S);
    int32 *best = malloc(sizeof(int32) * T), *cibest = malloc(sizeof(int32) * T);
    int32 *cnt = malloc(sizeof(int32) * 4 * T);
    int32 *ciscr = malloc(sizeof(int32) * (size_t)T * (n_ci_sen > 0 ? n_ci_sen : 1));

    if (S != g->n_mgau) { fprintf(stderr, "cd2cisen size mismatch\n"); return 2; }
    memset(&md, 0, sizeof md);
    md.n_sen = S;
    md.n_ci_sen = n_ci_sen;
    md.cd2cisen = cd2cisen;
    ptmr_init(&tm);

    /* as srch_TST_begin does: srch_time_switch_tree.c:485-490 */
    for (s = 0; s < S; s++) {
        g->mgau[s].bstidx = NO_BSTIDX;
        g->mgau[s].updatetime = NOT_UPDATED;
    }
    for (t = 0; t < T; t++) {
        float *fv = feat + (size_t)t * D;
        /* gmm_compute_lv1: gmm_wrap.c:170-211 */
        approx_cont_mgau_ci_eval(NULL, NULL, g, fg, &md, fv, a->cache_ci_senscr[0],
                                 &a->cache_best_list[0], t, lm);
        memcpy(ciscr + (size_t)t * n_ci_sen, a->cache_ci_senscr[0], 4 * n_ci_sen);
        cibest[t] = a->cache_best_list[0];
        cnt[4 * t + 2] = g->frm_ci_sen_eval;
        cnt[4 * t + 3] = g->frm_ci_gau_eval;
        /* select_active_gmm stand-in */
        if (active_in)
            memcpy(a->sen_active, active_in + (size_t)t * S, S);
        else
            memset(a->sen_active, 1, S);
        /* gmm_compute_lv2: gmm_wrap.c:108-167 */
        best[t] = approx_cont_mgau_frame_eval(&md, NULL, NULL, g, fg, a, fv, t,
                                              a->cache_ci_senscr[0], &tm, lm);
        cnt[4 * t + 0] = g->frm_sen_eval;
        cnt[4 * t + 1] = g->frm_gau_eval;
        memcpy(senscr + (size_t)t * S, a->senscr, 4 * S);
        memcpy(act + (size_t)t * S, a->sen_active, S);
        for (s = 0; s < S; s++) {
            bidx[(size_t)t * S + s] = g->mgau[s].bstidx;
            upd[(size_t)t * S + s] = g->mgau[s].updatetime;
        }
    }
    dump(out, "senscr", "i32", senscr, 4, 2, (long)T, (long)S);
    dump(out, "sen_active_out", "u8", act, 1, 2, (long)T, (long)S);
    dump(out, "best", "i32", best, 4, 1, (long)T);
    dump(out, "ci_best", "i32", cibest, 4, 1, (long)T);
    dump(out, "ci_senscr", "i32", ciscr, 4, 2, (long)T, (long)n_ci_sen);
    dump(out, "bstidx", "i32", bidx, 4, 2, (long)T, (long)S);
    dump(out, "updatetime", "i32", upd, 4, 2, (long)T, (long)S);
    dump(out, "counts", "i32", cnt, 4, 2, (long)T, 4L);
    {
        int32 p[2];
        p[0] = fg->gmms->ci_pbeam;
        p[1] = fg->gmms->dyn_ci_pbeam;
        dump(out, "beams", "i32", p, 4, 1, 2L);
    }
    return 0;
}

static int
cmd_tmat(int argc, char **argv)
{
    logmath_t *lm = logs3_init(atof(argv[2]), 0, 1);
    tmat_t *t;
    const char *out = argv[4];
    int32 ns, i, j, k, *flat;
    if (atoi(argv[3]) != 0) {
        logmath_free(lm);
        lm = logmath_init(atof(argv[2]), atoi(argv[3]), 1);
    }
    t = tmat_init(argv[0], atof(argv[1]), 0, lm);
    ns = t->n_state;
    flat = malloc(sizeof(int32) * t->n_tmat * ns * (ns + 1));
    for (i = 0; i < t->n_tmat; i++)
        for (j = 0; j < ns; j++)
            for (k = 0; k <= ns; k++)
                flat[(i * ns + j) * (ns + 1) + k] = t->tp[i][j][k];
    dump(out, "tp", "i32", flat, 4, 3, (long)t->n_tmat, (long)ns, (long)(ns + 1));
    return 0;
}

/*
 * hmm: run hmm_vit_eval over T frames for NHMM independent HMMs.
 *   SPEC.i32  [NHMM][3]   = {mpx, ssid, tmatid}
 *   ENTER.i32 [T][NHMM][2] = {score, histid}; score == INT32_MIN means "no entry this frame"
 * After every frame dumps score[5], history[5], out, bestscore, mpx ssids.
 */
static int
cmd_hmm(int argc, char **argv)
{
    int32 ne = atoi(argv[0]);
    int32 *tpflat = slurp(argv[1], NULL);
    int32 ntmat = atoi(argv[2]);
    s3senid_t *sseqflat = slurp(argv[3], NULL);
    int32 nsseq = atoi(argv[4]);
    int32 *senscr = slurp(argv[5], NULL);
    int32 nsen = atoi(argv[6]), T = atoi(argv[7]);
    int32 *spec = slurp(argv[8], NULL);
    int32 nh = atoi(argv[9]);
    int32 *enter = slurp(argv[10], NULL);
    const char *out = argv[11];
    int32 ***tp = (int32 ***)ckd_calloc_3d(ntmat, ne, ne + 1, sizeof(int32));
    s3senid_t **sseq = (s3senid_t **)ckd_calloc_2d(nsseq, ne, sizeof(s3senid_t));
    hmm_context_t *ctx;
    hmm_t *h = calloc(nh, sizeof(hmm_t));
    int32 i, j, k, t;
    /* per frame per hmm: 5 scores, out, best, 5 ssid = 12 ; histories 6 (i64) */
    int32 *o32 = malloc(sizeof(int32) * (size_t)T * nh * 12);
    long long *o64 = malloc(sizeof(long long) * (size_t)T * nh * 6);
    int32 *ret = malloc(sizeof(int32) * (size_t)T * nh);

    for (i = 0; i < ntmat; i++)
        for (j = 0; j < ne; j++)
            for (k = 0; k <= ne; k++)
                tp[i][j][k] = tpflat[(i * ne + j) * (ne + 1) + k];
    for (i = 0; i < nsseq; i++)
        for (j = 0; j < ne; j++)
            sseq[i][j] = sseqflat[i * ne + j];
    ctx = hmm_context_init(ne, tp, senscr, sseq);
    for (i = 0; i < nh; i++)
        hmm_init(ctx, &h[i], spec[3 * i], spec[3 * i + 1], (s3tmatid_t)spec[3 * i + 2]);

    for (t = 0; t < T; t++) {
        hmm_context_set_senscore(ctx, senscr + (size_t)t * nsen);
        for (i = 0; i < nh; i++) {
            int32 *e = enter + ((size_t)t * nh + i) * 2;
            int32 *p = o32 + ((size_t)t * nh + i) * 12;
            long long *q = o64 + ((size_t)t * nh + i) * 6;
            if (e[0] != (int32)0x80000000)
                hmm_enter(&h[i], e[0], e[1], t);
            ret[(size_t)t * nh + i] = hmm_vit_eval(&h[i]);
            for (j = 0; j < 5; j++) {
                p[j] = (j < ne) ? hmm_score(&h[i], j) : 0;
                q[j] = (j < ne) ? (long long)h[i].state[j].history.id : 0;
                p[7 + j] = (h[i].mpx && j < ne) ? h[i].s.mpx_ssid[j] : -2;
            }
            p[5] = hmm_out_score(&h[i]);
            p[6] = hmm_bestscore(&h[i]);
            q[5] = (long long)h[i].out.history.id;
        }
    }
    dump(out, "state", "i32", o32, 4, 3, (long)T, (long)nh, 12L);
    dump(out, "hist", "i64", o64, 8, 3, (long)T, (long)nh, 6L);
    dump(out, "ret", "i32", ret, 4, 2, (long)T, (long)nh);
    return 0;
}

/*
 * CPU baseline: the reference's own per-frame scoring path
 * (approx_cont_mgau_ci_eval + approx_cont_mgau_frame_eval, every senone active,
 * default -ci_pbeam 1e-80) over T frames; model load excluded from the timing.
 */
#include <time.h>
static int
cmd_bench_mgau(int argc, char **argv)
{
    logmath_t *lm = logs3_init(atof(argv[3]), 0, 1);
    mgau_model_t *g = load_mgau(argv[0], argv[1], argv[2], 0.0001, 0.0000001, lm);
    size_t nb;
    float *feat = slurp(argv[4], &nb);
    int32 T = atoi(argv[5]), S = g->n_mgau, D = g->veclen, t, s;
    int32 n_ci = (S >= 144) ? 144 : S / 4;
    fast_gmm_t *fg = fast_gmm_init(1, 0, 0, 1, 0, 3.2e-5, 1e-80, 0.5f, 100000, n_ci, lm);
    ascr_t *a = ascr_init(S, 0, 1, 0, 1, n_ci);
    s3senid_t *c2c = calloc(S, sizeof(s3senid_t));
    mdef_t md;
    ptmr_t tm;
    struct timespec t0, t1;
    long long chk = 0;

    if (nb < (size_t)T * D * 4) { fprintf(stderr, "feat too short\n"); return 2; }
    for (s = 0; s < S; s++) c2c[s] = (s < n_ci) ? s : (s % n_ci);
    memset(&md, 0, sizeof md);
    md.n_sen = S; md.n_ci_sen = n_ci; md.cd2cisen = c2c;
    ptmr_init(&tm);
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (t = 0; t < T; t++) {
        float *fv = feat + (size_t)t * D;
        approx_cont_mgau_ci_eval(NULL, NULL, g, fg, &md, fv, a->cache_ci_senscr[0],
                                 &a->cache_best_list[0], t, lm);
        memset(a->sen_active, 1, S);
        chk += approx_cont_mgau_frame_eval(&md, NULL, NULL, g, fg, a, fv, t,
                                           a->cache_ci_senscr[0], &tm, lm);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    printf("frames %d seconds %.6f checksum %lld\n", T,
           (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec), chk);
    return 0;
}

/* feat_s2mfc2feat exactly as utt_decode calls it (libAPI/utt.c:234) */
static int
cmd_feat(int argc, char **argv)
{
    /* optional: CMN(none|current) VARNORM(0|1) AGC(none|max) */
    feat_t *fcb = feat_init("1s_c_d_dd", argc >= 5 ? cmn_type_from_str(argv[2]) : CMN_CURRENT,
                            argc >= 5 ? atoi(argv[3]) : 0, argc >= 5 ? agc_type_from_str(argv[4]) : AGC_NONE, 0, 13);
    mfcc_t ***feat = feat_array_alloc(fcb, S3_MAX_FRAMES);
    int32 nfr = feat_s2mfc2feat(fcb, argv[0], NULL, "", 0, -1, feat, S3_MAX_FRAMES);
    int32 D = feat_stream_len(fcb, 0), t;
    float *flat;
    if (nfr <= 0) { fprintf(stderr, "feat_s2mfc2feat failed\n"); return 2; }
    flat = malloc(sizeof(float) * (size_t)nfr * D);
    for (t = 0; t < nfr; t++)
        memcpy(flat + (size_t)t * D, feat[t][0], sizeof(float) * D);
    dump(argv[1], "feat", "f32", flat, 4, 2, (long)nfr, (long)D);
    return 0;
}

static int
cmd_ms(int argc, char **argv)
{
    logmath_t *lm = logs3_init(atof(argv[5]), 0, 1);
    int32 topn_arg = atoi(argv[4]);
    ms_mgau_model_t *msg = ms_mgau_init(argv[0], argv[1], 0.0001, argv[2], 0.0000001, TRUE, argv[3], NULL,
                                        topn_arg, lm, NULL);
    gauden_t *g = ms_mgau_gauden(msg);
    senone_t *sn = ms_mgau_senone(msg);
    int32 topn = ms_mgau_topn(msg), S = sn->n_sen, T = atoi(argv[7]), D = 0, f, t, s, m, d, c, k;
    size_t nb;
    float *feat = slurp(argv[6], &nb);
    uint8 *active_in = strcmp(argv[8], "all") ? slurp(argv[8], NULL) : NULL;
    const char *out = argv[9];
    ascr_t *a = ascr_init(S, 0, 1, 0, 1, 0);
    mdef_t md;
    float32 **fv = ckd_calloc(g->n_feat, sizeof(float32 *));
    int32 *senscr = malloc(sizeof(int32) * (size_t)T * S), *best = malloc(sizeof(int32) * T);
    size_t nd = (size_t)g->n_mgau * g->n_feat * topn;
    int32 *dist = malloc(sizeof(int32) * T * nd), *dist_id = malloc(sizeof(int32) * T * nd);
    uint8 *act = malloc((size_t)T * S);
    float *det, *prec;
    int32 *pdf, *flen = malloc(sizeof(int32) * g->n_feat), *mg = malloc(sizeof(int32) * S);

    memset(&md, 0, sizeof md);
    md.n_sen = S;
    for (f = 0; f < g->n_feat; f++) { flen[f] = g->featlen[f]; D += g->featlen[f]; }
    if ((size_t)T * D * 4 > nb) { fprintf(stderr, "feature file too short\n"); return 2; }
    /* the precomputed model, flattened in file order [m][f][d][featlen] */
    det = malloc(sizeof(float) * (size_t)g->n_mgau * g->n_feat * g->n_density);
    prec = malloc(sizeof(float) * (size_t)g->n_mgau * g->n_density * D);
    for (m = 0, k = 0, c = 0; m < g->n_mgau; m++)
        for (f = 0; f < g->n_feat; f++)
            for (d = 0; d < g->n_density; d++) {
                int32 i;
                det[k++] = g->det[m][f][d];
                for (i = 0; i < g->featlen[f]; i++) prec[c++] = g->var[m][f][d][i];
            }
    pdf = malloc(sizeof(int32) * (size_t)S * sn->n_feat * sn->n_cw);
    for (s = 0, k = 0; s < S; s++) {
        mg[s] = sn->mgau[s];
        for (f = 0; f < sn->n_feat; f++)
            for (c = 0; c < sn->n_cw; c++)
                pdf[k++] = (sn->n_gauden > 1) ? (int32)sn->pdf[s][f][c] : (int32)sn->pdf[f][c][s];
    }
    memset(dist, 0, sizeof(int32) * T * nd); memset(dist_id, 0, sizeof(int32) * T * nd);
    for (t = 0; t < T; t++) {
        int32 off = 0;
        for (f = 0; f < g->n_feat; f++) { fv[f] = feat + (size_t)t * D + off; off += g->featlen[f]; }
        if (active_in) memcpy(a->sen_active, active_in + (size_t)t * S, S);
        else memset(a->sen_active, 1, S);
        best[t] = ms_cont_mgau_frame_eval(a, msg, &md, fv, t);
        memcpy(senscr + (size_t)t * S, a->senscr, 4 * S);
        memcpy(act + (size_t)t * S, a->sen_active, S);
        for (m = 0, k = 0; m < g->n_mgau; m++)
            for (f = 0; f < g->n_feat; f++)
                for (d = 0; d < topn; d++, k++)
                    if (msg->mgau_active[m]) {
                        dist[t * nd + k] = msg->dist[m][f][d].dist;
                        dist_id[t * nd + k] = msg->dist[m][f][d].id;
                    }
    }
    dump(out, "det", "f32", det, 4, 3, (long)g->n_mgau, (long)g->n_feat, (long)g->n_density);
    dump(out, "prec", "f32", prec, 4, 1, (long)((size_t)g->n_mgau * g->n_density * D));
    dump(out, "pdf", "i32", pdf, 4, 3, (long)S, (long)sn->n_feat, (long)sn->n_cw);
    dump(out, "mgau", "i32", mg, 4, 1, (long)S);
    dump(out, "featlen", "i32", flen, 4, 1, (long)g->n_feat);
    dump(out, "senscr", "i32", senscr, 4, 2, (long)T, (long)S);
    dump(out, "sen_active_out", "u8", act, 1, 2, (long)T, (long)S);
    dump(out, "best", "i32", best, 4, 1, (long)T);
    dump(out, "dist", "i32", dist, 4, 4, (long)T, (long)g->n_mgau, (long)g->n_feat, (long)topn);
    dump(out, "dist_id", "i32", dist_id, 4, 4, (long)T, (long)g->n_mgau, (long)g->n_feat, (long)topn);
    return 0;
}

/* the MFCC front end as sphinx_fe / the live decoders drive it: whole utterance + the final partial frame */
static int
cmd_fe(int argc, char **argv)
{
    size_t nbytes, nsamps;
    int16 *raw = (int16 *)slurp(argv[0], &nbytes);
    cmd_ln_t *config;
    fe_t *fe;
    mfcc_t **cep = NULL, *last;
    int32 nfr = 0, nlast = 0, D, t, rep, reps;
    float *flat;
    struct timespec t0, t1;
    char **av = calloc(argc + 2, sizeof(char *));
    av[0] = "ref_dump";
    for (t = 2; t < argc; t++) av[t - 1] = argv[t];
    av[argc - 1] = "-dither"; av[argc] = "no";      /* (an empty argument list only prints the help) */
    config = cmd_ln_parse_r(NULL, fe_get_args(), argc + 1, av, TRUE);
    if (!config || (fe = fe_init_auto_r(config)) == NULL) { fprintf(stderr, "fe_init failed\n"); return 2; }
    nsamps = nbytes / 2;
    D = fe_get_output_size(fe);
    last = calloc(D, sizeof(mfcc_t));
    reps = getenv("REF_FE_REPS") ? atoi(getenv("REF_FE_REPS")) : 1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (rep = 0; rep < reps; rep++) {
        if (cep) ckd_free_2d((void **)cep);
        fe_start_utt(fe);
        if (fe_process_utt(fe, raw, nsamps, &cep, &nfr) < 0) { fprintf(stderr, "fe_process_utt failed\n"); return 2; }
        fe_end_utt(fe, last, &nlast);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    printf("frames %d seconds %.6f\n", (nfr + nlast) * reps, (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec));
    flat = malloc(sizeof(float) * (size_t)(nfr + nlast + 1) * D);
    for (t = 0; t < nfr; t++) memcpy(flat + (size_t)t * D, cep[t], sizeof(float) * D);
    if (nlast) memcpy(flat + (size_t)nfr * D, last, sizeof(float) * D);
    dump(argv[1], "cep", "f32", flat, 4, 2, (long)(nfr + nlast), (long)D);
    return 0;
}

/* heap VALUES.i32 OUTDIR: the reference's own heap (sphinxbase util/heap.c): insert data = index with the
 * given values in order, pop everything; dumps the pop order (vithist_prune pops its entries this way) */
#include "sphinxbase/heap.h"
static int
cmd_heap(int argc, char **argv)
{
    size_t nb;
    long n, i;
    int32 *v = (int32 *)slurp(argv[0], &nb), *order;
    heap_t *h = heap_new();
    void *data;
    int32 val;
    n = (long)(nb / 4);
    order = calloc(n + 1, sizeof(int32));
    for (i = 0; i < n; i++) heap_insert(h, (void *)(long)(i + 1), v[i]);
    for (i = 0; heap_pop(h, &data, &val) > 0; i++) order[i] = (int32)((long)data - 1);
    if (i != n) { fprintf(stderr, "heap: popped %ld of %ld\n", i, n); return 2; }
    dump(argv[1], "order", "i32", order, 4, 1, n);
    heap_destroy(h);
    return 0;
}

int
main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: ref_dump CMD ...\n"); return 1; }
    err_set_logfp(NULL);        /* silence E_INFO chatter */
    if (!strcmp(argv[1], "logmath") && argc == 5) return cmd_logmath(argc - 2, argv + 2);
    if (!strcmp(argv[1], "mgau") && argc == 11) return cmd_mgau(argc - 2, argv + 2);
    if (!strcmp(argv[1], "frame") && argc == 16) return cmd_frame(argc - 2, argv + 2);
    if (!strcmp(argv[1], "tmat") && argc == 7) return cmd_tmat(argc - 2, argv + 2);
    if (!strcmp(argv[1], "bench_mgau") && argc == 8) return cmd_bench_mgau(argc - 2, argv + 2);
    if (!strcmp(argv[1], "feat") && argc == 4) return cmd_feat(argc - 2, argv + 2);
    if (!strcmp(argv[1], "feat") && argc == 7) {       /* feat MFC CMN VARNORM AGC OUTDIR */
        char *av[5] = { argv[2], argv[6], argv[3], argv[4], argv[5] };
        return cmd_feat(5, av);
    }
    if (!strcmp(argv[1], "fe") && argc >= 4) return cmd_fe(argc - 2, argv + 2);
    if (!strcmp(argv[1], "hmm") && argc == 14) return cmd_hmm(argc - 2, argv + 2);
    if (!strcmp(argv[1], "ms") && argc == 12) return cmd_ms(argc - 2, argv + 2);
    if (!strcmp(argv[1], "heap") && argc == 4) return cmd_heap(argc - 2, argv + 2);
    fprintf(stderr, "ref_dump: bad command/arity: %s (%d args)\n", argv[1], argc - 2);
    return 1;
}
