/*
 * s3o_psms.c -- CPU restatement (TEST INFRASTRUCTURE ONLY) of POCKETSPHINX's continuous scorer, the
 * object behind the ps_mgaufuncs_t vtable (pocketsphinx/src/libpocketsphinx/acmod.h:97-110) that
 * acmod_score calls (acmod.c:1076-1131) when the model is neither semi-continuous nor PTM:
 *
 *   ms_gauden.c:307-357  gauden_dist_precompute: det += (float32) logmath_log(1/sqrt(2 pi var)),
 *                        var <- (float32) logmath_ln_to_log(1/(2 var))    [log-base units, float32]
 *   ms_gauden.c:415-520  compute_dist(_all): float32 chain  dval -= diff*diff*var,  top-N kept in
 *                        DESCENDING order; a tie goes BEFORE the entry it ties with
 *   ms_senone.c:150-283  senone_mixw_read: normalise, floor, normalise, -logmath_log, + rounding,
 *                        >> SENSCR_SHIFT (10), saturated to 8 bits
 *   ms_senone.c:367-421  senone_eval: (int32)dist + 1023 >> 10, minus the 8-bit weight, log-add on a
 *                        logmath shifted by 10 bits, negated sum over streams, / -aw, int16 clamp
 *   ms_mgau.c:163-252    ms_cont_mgau_frame_eval: all senones or a delta-encoded active list; scores
 *                        are NEGATED (smaller = better) and normalised to best = 0
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "s3o.h"

#define PS_SHIFT 10
#define PS_WORST_DIST ((float)(int32_t)0x80000000)

static size_t
cb_off(const s3o_psms_t *ms, int32_t m, int32_t f, int32_t d)
{
    return (size_t)m * ms->n_density * ms->veclen + (size_t)ms->n_density * ms->featoff[f] + (size_t)d * ms->featlen[f];
}

s3o_psms_t *
s3o_psms_init(const float *mean, const float *var, const float *mixw, int32_t n_mgau, int32_t n_feat,
              int32_t n_density, const int32_t *featlen, int32_t n_sen, const int32_t *sen2mgau,
              double varfloor_d, double mixwfloor, int32_t topn, int32_t aw, double logbase)
{
    s3o_psms_t *ms = (s3o_psms_t *)calloc(1, sizeof *ms);
    float varfloor = (float)varfloor_d;
    int32_t m, f, d, i, s, c;
    size_t n;
    ms->n_mgau = n_mgau; ms->n_feat = n_feat; ms->n_density = n_density; ms->n_sen = n_sen; ms->aw = aw;
    ms->lm = s3o_logmath_init(logbase, 0, 0);            /* acmod's logmath: no shift, no table */
    ms->lm8 = s3o_logmath_init(logbase, PS_SHIFT, 1);    /* senone_init: shifted by SENSCR_SHIFT, table */
    ms->featlen = (int32_t *)malloc(sizeof(int32_t) * n_feat);
    ms->featoff = (int32_t *)malloc(sizeof(int32_t) * (n_feat + 1));
    for (f = 0, ms->veclen = 0; f < n_feat; f++) { ms->featlen[f] = featlen[f]; ms->featoff[f] = ms->veclen; ms->veclen += featlen[f]; }
    ms->featoff[n_feat] = ms->veclen;
    n = (size_t)n_mgau * n_density * ms->veclen;
    ms->mean = (float *)malloc(sizeof(float) * n);
    ms->var = (float *)malloc(sizeof(float) * n);
    memcpy(ms->mean, mean, sizeof(float) * n);
    memcpy(ms->var, var, sizeof(float) * n);
    ms->det = (float *)calloc((size_t)n_mgau * n_feat * n_density, sizeof(float));
    for (m = 0; m < n_mgau; m++)
        for (f = 0; f < n_feat; f++)
            for (d = 0; d < n_density; d++) {
                float *varp = ms->var + cb_off(ms, m, f, d);
                float *detp = &ms->det[((size_t)m * n_feat + f) * n_density + d];
                *detp = 0;
                for (i = 0; i < featlen[f]; i++, varp++) {
                    if (*varp < varfloor) *varp = varfloor;
                    *detp += (float)s3o_logmath_log(ms->lm, 1.0 / sqrt(*varp * 2.0 * M_PI));
                    *varp = (float)s3o_logmath_ln_to_log(ms->lm, (1.0 / (*varp * 2.0)));
                }
            }
    ms->pdf = (uint8_t *)malloc((size_t)n_sen * n_feat * n_density);
    {
        float *pdf = (float *)malloc(sizeof(float) * n_density);
        for (s = 0; s < n_sen; s++)
            for (f = 0; f < n_feat; f++) {
                double sum = 0.0;
                memcpy(pdf, mixw + ((size_t)s * n_feat + f) * n_density, sizeof(float) * n_density);
                for (c = 0; c < n_density; c++) sum += pdf[c];
                if (sum != 0.0) { double r = 1.0 / sum; for (c = 0; c < n_density; c++) pdf[c] = (float)((double)pdf[c] * r); }
                for (c = 0; c < n_density; c++) if (pdf[c] < mixwfloor) pdf[c] = (float)mixwfloor;
                sum = 0.0;
                for (c = 0; c < n_density; c++) sum += pdf[c];
                if (sum != 0.0) { double r = 1.0 / sum; for (c = 0; c < n_density; c++) pdf[c] = (float)((double)pdf[c] * r); }
                for (c = 0; c < n_density; c++) {
                    int32_t p = -(s3o_logmath_log(ms->lm, pdf[c]));
                    p += (1 << (PS_SHIFT - 1)) - 1;
                    ms->pdf[((size_t)s * n_feat + f) * n_density + c] = (uint8_t)((p < (255 << PS_SHIFT)) ? (p >> PS_SHIFT) : 255);
                }
            }
        free(pdf);
    }
    ms->mgau = (int32_t *)malloc(sizeof(int32_t) * n_sen);
    for (s = 0; s < n_sen; s++) ms->mgau[s] = sen2mgau ? sen2mgau[s] : s;
    ms->topn = (topn == 0 || topn > n_density) ? n_density : topn;
    ms->dist = (float *)calloc((size_t)n_mgau * n_feat * ms->topn, sizeof(float));
    ms->dist_id = (int32_t *)calloc((size_t)n_mgau * n_feat * ms->topn, sizeof(int32_t));
    ms->mgau_active = (uint8_t *)calloc(n_mgau, 1);
    return ms;
}

void
s3o_psms_free(s3o_psms_t *ms)
{
    if (!ms) return;
    free(ms->featlen); free(ms->featoff); free(ms->mean); free(ms->var); free(ms->det); free(ms->pdf);
    free(ms->mgau); free(ms->dist); free(ms->dist_id); free(ms->mgau_active);
    s3o_logmath_free(ms->lm); s3o_logmath_free(ms->lm8);
    free(ms);
}

static void
gauden_dist(s3o_psms_t *ms, int32_t m, const float *obs)
{
    int32_t f, d, i, j, n_top = ms->topn, nd = ms->n_density;
    for (f = 0; f < ms->n_feat; f++) {
        const float *x = obs + ms->featoff[f];
        const float *det = &ms->det[((size_t)m * ms->n_feat + f) * nd];
        float *od = ms->dist + ((size_t)m * ms->n_feat + f) * n_top;
        int32_t *oi = ms->dist_id + ((size_t)m * ms->n_feat + f) * n_top;
        int32_t flen = ms->featlen[f];
        if (n_top >= nd) {
            for (d = 0; d < nd; d++) {
                const float *mu = ms->mean + cb_off(ms, m, f, d), *v = ms->var + cb_off(ms, m, f, d);
                float dval = det[d];
                for (i = 0; i < flen; i++) { float diff = x[i] - mu[i]; dval -= diff * diff * v[i]; }
                od[d] = dval; oi[d] = d;
            }
            continue;
        }
        for (i = 0; i < n_top; i++) od[i] = PS_WORST_DIST;
        for (d = 0; d < nd; d++) {
            const float *mu = ms->mean + cb_off(ms, m, f, d), *v = ms->var + cb_off(ms, m, f, d);
            float dval = det[d];
            for (i = 0; (i < flen) && (dval >= od[n_top - 1]); i++) { float diff = x[i] - mu[i]; dval -= diff * diff * v[i]; }
            if ((i < flen) || (dval < od[n_top - 1])) continue;
            for (i = 0; (i < n_top) && (dval < od[i]); i++);
            for (j = n_top - 1; j > i; --j) { od[j] = od[j - 1]; oi[j] = oi[j - 1]; }
            od[i] = dval; oi[i] = d;
        }
    }
}

static int32_t
senone_eval(const s3o_psms_t *ms, int32_t id)
{
    int32_t scr = 0, f, t, m = ms->mgau[id];
    for (f = 0; f < ms->n_feat; f++) {
        const float *fd = ms->dist + ((size_t)m * ms->n_feat + f) * ms->topn;
        const int32_t *fi = ms->dist_id + ((size_t)m * ms->n_feat + f) * ms->topn;
        const uint8_t *pdf = ms->pdf + ((size_t)id * ms->n_feat + f) * ms->n_density;
        int32_t fden = ((int32_t)fd[0] + ((1 << PS_SHIFT) - 1)) >> PS_SHIFT;
        int32_t fscr = fden + -(int32_t)pdf[fi[0]];
        for (t = 1; t < ms->topn; t++) {
            fden = ((int32_t)fd[t] + ((1 << PS_SHIFT) - 1)) >> PS_SHIFT;
            fscr = s3o_logmath_add(ms->lm8, fscr, fden + -(int32_t)pdf[fi[t]]);
        }
        scr -= fscr;
    }
    scr /= ms->aw;
    if (scr > 32767) scr = 32767;
    if (scr < -32768) scr = -32768;
    return scr;
}

/* ms_cont_mgau_frame_eval: senone_active = delta-encoded ascending list (NULL / compallsen: all) */
void
s3o_psms_frame_eval(s3o_psms_t *ms, int16_t *senscr, const uint8_t *senone_active, int32_t n_senone_active,
                    const float *feat, int32_t compallsen)
{
    int32_t gid, s, i, n, best;
    if (compallsen) {
        for (gid = 0; gid < ms->n_mgau; gid++) gauden_dist(ms, gid, feat);
        best = 0x7fffffff;
        for (s = 0; s < ms->n_sen; s++) { senscr[s] = (int16_t)senone_eval(ms, s); if (best > senscr[s]) best = senscr[s]; }
        for (s = 0; s < ms->n_sen; s++) {
            int32_t bs = senscr[s] - best;
            if (bs > 32767) bs = 32767;
            if (bs < -32768) bs = -32768;
            senscr[s] = (int16_t)bs;
        }
        return;
    }
    for (gid = 0; gid < ms->n_mgau; gid++) ms->mgau_active[gid] = 0;
    for (i = 0, n = 0; i < n_senone_active; i++) { s = senone_active[i] + n; ms->mgau_active[ms->mgau[s]] = 1; n = s; }
    for (gid = 0; gid < ms->n_mgau; gid++) if (ms->mgau_active[gid]) gauden_dist(ms, gid, feat);
    best = 0x7fffffff;
    for (i = 0, n = 0; i < n_senone_active; i++) {
        s = senone_active[i] + n;
        senscr[s] = (int16_t)senone_eval(ms, s);
        if (best > senscr[s]) best = senscr[s];
        n = s;
    }
    for (i = 0, n = 0; i < n_senone_active; i++) {
        int32_t bs;
        s = senone_active[i] + n;
        bs = senscr[s] - best;
        if (bs > 32767) bs = 32767;
        if (bs < -32768) bs = -32768;
        senscr[s] = (int16_t)bs;
        n = s;
    }
}
