/*
 * s3o_ms.c -- CPU restatement (TEST INFRASTRUCTURE ONLY) of sphinx3's multi-stream
 * ("s3.0-style") senone scorer, the route `-senmgau .s3cont.` / `.semi.` selects
 * (kbcore.c:342-364):
 *
 *   sphinx3/src/libs3decoder/libam/ms_gauden.c:330-383  gauden_dist_precompute
 *   ms_gauden.c:487-592                                  compute_dist_all / compute_dist (top-N)
 *   ms_gauden.c:599-644                                  gauden_dist
 *   libam/ms_senone.c:212-360                            senone_mixw_read (normalise, floor, -logs3;
 *                                                        TRUNCATE_LOGPDF is not defined: no truncation)
 *   ms_senone.c:442-490                                  senone_eval
 *   libam/ms_mgau.c:242-329                              ms_cont_mgau_frame_eval (no interpolation file)
 *
 * Arithmetic that fixes the bit patterns: the determinant term is ACCUMULATED IN float32
 * (`*detp += (float32) log(*varp)`), the precision is (float32)(1.0 / (var * 2.0)), the
 * distance is a float64 chain det + sum((float32 diff)^2 * v) in dimension order, the top-N
 * list is kept in (distance, codeword) order by insertion, `-dist` is floored at
 * log_to_ln(S3_LOGPROB_ZERO) and truncated by logmath_ln_to_log, and the per-feature
 * log-add runs over the list in ITS order (sorted when topn < n_density, codeword order
 * otherwise).
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "s3o.h"

static size_t
cb_off(const s3o_ms_t *ms, int32_t m, int32_t f, int32_t d)
{
    return (size_t)m * ms->n_density * ms->veclen + (size_t)ms->n_density * ms->featoff[f]
        + (size_t)d * ms->featlen[f];
}

s3o_ms_t *
s3o_ms_init(const float *mean, const float *var, const float *mixw, int32_t n_mgau, int32_t n_feat,
            int32_t n_density, const int32_t *featlen, int32_t n_sen, const int32_t *sen2mgau,
            double varfloor_d, double mixwfloor, int32_t topn, const s3o_logmath_t *lm)
{
    s3o_ms_t *ms = (s3o_ms_t *)calloc(1, sizeof *ms);
    float varfloor = (float)varfloor_d;     /* gauden_init takes float32 varfloor */
    int32_t m, f, d, i, s, c;
    size_t n;
    ms->n_mgau = n_mgau; ms->n_feat = n_feat; ms->n_density = n_density; ms->n_sen = n_sen;
    ms->lm = lm;
    ms->featlen = (int32_t *)malloc(sizeof(int32_t) * n_feat);
    ms->featoff = (int32_t *)malloc(sizeof(int32_t) * (n_feat + 1));
    for (f = 0, ms->veclen = 0; f < n_feat; f++) {
        ms->featlen[f] = featlen[f];
        ms->featoff[f] = ms->veclen;
        ms->veclen += featlen[f];
    }
    ms->featoff[n_feat] = ms->veclen;
    n = (size_t)n_mgau * n_density * ms->veclen;
    ms->mean = (float *)malloc(sizeof(float) * n);
    ms->var = (float *)malloc(sizeof(float) * n);
    memcpy(ms->mean, mean, sizeof(float) * n);
    memcpy(ms->var, var, sizeof(float) * n);
    ms->det = (float *)calloc((size_t)n_mgau * n_feat * n_density, sizeof(float));
    /* gauden_dist_precompute, ms_gauden.c:330-383 */
    for (m = 0; m < n_mgau; m++)
        for (f = 0; f < n_feat; f++)
            for (d = 0; d < n_density; d++) {
                float *varp = ms->var + cb_off(ms, m, f, d);
                float *detp = &ms->det[((size_t)m * n_feat + f) * n_density + d];
                *detp = (float)0.0;
                for (i = 0; i < featlen[f]; i++, varp++) {
                    if (*varp < varfloor)
                        *varp = varfloor;
                    *detp += (float)(log(*varp));
                    *varp = (float)(1.0 / (*varp * 2.0));
                }
                *detp += (float)(featlen[f] * log(2.0 * M_PI));
                *detp *= (float)0.5;
            }
    ms->min_density = s3o_logmath_log_to_ln(lm, S3O_LOGPROB_ZERO);      /* ms_gauden.c:420 */
    /* senone_mixw_read, ms_senone.c:300-337 */
    ms->pdf = (int32_t *)malloc(sizeof(int32_t) * (size_t)n_sen * n_feat * n_density);
    {
        float *pdf = (float *)malloc(sizeof(float) * n_density);
        for (s = 0; s < n_sen; s++)
            for (f = 0; f < n_feat; f++) {
                double sum = 0.0;
                const float *src = mixw + ((size_t)s * n_feat + f) * n_density;
                memcpy(pdf, src, sizeof(float) * n_density);
                for (c = 0; c < n_density; c++) sum += pdf[c];                     /* vector_sum_norm */
                if (sum != 0.0) { double r = 1.0 / sum; for (c = 0; c < n_density; c++) pdf[c] = (float)((double)pdf[c] * r); }
                for (c = 0; c < n_density; c++) if (pdf[c] < mixwfloor) pdf[c] = (float)mixwfloor;   /* vector_floor */
                sum = 0.0;
                for (c = 0; c < n_density; c++) sum += pdf[c];
                if (sum != 0.0) { double r = 1.0 / sum; for (c = 0; c < n_density; c++) pdf[c] = (float)((double)pdf[c] * r); }
                for (c = 0; c < n_density; c++)
                    ms->pdf[((size_t)s * n_feat + f) * n_density + c] = -(s3o_logs3(lm, pdf[c]));
            }
        free(pdf);
    }
    ms->mgau = (int32_t *)malloc(sizeof(int32_t) * n_sen);
    for (s = 0; s < n_sen; s++)
        ms->mgau[s] = sen2mgau ? sen2mgau[s] : s;          /* ".s3cont.": 1-to-1 (ms_senone.c:399-409) */
    ms->topn = (topn == 0 || topn > n_density) ? n_density : topn;      /* ms_mgau.c:214-219 */
    ms->dist_id = (int32_t *)calloc((size_t)n_mgau * n_feat * ms->topn, sizeof(int32_t));
    ms->dist = (int32_t *)calloc((size_t)n_mgau * n_feat * ms->topn, sizeof(int32_t));
    ms->mgau_active = (uint8_t *)calloc(n_mgau, 1);
    return ms;
}

void
s3o_ms_free(s3o_ms_t *ms)
{
    if (!ms) return;
    free(ms->featlen); free(ms->featoff); free(ms->mean); free(ms->var); free(ms->det);
    free(ms->pdf); free(ms->mgau); free(ms->dist_id); free(ms->dist); free(ms->mgau_active);
    free(ms);
}

/* gauden_dist for one codebook (ms_gauden.c:599-644 with compute_dist :541-592) */
static void
gauden_dist(s3o_ms_t *ms, int32_t m, const float *obs)
{
    int32_t f, d, i, j, t, n_top = ms->topn, nd = ms->n_density;
    double *dd = (double *)malloc(sizeof(double) * n_top);
    int32_t *di = (int32_t *)calloc(n_top, sizeof(int32_t));
    for (f = 0; f < ms->n_feat; f++) {
        const float *x = obs + ms->featoff[f];
        const float *det = &ms->det[((size_t)m * ms->n_feat + f) * nd];
        int32_t flen = ms->featlen[f];
        if (n_top >= nd) {                  /* compute_dist_all: codeword order, no sorting */
            for (d = 0; d < nd; d++) {
                const float *mu = ms->mean + cb_off(ms, m, f, d), *v = ms->var + cb_off(ms, m, f, d);
                double dval = det[d], diff;
                for (i = 0; i < flen; i++) { diff = x[i] - mu[i]; dval += diff * diff * v[i]; }
                dd[d] = dval; di[d] = d;
            }
        }
        else {
            for (i = 0; i < n_top; i++) dd[i] = DBL_MAX;
            for (d = 0; d < nd; d++) {
                const float *mu = ms->mean + cb_off(ms, m, f, d), *v = ms->var + cb_off(ms, m, f, d);
                double dval = det[d], diff;
                for (i = 0; (i < flen) && (dval <= dd[n_top - 1]); i++) { diff = x[i] - mu[i]; dval += diff * diff * v[i]; }
                if ((i < flen) || (dval >= dd[n_top - 1]))
                    continue;
                for (i = 0; (i < n_top) && (dval >= dd[i]); i++);
                for (j = n_top - 1; j > i; --j) { dd[j] = dd[j - 1]; di[j] = di[j - 1]; }
                dd[i] = dval; di[i] = d;
            }
        }
        for (t = 0; t < n_top; t++) {
            size_t o = ((size_t)m * ms->n_feat + f) * n_top + t;
            double v = -dd[t];
            if (v < ms->min_density) v = ms->min_density;
            ms->dist_id[o] = di[t];
            ms->dist[o] = (int32_t)s3o_logmath_ln_to_log(ms->lm, v);
        }
    }
    free(dd); free(di);
}

/* senone_eval, ms_senone.c:442-490 (int32 - uint32 arithmetic == wrapping int32) */
static int32_t
senone_eval(const s3o_ms_t *ms, int32_t id)
{
    int32_t scr = 0, f, t, m = ms->mgau[id];
    for (f = 0; f < ms->n_feat; f++) {
        const int32_t *fd = ms->dist + ((size_t)m * ms->n_feat + f) * ms->topn;
        const int32_t *fi = ms->dist_id + ((size_t)m * ms->n_feat + f) * ms->topn;
        const int32_t *pdf = ms->pdf + ((size_t)id * ms->n_feat + f) * ms->n_density;
        int32_t fscr = (int32_t)((uint32_t)fd[0] - (uint32_t)pdf[fi[0]]);
        for (t = 1; t < ms->topn; t++)
            fscr = s3o_logmath_add(ms->lm, fscr, (int32_t)((uint32_t)fd[t] - (uint32_t)pdf[fi[t]]));
        scr = (int32_t)((uint32_t)scr + (uint32_t)fscr);
    }
    return scr;
}

/* ms_cont_mgau_frame_eval, ms_mgau.c:242-329, interp == NULL */
int32_t
s3o_ms_cont_mgau_frame_eval(s3o_ms_t *ms, const uint8_t *sen_active, int32_t *senscr, const float *feat)
{
    int32_t s, gid, best;
    for (gid = 0; gid < ms->n_mgau; gid++) ms->mgau_active[gid] = 0;
    for (s = 0; s < ms->n_sen; s++)
        if (sen_active[s]) ms->mgau_active[ms->mgau[s]] = 1;
    for (gid = 0; gid < ms->n_mgau; gid++)
        if (ms->mgau_active[gid]) gauden_dist(ms, gid, feat);
    best = (int32_t)0x80000000;
    for (s = 0; s < ms->n_sen; s++)
        if (sen_active[s]) {
            senscr[s] = senone_eval(ms, s);
            if (best < senscr[s]) best = senscr[s];
        }
    for (s = 0; s < ms->n_sen; s++)
        if (sen_active[s]) senscr[s] = (int32_t)((uint32_t)senscr[s] - (uint32_t)best);
    return best;
}
