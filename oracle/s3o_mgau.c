/*
 * oracle/s3o_mgau.c -- CPU ORACLE (test infrastructure only; see s3o.h).
 *
 * Continuous-density GMM senone scoring, restating
 *   sphinx3/src/libs3decoder/libam/cont_mgau.c:507-683   (mixw floor/normalise/logs3)
 *   sphinx3/src/libs3decoder/libam/cont_mgau.c:700-783   (mgau_uninit_compact)
 *   sphinx3/src/libs3decoder/libam/cont_mgau.c:792-816   (mgau_var_floor)
 *   sphinx3/src/libs3decoder/libam/cont_mgau.c:857-894   (mgau_precomp)
 *   sphinx3/src/libs3decoder/libam/cont_mgau.c:901-956   (mgau_init ordering)
 *   sphinx3/src/libs3decoder/libam/cont_mgau.c:1034-1205 (mgau_eval_all/_active/mgau_eval)
 *   sphinx3/src/libs3decoder/libam/approx_cont_mgau.c:94-143, 188-284, 303-357, 367-616
 *   sphinx3/src/libs3decoder/libcommon/vector.c:105-204  (vector helpers)
 *   sphinx3/src/libs3decoder/libsearch/dict2pid.c:1029-1048 (dict2pid_comsenscr)
 *
 * Arithmetic notes that fix the bit pattern (SURVEY.md 7 "hard parts"):
 *   diff = x[i] - m[i]   is a float32 subtraction widened to double;
 *   dval -= diff*diff*v  is two double multiplies and one double subtract,
 *                        no FMA (build with gcc -O2, x86-64 SSE2);
 *   gauscr = (int32)(f*dval) + mixw  truncates toward zero;
 *   the log-add over components is sequential in component order.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <assert.h>
#include "s3o.h"

/* vector.c:181-187 */
static int
vec_is_zero(const float *v, int n)
{
    int i;
    for (i = 0; (i < n) && (v[i] == 0.0); i++);
    return i == n;
}

/* vector.c:190-204 */
static int
vec_is_nan(const float *v, int n)
{
    int i;
    for (i = 0; i < n; i++)
        if (isnan(v[i]))
            return 1;
    return 0;
}

/* vector.c:105-123 */
static double
vec_sum_norm(float *v, int n)
{
    double sum = 0.0, f;
    int i;
    for (i = 0; i < n; i++)
        sum += v[i];
    if (sum != 0.0) {
        f = 1.0 / sum;
        for (i = 0; i < n; i++)
            v[i] = (float)((double)v[i] * (double)f);
    }
    return sum;
}

/* vector.c:138-145 */
static void
vec_nz_floor(float *v, int n, double flr)
{
    int i;
    for (i = 0; i < n; i++)
        if ((v[i] != 0.0) && (v[i] < flr))
            v[i] = (float)flr;
}

s3o_mgau_t *
s3o_mgau_init(const float *mean, const float *var, const float *mixw,
              int32_t n_mgau, int32_t n_density, int32_t veclen,
              double varfloor, double mixwfloor, int precomp,
              const s3o_logmath_t *lm)
{
    s3o_mgau_t *g = (s3o_mgau_t *)calloc(1, sizeof(*g));
    size_t ng = (size_t)n_mgau * n_density;
    int32_t m, c, c2, i, j;
    float *pdf;

    g->n_mgau = n_mgau;
    g->max_comp = n_density;
    g->veclen = veclen;
    g->lm = lm;
    g->n_comp = (int32_t *)malloc(sizeof(int32_t) * n_mgau);
    g->mean = (float *)malloc(sizeof(float) * ng * veclen);
    g->var = (float *)malloc(sizeof(float) * ng * veclen);
    g->lrd = (float *)calloc(ng, sizeof(float));
    g->mixw = (int32_t *)calloc(ng, sizeof(int32_t));
    g->bstidx = (int32_t *)malloc(sizeof(int32_t) * n_mgau);
    g->bstscr = (int32_t *)malloc(sizeof(int32_t) * n_mgau);
    g->updatetime = (int32_t *)malloc(sizeof(int32_t) * n_mgau);
    memcpy(g->mean, mean, sizeof(float) * ng * veclen);
    memcpy(g->var, var, sizeof(float) * ng * veclen);
    for (m = 0; m < n_mgau; m++)
        g->n_comp[m] = n_density;
    s3o_mgau_reset_state(g);

    /* mixture weights: cont_mgau.c:624-661 */
    pdf = (float *)malloc(sizeof(float) * n_density);
    for (m = 0; m < n_mgau; m++) {
        memcpy(pdf, mixw + (size_t)m * n_density, sizeof(float) * n_density);
        if (vec_is_zero(pdf, n_density)) {
            for (j = 0; j < n_density; j++)
                g->mixw[(size_t)m * n_density + j] = S3O_LOGPROB_ZERO;
        }
        else {
            vec_nz_floor(pdf, n_density, mixwfloor);
            vec_sum_norm(pdf, n_density);
            for (j = 0; j < n_density; j++)
                g->mixw[(size_t)m * n_density + j] =
                    (pdf[j] != 0.0) ? s3o_logs3(lm, pdf[j]) : S3O_LOGPROB_ZERO;
        }
    }
    free(pdf);

    /* mgau_uninit_compact: cont_mgau.c:700-783 (diagonal case) */
    for (m = 0; m < n_mgau; m++) {
        for (c = 0, c2 = 0; c < g->n_comp[m]; c++) {
            float *mc = g->mean + ((size_t)m * n_density + c) * veclen;
            float *vc = g->var + ((size_t)m * n_density + c) * veclen;
            int keep = !(vec_is_nan(mc, veclen) || vec_is_nan(vc, veclen)
                         || vec_is_zero(vc, veclen));
            if (keep) {
                if (c2 != c) {
                    memcpy(g->mean + ((size_t)m * n_density + c2) * veclen, mc,
                           sizeof(float) * veclen);
                    memcpy(g->var + ((size_t)m * n_density + c2) * veclen, vc,
                           sizeof(float) * veclen);
                    g->mixw[(size_t)m * n_density + c2] = g->mixw[(size_t)m * n_density + c];
                }
                c2++;
            }
        }
        g->n_comp[m] = c2;
    }

    /* mgau_var_floor: cont_mgau.c:792-816 */
    if (varfloor > 0.0) {
        for (m = 0; m < n_mgau; m++)
            for (c = 0; c < g->n_comp[m]; c++) {
                float *vc = g->var + ((size_t)m * n_density + c) * veclen;
                for (i = 0; i < veclen; i++)
                    if (vc[i] < varfloor)
                        vc[i] = (float)varfloor;
            }
    }

    /* mgau_precomp: cont_mgau.c:857-894 (diagonal case) */
    if (precomp) {
        for (m = 0; m < n_mgau; m++)
            for (c = 0; c < g->n_comp[m]; c++) {
                float *vc = g->var + ((size_t)m * n_density + c) * veclen;
                double lrd = 0.0;
                for (i = 0; i < veclen; i++) {
                    lrd += log(vc[i]);
                    vc[i] = (float)(1.0 / (vc[i] * 2.0));
                }
                lrd += veclen * log(2.0 * M_PI);
                g->lrd[(size_t)m * n_density + c] = (float)(-0.5 * lrd);
            }
    }

    /* cont_mgau.c:950-951 */
    g->distfloor = s3o_logmath_log_to_ln(lm, S3O_LOGPROB_ZERO);
    return g;
}

void
s3o_mgau_free(s3o_mgau_t *g)
{
    if (!g) return;
    free(g->n_comp); free(g->mean); free(g->var); free(g->lrd); free(g->mixw);
    free(g->bstidx); free(g->bstscr); free(g->updatetime);
    free(g);
}

void
s3o_mgau_reset_state(s3o_mgau_t *g)
{
    int32_t m;
    for (m = 0; m < g->n_mgau; m++) {
        g->bstidx[m] = S3O_NO_BSTIDX;
        g->bstscr[m] = S3O_LOGPROB_ZERO;
        g->updatetime[m] = S3O_NOT_UPDATED;
    }
}

/* the inner loop of mgau_eval_all / mgau_eval_active: cont_mgau.c:1058-1063 */
static double
gau_dval(const s3o_mgau_t *g, int32_t m, int32_t c, const float *x)
{
    const float *mc = g->mean + ((size_t)m * g->max_comp + c) * g->veclen;
    const float *vc = g->var + ((size_t)m * g->max_comp + c) * g->veclen;
    double dval = g->lrd[(size_t)m * g->max_comp + c];
    double diff;
    int32_t i;
    for (i = 0; i < g->veclen; i++) {
        diff = x[i] - mc[i];            /* float32 subtract, then widened */
        dval -= diff * diff * vc[i];
    }
    return dval;
}

int32_t
s3o_mgau_eval(s3o_mgau_t *g, int32_t m, const int32_t *active,
              const float *x, int32_t fr, int32_t update_best_id)
{
    const s3o_logmath_t *lm = g->lm;
    const int32_t *mixw = g->mixw + (size_t)m * g->max_comp;
    double f = 1.0 / log(lm->base);
    double dval;
    int32_t score = S3O_LOGPROB_ZERO, gauscr, c, j;

    if (update_best_id) {               /* cont_mgau.c:1185-1189 */
        g->bstidx[m] = S3O_NO_BSTIDX;
        g->bstscr[m] = S3O_LOGPROB_ZERO;
        g->updatetime[m] = fr;
    }

    if (!active) {
        /* mgau_eval_all, cont_mgau.c:1034-1122.  The reference interleaves
         * components pairwise; the arithmetic per component and the order of
         * the log-adds are those of a plain loop, EXCEPT that the bstidx
         * update of the first component of each pair is not guarded by
         * update_best_id (cont_mgau.c:1076-1079 vs 1085-1088). */
        int32_t n = g->n_comp[m];
        for (c = 0; c < n; c++) {
            int in_pair_first = ((c & 1) == 0) && (c + 1 < n);
            dval = gau_dval(g, m, c, x);
            if (dval < g->distfloor)
                dval = g->distfloor;
            gauscr = (int32_t)(f * dval) + mixw[c];
            score = s3o_logmath_add(lm, score, gauscr);
            if ((in_pair_first || update_best_id) && gauscr > g->bstscr[m]) {
                g->bstidx[m] = c;
                g->bstscr[m] = gauscr;
            }
        }
    }
    else {
        /* mgau_eval_active, cont_mgau.c:1125-1167 */
        for (j = 0; active[j] >= 0; j++) {
            c = active[j];
            dval = gau_dval(g, m, c, x);
            if (dval < g->distfloor)
                dval = g->distfloor;
            gauscr = (int32_t)(f * dval) + mixw[c];
            score = s3o_logmath_add(lm, score, gauscr);
            if (update_best_id && gauscr > g->bstscr[m]) {
                g->bstidx[m] = c;
                g->bstscr[m] = gauscr;
            }
        }
    }
    if (score <= S3O_LOGPROB_ZERO)      /* cont_mgau.c:1200-1203 */
        score = S3O_LOGPROB_ZERO;
    return score;
}

/* approx_mgau_eval without GS / SVQ short-lists: approx_cont_mgau.c:188-284.
 * With mgau_sl == NULL the "recompute" branch (:258-281) never fires. */
static int32_t
approx_mgau_eval(s3o_mgau_t *g, int32_t s, int32_t *senscr, const float *feat, int32_t fr)
{
    senscr[s] = s3o_mgau_eval(g, s, NULL, feat, fr, 1);
    return g->n_comp[s];
}

void
s3o_approx_cont_mgau_ci_eval(s3o_mgau_t *g, const int16_t *cd2cisen, int32_t n_sen,
                             const float *feat, int32_t *ci_senscr,
                             int32_t *best_score, int32_t fr)
{
    int32_t s, n_cis = 0, n_cig = 0;
    /* mdef_is_cisenone(mdef, s) == (s < n_sen && s == cd2cisen[s]), mdef.c:414-425 */
    for (s = 0; s < n_sen && s == cd2cisen[s]; s++) {
        n_cig += approx_mgau_eval(g, s, ci_senscr, feat, fr);
        n_cis++;
    }
    *best_score = S3O_MAX_NEG_INT32;
    for (s = 0; s < n_sen && s == cd2cisen[s]; s++)
        if (ci_senscr[s] > *best_score)
            *best_score = ci_senscr[s];
    g->frm_ci_sen_eval = n_cis;
    g->frm_ci_gau_eval = n_cig;
}

/* approx_isskip, approx_cont_mgau.c:94-143 (no Gaussian selector: best_cid == -1
 * always, so rec_bstcid == best_cid holds from the second frame on; cond_ds
 * needs a selector to be meaningful and is restated only for completeness) */
static int
approx_isskip(int32_t frame, s3o_fastgmm_t *fg, int same_best_idx)
{
    assert(fg->ds_ratio != 0);
    if (fg->cond_ds > 0) {
        if (same_best_idx) {
            if (fg->skip_count < fg->ds_ratio - 1) {
                ++fg->skip_count;
                return 1;
            }
            fg->skip_count = 0;
            return 0;
        }
        return 0;
    }
    return (frame % fg->ds_ratio == 0) ? 0 : 1;
}

static const int32_t *g_ci_sort_key;
static int
intcmp(const void *v1, const void *v2)      /* approx_cont_mgau.c:288-292 */
{
    return g_ci_sort_key[*(const int32_t *)v2] - g_ci_sort_key[*(const int32_t *)v1];
}

/* approx_compute_dyn_ci_pbeam, approx_cont_mgau.c:303-357 */
static int32_t
compute_dyn_ci_pbeam(s3o_mgau_t *g, s3o_fastgmm_t *fg, const int16_t *cd2cisen,
                     int32_t n_ci_sen, const uint8_t *sen_active,
                     const int32_t *cache_ci_senscr)
{
    int32_t *ci_occ = (int32_t *)calloc(g->n_mgau, sizeof(int32_t));
    int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * (n_ci_sen > 0 ? n_ci_sen : 1));
    int32_t s, total, pbest;

    for (s = 0; s < g->n_mgau; s++) {
        if (s == cd2cisen[s])
            ci_occ[s] = 0;
        else if (!sen_active || sen_active[s])
            ci_occ[cd2cisen[s]]++;
    }
    for (s = 0; s < n_ci_sen; s++)
        idx[s] = s;
    g_ci_sort_key = cache_ci_senscr;
    qsort(idx, n_ci_sen, sizeof(int32_t), intcmp);

    total = 0;
    pbest = cache_ci_senscr[idx[0]];
    fg->dyn_ci_pbeam = fg->ci_pbeam;
    for (s = 0; s < n_ci_sen && cache_ci_senscr[idx[s]] > pbest + fg->ci_pbeam; s++) {
        total += ci_occ[idx[s]];
        if (total > fg->max_cd) {
            fg->dyn_ci_pbeam = cache_ci_senscr[idx[s]] - pbest;
            break;
        }
    }
    free(ci_occ);
    free(idx);
    return fg->dyn_ci_pbeam;
}

int32_t
s3o_approx_cont_mgau_frame_eval(s3o_mgau_t *g, s3o_fastgmm_t *fg,
                                const int16_t *cd2cisen, int32_t n_ci_sen,
                                uint8_t *sen_active, uint8_t *rec_sen_active,
                                int32_t *senscr, const float *feat, int32_t frame,
                                const int32_t *cache_ci_senscr)
{
    int32_t s, best, pbest, ns, ng, dyn_ci_pbeam, is_skip;
    int32_t single_el_list[2];

    best = S3O_MAX_NEG_INT32;
    pbest = S3O_MAX_NEG_INT32;
    ns = ng = 0;
    single_el_list[0] = -1;
    single_el_list[1] = -1;

    if (fg->max_cd < g->n_mgau - n_ci_sen)
        dyn_ci_pbeam = compute_dyn_ci_pbeam(g, fg, cd2cisen, n_ci_sen, sen_active,
                                            cache_ci_senscr);
    else
        dyn_ci_pbeam = fg->ci_pbeam;

    /* best_cid is -1 on every frame without a selector; rec_bstcid starts at -1 */
    is_skip = approx_isskip(frame, fg, 1);
    if (is_skip)
        dyn_ci_pbeam = (int32_t)((float)dyn_ci_pbeam * fg->tighten_factor);

    for (s = 0; s < g->n_mgau; s++) {
        int is_compute = !sen_active || sen_active[s];
        int is_ciphone = (s == cd2cisen[s]);

        if (is_ciphone) {
            senscr[s] = cache_ci_senscr[s];
            if (pbest < senscr[s]) pbest = senscr[s];
            if (best < senscr[s]) best = senscr[s];
            sen_active[s] = 1;
        }
        else if (is_compute) {
            if (senscr[cd2cisen[s]] >= pbest + dyn_ci_pbeam) {
                ng += approx_mgau_eval(g, s, senscr, feat, frame);
                ns++;
            }
            else if (g->bstidx[s] == S3O_NO_BSTIDX || g->updatetime[s] != frame - 1) {
                senscr[s] = senscr[cd2cisen[s]];        /* CI back-off */
            }
            else {                                       /* best-Gaussian back-off */
                single_el_list[0] = g->bstidx[s];
                senscr[s] = s3o_mgau_eval(g, s, single_el_list, feat, frame, is_skip ? 1 : 0);
                ng++;
            }
            if (best < senscr[s]) best = senscr[s];
        }
        rec_sen_active[s] = sen_active[s];
    }

    for (s = 0; s < g->n_mgau; s++)
        if (sen_active[s])
            senscr[s] -= best;

    g->frm_sen_eval = ns;
    g->frm_gau_eval = ng;
    return best;
}

void
s3o_dict2pid_comsenscr(int32_t n_comstate, const int32_t *comstate_off,
                       const int16_t *comstate, const int32_t *comwt,
                       const int32_t *senscr, int32_t *comsenscr)
{
    int32_t i, j, best;
    for (i = 0; i < n_comstate; i++) {
        const int16_t *cs = comstate + comstate_off[i];
        int32_t n = comstate_off[i + 1] - comstate_off[i];
        best = senscr[cs[0]];
        for (j = 1; j < n; j++)
            if (best < senscr[cs[j]])
                best = senscr[cs[j]];
        comsenscr[i] = best + comwt[i];
    }
}
