/*
 * oracle/s3o_hmm.c -- CPU ORACLE (test infrastructure only; see s3o.h).
 *
 * Transition-matrix conversion and per-HMM Viterbi update, restating
 *   sphinx3/src/libs3decoder/libam/tmat.c:232-247   (normalise, floor, logs3)
 *   sphinx3/src/libs3decoder/libam/hmm.c:130-147    (hmm_init)
 *   sphinx3/src/libs3decoder/libam/hmm.c:225-271    (hmm_clear, hmm_enter, hmm_normalize)
 *   sphinx3/src/libs3decoder/libam/hmm.c:285-412    (hmm_vit_eval_5st_lr)
 *   sphinx3/src/libs3decoder/libam/hmm.c:418-587    (hmm_vit_eval_5st_lr_mpx)
 *   sphinx3/src/libs3decoder/libam/hmm.c:592-674    (hmm_vit_eval_3st_lr)
 *   sphinx3/src/libs3decoder/libam/hmm.c:677-776    (hmm_vit_eval_3st_lr_mpx)
 *   sphinx3/src/libs3decoder/libam/hmm.c:779-852    (hmm_vit_eval_anytopo)
 *   sphinx3/src/libs3decoder/libam/hmm.c:855-873    (hmm_vit_eval dispatch)
 *
 * All score arithmetic is two's-complement int32 (the reference relies on
 * wrap-around of WORST_SCORE + senscr staying below WORST_SCORE); it is done
 * in uint32 here so the behaviour is defined.
 */
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include "s3o.h"

#define WORST S3O_LOGPROB_ZERO

static inline int32_t
add32(int32_t a, int32_t b)
{
    return (int32_t)((uint32_t)a + (uint32_t)b);
}

void
s3o_tmat_logs3(const float *tp_in, int32_t n_tmat, int32_t n_src, double tpfloor,
               const s3o_logmath_t *lm, int32_t *tp_out)
{
    int32_t n_dst = n_src + 1, i, j, k;
    float *row = (float *)malloc(sizeof(float) * n_dst);
    for (i = 0; i < n_tmat; i++)
        for (j = 0; j < n_src; j++) {
            double sum, f;
            memcpy(row, tp_in + ((size_t)i * n_src + j) * n_dst, sizeof(float) * n_dst);
            /* vector_sum_norm, vector_nz_floor, vector_sum_norm (vector.c:105-145) */
            sum = 0.0;
            for (k = 0; k < n_dst; k++) sum += row[k];
            if (sum != 0.0) {
                f = 1.0 / sum;
                for (k = 0; k < n_dst; k++) row[k] = (float)((double)row[k] * f);
            }
            for (k = 0; k < n_dst; k++)
                if ((row[k] != 0.0) && (row[k] < tpfloor)) row[k] = (float)tpfloor;
            sum = 0.0;
            for (k = 0; k < n_dst; k++) sum += row[k];
            if (sum != 0.0) {
                f = 1.0 / sum;
                for (k = 0; k < n_dst; k++) row[k] = (float)((double)row[k] * f);
            }
            for (k = 0; k < n_dst; k++)
                tp_out[((size_t)i * n_src + j) * n_dst + k] =
                    (row[k] == 0.0) ? S3O_LOGPROB_ZERO : s3o_logs3(lm, row[k]);
        }
    free(row);
}

void
s3o_hmm_clear(const s3o_hmm_ctx_t *ctx, s3o_hmm_t *h)
{
    int32_t i;
    for (i = 0; i < ctx->n_emit_state; i++) {
        h->score[i] = WORST;
        h->history[i] = -1;
    }
    h->out_score = WORST;
    h->out_history = -1;
    h->bestscore = WORST;
    h->frame = -1;
}

void
s3o_hmm_init(const s3o_hmm_ctx_t *ctx, s3o_hmm_t *h, int mpx, int32_t ssid, int32_t tmatid)
{
    int32_t i;
    memset(h, 0, sizeof(*h));
    h->mpx = (uint8_t)mpx;
    if (mpx) {
        for (i = 0; i < S3O_MAX_HMM_NSTATE; i++)
            h->mpx_ssid[i] = -1;
        h->mpx_ssid[0] = ssid;
    }
    else
        h->ssid = ssid;
    h->tmatid = tmatid;
    s3o_hmm_clear(ctx, h);
}

void
s3o_hmm_enter(s3o_hmm_t *h, int32_t score, int64_t histid, int32_t frame)
{
    h->score[0] = score;
    h->history[0] = histid;
    h->frame = frame;
}

void
s3o_hmm_normalize(const s3o_hmm_ctx_t *ctx, s3o_hmm_t *h, int32_t bestscr)
{
    int32_t i;
    for (i = 0; i < ctx->n_emit_state; i++)
        if (h->score[i] > WORST)
            h->score[i] -= bestscr;
    if (h->out_score > WORST)
        h->out_score -= bestscr;
}

#define TP(i, j) (tp[(i) * nd + (j)])
#define SEN(st) (ctx->senscore[sseq[st]])
#define MSEN(st) (ctx->senscore[ctx->sseq[(size_t)ssid[st] * ne + (st)]])

static int32_t
vit_5st_lr(const s3o_hmm_ctx_t *ctx, s3o_hmm_t *h)
{
    const int nd = 6;
    const int32_t *tp = ctx->tp + (size_t)h->tmatid * 5 * 6;
    const int16_t *sseq = ctx->sseq + (size_t)h->ssid * 5;
    int32_t s5, s4, s3, s2, s1, s0, t2, t1, t0, best = WORST;

    s4 = add32(h->score[4], SEN(4));
    s3 = add32(h->score[3], SEN(3));
    if (s3 > WORST) {
        t1 = add32(s4, TP(4, 5));
        t2 = add32(s3, TP(3, 5));
        if (t1 > t2) { s5 = t1; h->out_history = h->history[4]; }
        else         { s5 = t2; h->out_history = h->history[3]; }
        if (s5 < WORST) s5 = WORST;
        h->out_score = s5;
        best = s5;
    }
    s2 = add32(h->score[2], SEN(2));
    if (s2 > WORST) {
        t0 = add32(s4, TP(4, 4));
        t1 = add32(s3, TP(3, 4));
        t2 = add32(s2, TP(2, 4));
        if (t0 > t1) {
            if (t2 > t0) { s4 = t2; h->history[4] = h->history[2]; }
            else s4 = t0;
        }
        else {
            if (t2 > t1) { s4 = t2; h->history[4] = h->history[2]; }
            else { s4 = t1; h->history[4] = h->history[3]; }
        }
        if (s4 < WORST) s4 = WORST;
        if (s4 > best) best = s4;
        h->score[4] = s4;
    }
    s1 = add32(h->score[1], SEN(1));
    if (s1 > WORST) {
        t0 = add32(s3, TP(3, 3));
        t1 = add32(s2, TP(2, 3));
        t2 = add32(s1, TP(1, 3));
        if (t0 > t1) {
            if (t2 > t0) { s3 = t2; h->history[3] = h->history[1]; }
            else s3 = t0;
        }
        else {
            if (t2 > t1) { s3 = t2; h->history[3] = h->history[1]; }
            else { s3 = t1; h->history[3] = h->history[2]; }
        }
        if (s3 < WORST) s3 = WORST;
        if (s3 > best) best = s3;
        h->score[3] = s3;
    }
    s0 = add32(h->score[0], SEN(0));
    t0 = add32(s2, TP(2, 2));
    t1 = add32(s1, TP(1, 2));
    t2 = add32(s0, TP(0, 2));
    if (t0 > t1) {
        if (t2 > t0) { s2 = t2; h->history[2] = h->history[0]; }
        else s2 = t0;
    }
    else {
        if (t2 > t1) { s2 = t2; h->history[2] = h->history[0]; }
        else { s2 = t1; h->history[2] = h->history[1]; }
    }
    if (s2 < WORST) s2 = WORST;
    if (s2 > best) best = s2;
    h->score[2] = s2;

    t0 = add32(s1, TP(1, 1));
    t1 = add32(s0, TP(0, 1));
    if (t0 > t1) s1 = t0;
    else { s1 = t1; h->history[1] = h->history[0]; }
    if (s1 < WORST) s1 = WORST;
    if (s1 > best) best = s1;
    h->score[1] = s1;

    s0 = add32(s0, TP(0, 0));
    if (s0 < WORST) s0 = WORST;
    if (s0 > best) best = s0;
    h->score[0] = s0;

    h->bestscore = best;
    return best;
}

static int32_t
vit_5st_lr_mpx(const s3o_hmm_ctx_t *ctx, s3o_hmm_t *h)
{
    const int nd = 6, ne = 5;
    const int32_t *tp = ctx->tp + (size_t)h->tmatid * 5 * 6;
    int32_t *ssid = h->mpx_ssid;
    int32_t s5, s4, s3, s2, s1, s0, t2, t1, t0, best;

    if (ssid[4] == -1) s4 = t1 = WORST;
    else { s4 = add32(h->score[4], MSEN(4)); t1 = add32(s4, TP(4, 5)); }
    if (ssid[3] == -1) s3 = t2 = WORST;
    else { s3 = add32(h->score[3], MSEN(3)); t2 = add32(s3, TP(3, 5)); }
    if (t1 > t2) { s5 = t1; h->out_history = h->history[4]; }
    else         { s5 = t2; h->out_history = h->history[3]; }
    if (s5 < WORST) s5 = WORST;
    h->out_score = s5;
    best = s5;

    if (ssid[2] == -1) s2 = t2 = WORST;
    else { s2 = add32(h->score[2], MSEN(2)); t2 = add32(s2, TP(2, 4)); }
    t0 = t1 = WORST;
    if (s4 != WORST) t0 = add32(s4, TP(4, 4));
    if (s3 != WORST) t1 = add32(s3, TP(3, 4));
    if (t0 > t1) {
        if (t2 > t0) { s4 = t2; h->history[4] = h->history[2]; ssid[4] = ssid[2]; }
        else s4 = t0;
    }
    else {
        if (t2 > t1) { s4 = t2; h->history[4] = h->history[2]; ssid[4] = ssid[2]; }
        else { s4 = t1; h->history[4] = h->history[3]; ssid[4] = ssid[3]; }
    }
    if (s4 < WORST) s4 = WORST;
    if (s4 > best) best = s4;
    h->score[4] = s4;

    if (ssid[1] == -1) s1 = t2 = WORST;
    else { s1 = add32(h->score[1], MSEN(1)); t2 = add32(s1, TP(1, 3)); }
    t0 = t1 = WORST;
    if (s3 != WORST) t0 = add32(s3, TP(3, 3));
    if (s2 != WORST) t1 = add32(s2, TP(2, 3));
    if (t0 > t1) {
        if (t2 > t0) { s3 = t2; h->history[3] = h->history[1]; ssid[3] = ssid[1]; }
        else s3 = t0;
    }
    else {
        if (t2 > t1) { s3 = t2; h->history[3] = h->history[1]; ssid[3] = ssid[1]; }
        else { s3 = t1; h->history[3] = h->history[2]; ssid[3] = ssid[2]; }
    }
    if (s3 < WORST) s3 = WORST;
    if (s3 > best) best = s3;
    h->score[3] = s3;

    s0 = add32(h->score[0], MSEN(0));
    t0 = t1 = WORST;
    if (s2 != WORST) t0 = add32(s2, TP(2, 2));
    if (s1 != WORST) t1 = add32(s1, TP(1, 2));
    t2 = add32(s0, TP(0, 2));
    if (t0 > t1) {
        if (t2 > t0) { s2 = t2; h->history[2] = h->history[0]; ssid[2] = ssid[0]; }
        else s2 = t0;
    }
    else {
        if (t2 > t1) { s2 = t2; h->history[2] = h->history[0]; ssid[2] = ssid[0]; }
        else { s2 = t1; h->history[2] = h->history[1]; ssid[2] = ssid[1]; }
    }
    if (s2 < WORST) s2 = WORST;
    if (s2 > best) best = s2;
    h->score[2] = s2;

    t0 = WORST;
    if (s1 != WORST) t0 = add32(s1, TP(1, 1));
    t1 = add32(s0, TP(0, 1));
    if (t0 > t1) s1 = t0;
    else { s1 = t1; h->history[1] = h->history[0]; ssid[1] = ssid[0]; }
    if (s1 < WORST) s1 = WORST;
    if (s1 > best) best = s1;
    h->score[1] = s1;

    s0 = add32(s0, TP(0, 0));
    if (s0 < WORST) s0 = WORST;
    if (s0 > best) best = s0;
    h->score[0] = s0;

    h->bestscore = best;
    return best;
}

static int32_t
vit_3st_lr(const s3o_hmm_ctx_t *ctx, s3o_hmm_t *h)
{
    const int nd = 4;
    const int32_t *tp = ctx->tp + (size_t)h->tmatid * 3 * 4;
    const int16_t *sseq = ctx->sseq + (size_t)h->ssid * 3;
    int32_t s3, s2, s1, s0, t2, t1, t0, best;

    s2 = add32(h->score[2], SEN(2));
    s1 = add32(h->score[1], SEN(1));
    s0 = add32(h->score[0], SEN(0));

    t0 = t1 = best = WORST;
    t2 = INT_MIN;
    if (s2 > WORST) {
        t1 = add32(s2, TP(2, 3));
        t0 = add32(s2, TP(2, 2));
    }
    if (s1 > WORST && TP(1, 3) > WORST)
        t2 = add32(s1, TP(1, 3));
    if (t1 > t2) { s3 = t1; h->out_history = h->history[2]; }
    else         { s3 = t2; h->out_history = h->history[1]; }
    if (s3 < WORST) s3 = WORST;
    h->out_score = s3;
    best = s3;

    t1 = t2 = WORST;
    if (s1 > WORST) t1 = add32(s1, TP(1, 2));
    if (TP(0, 2) > WORST) t2 = add32(s0, TP(0, 2));
    if (t0 > t1) {
        if (t2 > t0) { s2 = t2; h->history[2] = h->history[0]; }
        else s2 = t0;
    }
    else {
        if (t2 > t1) { s2 = t2; h->history[2] = h->history[0]; }
        else { s2 = t1; h->history[2] = h->history[1]; }
    }
    if (s2 < WORST) s2 = WORST;
    if (s2 > best) best = s2;
    h->score[2] = s2;

    t0 = t1 = WORST;
    if (s1 > WORST) t0 = add32(s1, TP(1, 1));
    if (s0 > WORST) t1 = add32(s0, TP(0, 1));
    if (t0 > t1) s1 = t0;
    else { s1 = t1; h->history[1] = h->history[0]; }
    if (s1 < WORST) s1 = WORST;
    if (s1 > best) best = s1;
    h->score[1] = s1;

    s0 = add32(s0, TP(0, 0));
    if (s0 < WORST) s0 = WORST;
    if (s0 > best) best = s0;
    h->score[0] = s0;

    h->bestscore = best;
    return best;
}

static int32_t
vit_3st_lr_mpx(const s3o_hmm_ctx_t *ctx, s3o_hmm_t *h)
{
    const int nd = 4, ne = 3;
    const int32_t *tp = ctx->tp + (size_t)h->tmatid * 3 * 4;
    int32_t *ssid = h->mpx_ssid;
    int32_t s3, s2, s1, s0, t2, t1, t0, best;

    t2 = INT_MIN;
    if (ssid[2] == -1) s2 = t1 = WORST;
    else {
        s2 = add32(h->score[2], MSEN(2));
        if (s2 < WORST) s2 = WORST;
        t1 = add32(s2, TP(2, 3));
    }
    if (ssid[1] == -1) s1 = WORST;
    else {
        s1 = add32(h->score[1], MSEN(1));
        if (s1 < WORST) s1 = WORST;
        t2 = add32(s1, TP(1, 3));
    }
    if (t1 > t2) { s3 = t1; h->out_history = h->history[2]; }
    else         { s3 = t2; h->out_history = h->history[1]; }
    if (s3 < WORST) s3 = WORST;
    h->out_score = s3;
    best = s3;

    s0 = add32(h->score[0], MSEN(0));
    if (s0 < WORST) s0 = WORST;

    t0 = t1 = WORST;
    if (s2 != WORST) t0 = add32(s2, TP(2, 2));
    if (s1 != WORST) t1 = add32(s1, TP(1, 2));
    if (TP(0, 2) > WORST) t2 = add32(s0, TP(0, 2));
    if (t0 > t1) {
        if (t2 > t0) { s2 = t2; h->history[2] = h->history[0]; ssid[2] = ssid[0]; }
        else s2 = t0;
    }
    else {
        if (t2 > t1) { s2 = t2; h->history[2] = h->history[0]; ssid[2] = ssid[0]; }
        else { s2 = t1; h->history[2] = h->history[1]; ssid[2] = ssid[1]; }
    }
    if (s2 < WORST) s2 = WORST;
    if (s2 > best) best = s2;
    h->score[2] = s2;

    t0 = WORST;
    if (s1 != WORST) t0 = add32(s1, TP(1, 1));
    t1 = add32(s0, TP(0, 1));
    if (t0 > t1) s1 = t0;
    else { s1 = t1; h->history[1] = h->history[0]; ssid[1] = ssid[0]; }
    if (s1 < WORST) s1 = WORST;
    if (s1 > best) best = s1;
    h->score[1] = s1;

    s0 = add32(s0, TP(0, 0));
    if (s0 < WORST) s0 = WORST;
    if (s0 > best) best = s0;
    h->score[0] = s0;

    h->bestscore = best;
    return best;
}

/* hmm_senscr(h,st), hmm.h:223-226 */
static int32_t
any_senscr(const s3o_hmm_ctx_t *ctx, const s3o_hmm_t *h, int st)
{
    int32_t ssid = h->mpx ? h->mpx_ssid[st] : h->ssid;
    if (ssid == -1)
        return S3O_LOGPROB_ZERO;
    return ctx->senscore[ctx->sseq[(size_t)ssid * ctx->n_emit_state + st]];
}

static int32_t
vit_anytopo(const s3o_hmm_ctx_t *ctx, s3o_hmm_t *h)
{
    const int ne = ctx->n_emit_state, nd = ne + 1;
    const int32_t *tp = ctx->tp + (size_t)h->tmatid * ne * nd;
    int32_t st_sen_scr[S3O_MAX_HMM_NSTATE];
    int32_t to, from, bestfrom, newscr, scr, bestscr;

    st_sen_scr[0] = add32(h->score[0], any_senscr(ctx, h, 0));
    for (from = 1; from < ne; ++from) {
        if ((st_sen_scr[from] = add32(h->score[from], any_senscr(ctx, h, from))) < WORST)
            st_sen_scr[from] = WORST;
    }
    to = ne;
    scr = WORST;
    bestfrom = -1;
    for (from = to - 1; from >= 0; --from) {
        if ((TP(from, to) > WORST) &&
            ((newscr = add32(st_sen_scr[from], TP(from, to))) > scr)) {
            scr = newscr;
            bestfrom = from;
        }
    }
    h->out_score = scr;
    if (bestfrom >= 0)
        h->out_history = h->history[bestfrom];
    bestscr = scr;

    for (to = ne - 1; to >= 0; --to) {
        scr = (TP(to, to) > WORST) ? add32(st_sen_scr[to], TP(to, to)) : WORST;
        bestfrom = -1;
        for (from = to - 1; from >= 0; --from) {
            if ((TP(from, to) > WORST) &&
                ((newscr = add32(st_sen_scr[from], TP(from, to))) > scr)) {
                scr = newscr;
                bestfrom = from;
            }
        }
        h->score[to] = scr;
        if (bestfrom >= 0)
            h->history[to] = h->history[bestfrom];
        if (bestfrom >= 0 && h->mpx)
            h->mpx_ssid[to] = h->mpx_ssid[bestfrom];
        if (bestscr < scr)
            bestscr = scr;
    }
    h->bestscore = bestscr;
    return bestscr;
}

int32_t
s3o_hmm_vit_eval(const s3o_hmm_ctx_t *ctx, s3o_hmm_t *h)
{
    if (h->mpx) {
        if (ctx->n_emit_state == 5) return vit_5st_lr_mpx(ctx, h);
        if (ctx->n_emit_state == 3) return vit_3st_lr_mpx(ctx, h);
        return vit_anytopo(ctx, h);
    }
    if (ctx->n_emit_state == 5) return vit_5st_lr(ctx, h);
    if (ctx->n_emit_state == 3) return vit_3st_lr(ctx, h);
    return vit_anytopo(ctx, h);
}
