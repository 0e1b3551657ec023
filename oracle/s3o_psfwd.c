/*
 * oracle/s3o_psfwd.c -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY (see s3o.h, s3o_psfwd.h).
 *
 * pocketsphinx's first pass restated on flat arrays, one utterance at a time, sequential, in the reference's
 * order of operations (the order is the tie-break contract: active lists, candidate lists and backpointer
 * entries are created in it).  File:line references are to /root/reference/pocketsphinx/src/libpocketsphinx/.
 *
 * What differs in FORM from the reference: channels live in one array (roots, interior channels, the
 * single-phone words' channels, then every word's right-context channels), linked lists are index ranges,
 * "allocated" is a flag, the language model is the flat trigram of the descriptor.  What may not differ is
 * any value or any order.
 */
#include <limits.h>
#include <stdlib.h>
#include <string.h>
#include "s3o_psfwd.h"

#define WORST S3O_PS_WORST_SCORE
#define NO_BP S3O_PS_NO_BP

static inline int32_t add32(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static inline int32_t sub32(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }

/* ------------------------------------------------------------------ */
/* hmm.c                                                              */
/* ------------------------------------------------------------------ */
/* hmm_clear :189-204 */
static void
hmm_clear(s3o_pshmm_t *h, int n)
{
    for (int i = 0; i < n; i++) { h->score[i] = WORST; h->hist[i] = -1; }
    h->out_score = WORST; h->out_hist = -1; h->bestscore = WORST; h->frame = -1;
}
/* hmm_clear_scores :176-187 */
static void
hmm_clear_scores(s3o_pshmm_t *h, int n)
{
    for (int i = 0; i < n; i++) h->score[i] = WORST;
    h->out_score = WORST; h->bestscore = WORST;
}
/* hmm_init :88-108 */
static void
hmm_init(s3o_pshmm_t *h, const s3o_psfwd_desc_t *d, int mpx, int ssid, int tmatid)
{
    h->mpx = (uint8_t)mpx;
    if (mpx) {
        h->ssid = S3O_PS_BAD_SSID;
        h->senid[0] = (uint16_t)ssid;
        for (int i = 1; i < d->n_emit; i++) h->senid[i] = S3O_PS_BAD_SSID;
    }
    else {
        h->ssid = (uint16_t)ssid;
        for (int i = 0; i < d->n_emit; i++) h->senid[i] = d->sseq[ssid * d->n_emit + i];
    }
    h->tmatid = (int16_t)tmatid;
    hmm_clear(h, d->n_emit);
}
/* hmm_enter :206-212 */
static inline void
hmm_enter(s3o_pshmm_t *h, int32_t score, int32_t hist, int32_t frame)
{
    h->score[0] = score; h->hist[0] = hist; h->frame = frame;
}
/* hmm_normalize :214-225 */
static void
hmm_normalize(s3o_pshmm_t *h, int n, int32_t bestscr)
{
    for (int i = 0; i < n; i++)
        if (h->score[i] > WORST) h->score[i] = sub32(h->score[i], bestscr);
    if (h->out_score > WORST) h->out_score = sub32(h->out_score, bestscr);
}

/*
 * hmm_vit_eval :789-809 for the two hard-wired topologies: _3st_lr :532-609, _3st_lr_mpx :612-712,
 * _5st_lr :230-345, _5st_lr_mpx :350-528.  S[k] = the state's score plus its (negated) senone score, taken
 * before any state is updated; transitions are -tp.  The plain and the multiplexed versions differ in their
 * guards (plain: a state is updated only if the state TWO below it is alive; multiplexed: a dead source
 * contributes WORST_SCORE instead of a sum) and the 3-state versions honour skip arcs only where the matrix
 * has them, with the variable t2 carried from one stage into the next (:547, :565-567).
 */
int32_t
s3o_ps_hmm_vit_eval(s3o_pshmm_t *h, int32_t n, const uint8_t *tpm, const uint16_t *sseq, const int16_t *senscr)
{
    const uint8_t *tp = tpm + (size_t)h->tmatid * n * (n + 1);
    const int mpx = h->mpx;
    int32_t S[5] = { 0, 0, 0, 0, 0 }, best = WORST, t0, t1, t2, v;
    int bad[5] = { 0, 0, 0, 0, 0 };
#define TP(i, j) (-(int32_t)tp[(i) * (n + 1) + (j)])
    for (int k = 0; k < n; k++) {
        if (mpx) {
            if (k > 0 && h->senid[k] == S3O_PS_BAD_SSID) { bad[k] = 1; S[k] = WORST; }
            else S[k] = add32(h->score[k], -(int32_t)senscr[sseq[h->senid[k] * n + k]]);
        }
        else S[k] = add32(h->score[k], -(int32_t)senscr[h->senid[k]]);
    }
    if (n == 3) {
        t2 = INT_MIN;
        if (mpx || S[1] > WORST) {
            t1 = bad[2] ? WORST : add32(S[2], TP(2, 3));
            if (bad[1]) t2 = WORST;
            else if (TP(1, 3) > S3O_PS_TMAT_WORST) t2 = add32(S[1], TP(1, 3));
            if (t1 > t2) { v = t1; h->out_hist = h->hist[2]; }
            else { v = t2; h->out_hist = h->hist[1]; }
            if (v < WORST) v = WORST;
            h->out_score = v; best = v;
        }
        if (mpx) {
            t0 = S[2] != WORST ? add32(S[2], TP(2, 2)) : WORST;
            t1 = S[1] != WORST ? add32(S[1], TP(1, 2)) : WORST;
        }
        else { t0 = add32(S[2], TP(2, 2)); t1 = add32(S[1], TP(1, 2)); }
        if (TP(0, 2) > S3O_PS_TMAT_WORST) t2 = add32(S[0], TP(0, 2));
        if (t0 > t1) {
            if (t2 > t0) { v = t2; h->hist[2] = h->hist[0]; if (mpx) h->senid[2] = h->senid[0]; }
            else v = t0;
        }
        else {
            if (t2 > t1) { v = t2; h->hist[2] = h->hist[0]; if (mpx) h->senid[2] = h->senid[0]; }
            else { v = t1; h->hist[2] = h->hist[1]; if (mpx) h->senid[2] = h->senid[1]; }
        }
        if (v < WORST) v = WORST;
        if (v > best) best = v;
        h->score[2] = v;
    }
    else {
        /* exit state */
        if (mpx || S[3] > WORST) {
            t1 = bad[4] ? WORST : add32(S[4], TP(4, 5));
            t2 = bad[3] ? WORST : add32(S[3], TP(3, 5));
            if (t1 > t2) { v = t1; h->out_hist = h->hist[4]; }
            else { v = t2; h->out_hist = h->hist[3]; }
            if (v < WORST) v = WORST;
            h->out_score = v; best = v;
        }
        /* states 4, 3, 2: sources j, j-1, j-2 */
        for (int j = 4; j >= 2; j--) {
            if (!mpx && j > 2 && !(S[j - 2] > WORST)) continue;
            if (mpx) {
                t0 = S[j] != WORST ? add32(S[j], TP(j, j)) : WORST;
                t1 = S[j - 1] != WORST ? add32(S[j - 1], TP(j - 1, j)) : WORST;
                t2 = (j > 2 && bad[j - 2]) ? WORST : add32(S[j - 2], TP(j - 2, j));
            }
            else {
                t0 = add32(S[j], TP(j, j)); t1 = add32(S[j - 1], TP(j - 1, j)); t2 = add32(S[j - 2], TP(j - 2, j));
            }
            if (t0 > t1) {
                if (t2 > t0) { v = t2; h->hist[j] = h->hist[j - 2]; if (mpx) h->senid[j] = h->senid[j - 2]; }
                else v = t0;
            }
            else {
                if (t2 > t1) { v = t2; h->hist[j] = h->hist[j - 2]; if (mpx) h->senid[j] = h->senid[j - 2]; }
                else { v = t1; h->hist[j] = h->hist[j - 1]; if (mpx) h->senid[j] = h->senid[j - 1]; }
            }
            if (v < WORST) v = WORST;
            if (v > best) best = v;
            h->score[j] = v;
        }
    }
    /* state 1 */
    t0 = mpx ? (S[1] != WORST ? add32(S[1], TP(1, 1)) : WORST) : add32(S[1], TP(1, 1));
    t1 = add32(S[0], TP(0, 1));
    if (t0 > t1) v = t0;
    else { v = t1; h->hist[1] = h->hist[0]; if (mpx) h->senid[1] = h->senid[0]; }
    if (v < WORST) v = WORST;
    if (v > best) best = v;
    h->score[1] = v;
    /* state 0 */
    v = add32(S[0], TP(0, 0));
    if (v < WORST) v = WORST;
    if (v > best) best = v;
    h->score[0] = v;
    h->bestscore = best;
    return best;
#undef TP
}

/* ------------------------------------------------------------------ */
/* the language model: ngram_tg_score (sphinxbase ngram_model.c:555) through ngram_model_set_score         */
/* (ngram_model_set.c:709-756, one current model) to lm3g_tg_score / lm3g_bg_score (lm3g_templates.c)     */
/* ------------------------------------------------------------------ */
static int32_t
find_wid(const int32_t *wids, int32_t b, int32_t e, int32_t w)
{
    /* find_bg / find_tg (lm3g_templates.c:46-66, :134-152), step for step: bisection while the segment has more
     * than 16 entries, then a linear scan.  The steps matter: LM files whose runs are not sorted by word id exist
     * (model/lm/zh_CN/gigatdt.5000.DMP), and what the reference finds in them is what these steps find. */
    int32_t i;
    while (e - b > 16) {
        i = (b + e) >> 1;
        if (wids[i] < w) b = i + 1;
        else if (wids[i] > w) e = i;
        else return i;
    }
    for (i = b; i < e && wids[i] != w; i++) ;
    return i < e ? i : -1;
}
static int32_t
lm_bg_score(const s3o_psfwd_desc_t *d, int32_t lw1, int32_t lw2)
{
    int32_t i;
    if (lw1 < 0 || d->lm_order < 2) return d->ug_prob[lw2];                         /* :73-78 */
    i = find_wid(d->bg_wid, d->ug_firstbg[lw1], d->ug_firstbg[lw1 + 1], lw2);
    if (i >= 0) return d->bg_prob[i];
    return add32(d->ug_bowt[lw1], d->ug_prob[lw2]);                                 /* :92 */
}
static int32_t
lm_tg_score(const s3o_psfwd_desc_t *d, int32_t lw1, int32_t lw2, int32_t lw3)
{
    int32_t b, bowt = 0, tb = 0, te = 0, i;
    if (d->lm_order < 3 || lw1 < 0 || lw2 < 0) return lm_bg_score(d, lw2, lw3);    /* :165-166 */
    b = find_wid(d->bg_wid, d->ug_firstbg[lw1], d->ug_firstbg[lw1 + 1], lw2);      /* load_tginfo :98-131 */
    if (b >= 0) { bowt = d->bg_bowt[b]; tb = d->bg_firsttg[b]; te = d->bg_firsttg[b + 1]; }
    i = find_wid(d->tg_wid, tb, te, lw3);
    if (i >= 0) return d->tg_prob[i];
    return add32(bowt, lm_bg_score(d, lw2, lw3));                                   /* :192 */
}
int32_t
s3o_psfwd_tg_score(const s3o_psfwd_t *s, int32_t w3, int32_t w2, int32_t w1)
{
    const s3o_psfwd_desc_t *d = &s->d;
    int32_t m3 = d->w_lmwid[w3], m2 = w2 < 0 ? -1 : d->w_lmwid[w2], m1 = w1 < 0 ? -1 : d->w_lmwid[w1];
    /* a class word: w_lmwid is its class's tag word, the in-class weight is added to the tag's score (ngram_model.c:505-520) */
    const int32_t cw = d->w_lmcw ? d->w_lmcw[w3] : 0;
    if (m3 < 0) return d->lm_zero;                  /* ngram_ng_score, ngram_model.c:501-502 */
    if (cw == 1) return d->lm_zero;                 /* "not found in class" :509-510 */
    if (d->lm_order < 2) return add32(d->ug_prob[m3], cw);     /* history truncated to n - 1 words, ngram_model_set.c:719-720 */
    if (d->lm_order < 3) return add32(lm_bg_score(d, m2, m3), cw);
    return add32(lm_tg_score(d, m1, m2, m3), cw);
}

/* ------------------------------------------------------------------ */
s3o_psfwd_t *
s3o_psfwd_init(const s3o_psfwd_desc_t *d)
{
    s3o_psfwd_t *s = calloc(1, sizeof(*s));
    int32_t n_rc = d->w_rc_off[d->n_words], i;
    s->d = *d;
    s->n_ch = d->n_root + d->n_nonroot;
    s->sp_base = s->n_ch;
    s->rc_base = s->sp_base + d->n_1ph;
    s->n_hmm = s->rc_base + n_rc;
    s->hmm = calloc(s->n_hmm ? s->n_hmm : 1, sizeof(*s->hmm));
    s->w_sp = malloc(sizeof(int32_t) * d->n_words);
    for (i = 0; i < d->n_words; i++) s->w_sp[i] = -1;
    /* init_search_tree :66-148, create_search_tree :173-317 */
    for (i = 0; i < d->n_root; i++) hmm_init(&s->hmm[i], d, 1, d->root_ssid0[i], d->root_tmat[i]);
    for (i = 0; i < d->n_nonroot; i++) hmm_init(&s->hmm[d->n_root + i], d, 0, d->nr_ssid[i], d->nr_tmat[i]);
    for (i = 0; i < d->n_1ph; i++) {
        hmm_init(&s->hmm[s->sp_base + i], d, 1, d->sp_ssid0[i], d->sp_tmat[i]);
        s->w_sp[d->sp_wid[i]] = i;
    }
    for (i = 0; i < 2; i++) {
        s->acl[i] = malloc(sizeof(int32_t) * (d->n_nonroot + 1));
        s->awl[i] = malloc(sizeof(int32_t) * (d->n_words + 1));
    }
    s->word_active = calloc(d->n_words, 1);
    s->cand = calloc(d->n_words + 1, sizeof(*s->cand));
    s->ltrans = calloc(d->n_words, sizeof(*s->ltrans));
    s->bestrc = calloc(d->n_ci, sizeof(*s->bestrc));
    s->pl = calloc(d->n_ci > 0 ? d->n_ci : 1, 4);
    s->word_lat_idx = malloc(sizeof(int32_t) * d->n_words);
    s->bp_cap = 5000;
    s->bp_frame = malloc(4 * s->bp_cap); s->bp_wid = malloc(4 * s->bp_cap); s->bp_bp = malloc(4 * s->bp_cap);
    s->bp_score = malloc(4 * s->bp_cap); s->bp_sidx = malloc(4 * s->bp_cap); s->bp_realwid = calloc(s->bp_cap, 4);
    s->bp_valid = malloc(s->bp_cap);
    s->bss_cap = s->bp_cap * 20;
    s->bss = malloc(4 * (size_t)s->bss_cap);
    s->n_frame_alloc = 256;
    s->bp_table_idx = calloc(s->n_frame_alloc + 2, 4);
    return s;
}

/* a freshly initialised search (init_search_tree + create_search_tree state): what a NEW ps_decoder_t has.  The
 * reference's start does not restore this: channel frames, histories and multiplexed ids survive from one utterance
 * to the next (hmm_clear_scores leaves them), which is observable (a stale frame number equal to nf loses an entry
 * in prune_nonroot_chan :833-837; stale multiplexed ids activate senones). */
void
s3o_psfwd_reset(s3o_psfwd_t *s)
{
    const s3o_psfwd_desc_t *d = &s->d;
    int32_t i;
    for (i = 0; i < d->n_root; i++) hmm_init(&s->hmm[i], d, 1, d->root_ssid0[i], d->root_tmat[i]);
    for (i = 0; i < d->n_nonroot; i++) hmm_init(&s->hmm[d->n_root + i], d, 0, d->nr_ssid[i], d->nr_tmat[i]);
    for (i = 0; i < d->n_1ph; i++) hmm_init(&s->hmm[s->sp_base + i], d, 1, d->sp_ssid0[i], d->sp_tmat[i]);
    for (i = s->rc_base; i < s->n_hmm; i++) { s->hmm[i].alloc = 0; hmm_clear(&s->hmm[i], d->n_emit); }
    memset(s->word_active, 0, d->n_words);
    memset(s->ltrans, 0, sizeof(*s->ltrans) * d->n_words);
    memset(s->bp_realwid, 0, 4 * (size_t)s->bp_cap);
}

void
s3o_psfwd_free(s3o_psfwd_t *s)
{
    if (!s) return;
    free(s->hmm); free(s->w_sp);
    for (int i = 0; i < 2; i++) { free(s->acl[i]); free(s->awl[i]); }
    free(s->word_active); free(s->cand); free(s->ltrans); free(s->bestrc); free(s->word_lat_idx);
    free(s->cand_sf_ef); free(s->cand_sf_cand);
    free(s->bp_frame); free(s->bp_wid); free(s->bp_bp); free(s->bp_score); free(s->bp_sidx); free(s->bp_realwid);
    free(s->bp_valid); free(s->bss); free(s->bp_table_idx); free(s->pl);
    free(s);
}

/* what the next step adds at its transitions: phone_loop_search_score(pls, ci) of the decoder's phone loop
 * (phone_loop_search.h:103-105), which ps_search_forward steps in front of this search (pocketsphinx.c:704-712) */
void
s3o_psfwd_set_lookahead(s3o_psfwd_t *s, const int32_t *pl)
{
    for (int32_t i = 0; i < s->d.n_ci; i++) s->pl[i] = pl ? pl[i] : 0;
}

/* ngram_fwdtree_start :464-507 */
void
s3o_psfwd_start(s3o_psfwd_t *s)
{
    const s3o_psfwd_desc_t *d = &s->d;
    int32_t i;
    s->st_n_root_chan_eval = s->st_n_nonroot_chan_eval = s->st_n_last_chan_eval = 0;
    s->st_n_word_lastchan_eval = s->st_n_lastphn_cand_utt = s->st_n_senone_active_utt = 0;
    s->bpidx = 0; s->bss_head = 0;
    for (i = 0; i < d->n_words; i++) s->word_lat_idx[i] = NO_BP;
    s->n_acl[0] = s->n_acl[1] = 0;
    s->n_awl[0] = s->n_awl[1] = 0;
    s->best_score = 0; s->renormalized = 0;
    for (i = 0; i < d->n_words; i++) s->ltrans[i].sf = -1;
    s->n_frame = 0;
    for (i = 0; i < d->n_1ph; i++) hmm_clear(&s->hmm[s->sp_base + i], d->n_emit);
    i = s->w_sp[d->start_wid];
    hmm_clear(&s->hmm[s->sp_base + i], d->n_emit);
    hmm_enter(&s->hmm[s->sp_base + i], 0, NO_BP, 0);
}

/* acmod_activate_hmm, acmod.c:1173-1214 */
static void
activate(const s3o_psfwd_desc_t *d, const s3o_pshmm_t *h, uint8_t *flags)
{
    for (int i = 0; i < d->n_emit; i++) {
        if (h->mpx) { if (h->senid[i] != S3O_PS_BAD_SSID) flags[d->sseq[h->senid[i] * d->n_emit + i]] = 1; }
        else flags[h->senid[i]] = 1;
    }
}

/* compute_sen_active :513-552 */
int32_t
s3o_psfwd_sen_active(s3o_psfwd_t *s, int32_t frame_idx, uint8_t *flags)
{
    const s3o_psfwd_desc_t *d = &s->d;
    int32_t i, n = 0;
    memset(flags, 0, d->n_sen);
    for (i = 0; i < d->n_root; i++)
        if (s->hmm[i].frame == frame_idx) activate(d, &s->hmm[i], flags);
    for (i = 0; i < s->n_acl[frame_idx & 1]; i++) activate(d, &s->hmm[s->acl[frame_idx & 1][i]], flags);
    for (i = 0; i < s->n_awl[frame_idx & 1]; i++) {
        int32_t w = s->awl[frame_idx & 1][i];
        for (int32_t c = d->w_rc_off[w]; c < d->w_rc_off[w + 1]; c++)
            if (s->hmm[s->rc_base + c].alloc) activate(d, &s->hmm[s->rc_base + c], flags);
    }
    for (i = 0; i < d->n_1ph; i++)
        if (s->hmm[s->sp_base + i].frame == frame_idx) activate(d, &s->hmm[s->sp_base + i], flags);
    for (i = 0; i < d->n_sen; i++) n += flags[i];
    return n;
}

/* renormalize_scores :555-592 */
static void
renormalize(s3o_psfwd_t *s, int32_t frame_idx, int32_t norm)
{
    const s3o_psfwd_desc_t *d = &s->d;
    int32_t i;
    for (i = 0; i < d->n_root; i++)
        if (s->hmm[i].frame == frame_idx) hmm_normalize(&s->hmm[i], d->n_emit, norm);
    for (i = 0; i < s->n_acl[frame_idx & 1]; i++) hmm_normalize(&s->hmm[s->acl[frame_idx & 1][i]], d->n_emit, norm);
    for (i = 0; i < s->n_awl[frame_idx & 1]; i++) {
        int32_t w = s->awl[frame_idx & 1][i];
        for (int32_t c = d->w_rc_off[w]; c < d->w_rc_off[w + 1]; c++)
            if (s->hmm[s->rc_base + c].alloc) hmm_normalize(&s->hmm[s->rc_base + c], d->n_emit, norm);
    }
    for (i = 0; i < d->n_1ph; i++)
        if (s->hmm[s->sp_base + i].frame == frame_idx) hmm_normalize(&s->hmm[s->sp_base + i], d->n_emit, norm);
    s->renormalized = 1;
}

#define EVAL(h) s3o_ps_hmm_vit_eval((h), d->n_emit, d->tp, d->sseq, s->senscr)

/* evaluate_channels :694-706 = eval_root_chan :595-610, eval_nonroot_chan :613-631, eval_word_chan :634-691 */
static void
evaluate_channels(s3o_psfwd_t *s, int32_t frame_idx)
{
    const s3o_psfwd_desc_t *d = &s->d;
    int32_t i, best = WORST, bs, k = 0, j = 0;
    for (i = 0; i < d->n_root; i++)
        if (s->hmm[i].frame == frame_idx) {
            int32_t sc = EVAL(&s->hmm[i]);
            if (sc > best) best = sc;
            s->st_n_root_chan_eval++;
        }
    s->best_score = best;
    bs = WORST;
    s->st_n_nonroot_chan_eval += s->n_acl[frame_idx & 1];
    for (i = 0; i < s->n_acl[frame_idx & 1]; i++) {
        int32_t sc = EVAL(&s->hmm[s->acl[frame_idx & 1][i]]);
        if (sc > bs) bs = sc;
    }
    if (bs > s->best_score) s->best_score = bs;
    bs = WORST;
    for (i = 0; i < s->n_awl[frame_idx & 1]; i++) {
        int32_t w = s->awl[frame_idx & 1][i];
        s->word_active[w] = 0;
        for (int32_t c = d->w_rc_off[w]; c < d->w_rc_off[w + 1]; c++)
            if (s->hmm[s->rc_base + c].alloc) {
                int32_t sc = EVAL(&s->hmm[s->rc_base + c]);
                if (sc > bs) bs = sc;
                k++;
            }
    }
    for (i = 0; i < d->n_1ph; i++) {
        s3o_pshmm_t *h = &s->hmm[s->sp_base + i];
        int32_t sc;
        if (h->frame < frame_idx) continue;
        sc = EVAL(h);
        if (sc > bs && d->sp_wid[i] != d->finish_wid) bs = sc;
        j++;
    }
    s->st_n_last_chan_eval += k + j;
    s->st_n_nonroot_chan_eval += k + j;
    s->st_n_word_lastchan_eval += s->n_awl[frame_idx & 1] + j;
    if (bs > s->best_score) s->best_score = bs;
    s->last_phone_best_score = bs;
}

/* the successor transitions shared by prune_root_chan :737-784 and prune_nonroot_chan :822-863.  With the phone loop
 * (pls != NULL: d->pl_window > 0) the tests in front of the two loops are not made and every transition adds
 * phone_loop_search_score of the phone it enters (:745-753, :765-780, :824-840, :847-861); without it s->pl is all zeros
 * and the tests inside the loops are the ones in front of them. */
static void
phone_transitions(s3o_psfwd_t *s, int32_t c, int32_t frame_idx, int is_root, int32_t **nacl)
{
    const s3o_psfwd_desc_t *d = &s->d;
    s3o_pshmm_t *h = &s->hmm[c];
    const int32_t nf = frame_idx + 1;
    const int32_t newphone_thresh = add32(s->best_score, d->pbeam), lastphn_thresh = add32(s->best_score, d->lpbeam);
    const int32_t newphone_score = add32(h->out_score, d->pip);
    const int pls = d->pl_window > 0;
    if (pls || newphone_score > newphone_thresh)
        for (int32_t e = d->ch_child_off[c]; e < d->ch_child_off[c + 1]; e++) {
            const int32_t x = d->ch_child[e];
            s3o_pshmm_t *nh = &s->hmm[x];
            const int32_t pl_score = add32(newphone_score, s->pl[d->nr_ci[x - d->n_root]]);
            if (pl_score > newphone_thresh && (nh->frame < frame_idx || pl_score > nh->score[0])) {
                if (is_root || nh->frame != nf) *((*nacl)++) = x;
                hmm_enter(nh, pl_score, h->out_hist, nf);
            }
        }
    if (pls || newphone_score > lastphn_thresh)
        for (int32_t e = d->ch_pen_off[c]; e < d->ch_pen_off[c + 1]; e++) {
            const int32_t w = d->ch_pen_wid[e], pl_score = add32(newphone_score, s->pl[d->w_last_ci[w]]);
            if (pl_score > lastphn_thresh) {
                s3o_pscand_t *cp = &s->cand[s->n_cand++];
                cp->wid = w;
                cp->score = sub32(pl_score, d->nwpen);
                cp->bp = h->out_hist;
            }
        }
}

/* prune_root_chan :714-788 */
static void
prune_root_chan(s3o_psfwd_t *s, int32_t frame_idx)
{
    const s3o_psfwd_desc_t *d = &s->d;
    const int32_t nf = frame_idx + 1, thresh = add32(s->best_score, s->dynamic_beam);
    int32_t *nacl = s->acl[nf & 1];
    for (int32_t i = 0; i < d->n_root; i++) {
        s3o_pshmm_t *h = &s->hmm[i];
        if (h->frame < frame_idx) continue;
        if (h->bestscore > thresh) {
            h->frame = nf;
            phone_transitions(s, i, frame_idx, 1, &nacl);
        }
    }
    s->n_acl[nf & 1] = (int32_t)(nacl - s->acl[nf & 1]);
}

/* prune_nonroot_chan :794-870 */
static void
prune_nonroot_chan(s3o_psfwd_t *s, int32_t frame_idx)
{
    const s3o_psfwd_desc_t *d = &s->d;
    const int32_t nf = frame_idx + 1, thresh = add32(s->best_score, s->dynamic_beam);
    int32_t *nacl = s->acl[nf & 1] + s->n_acl[nf & 1];
    for (int32_t i = 0; i < s->n_acl[frame_idx & 1]; i++) {
        int32_t c = s->acl[frame_idx & 1][i];
        s3o_pshmm_t *h = &s->hmm[c];
        if (h->bestscore > thresh) {
            if (h->frame != nf) { h->frame = nf; *(nacl++) = c; }
            phone_transitions(s, c, frame_idx, 0, &nacl);
        }
        else if (h->frame != nf) hmm_clear_scores(h, d->n_emit);
    }
    s->n_acl[nf & 1] = (int32_t)(nacl - s->acl[nf & 1]);
}

/* ngram_search_exit_score, ngram_search.c:601-622 */
int32_t
s3o_psfwd_exit_score(const s3o_psfwd_t *s, int32_t bp, int32_t rcphone)
{
    const s3o_psfwd_desc_t *d = &s->d;
    const int32_t w = s->bp_wid[bp];
    if (d->w_last2_ci[w] == -1) return s->bss[s->bp_sidx[bp]];
    return s->bss[s->bp_sidx[bp] + d->rc_cimap[d->w_rc_row[w] * d->n_ci + rcphone]];
}

static inline int32_t
prev_real_wid(const s3o_psfwd_t *s, int32_t bp)
{
    return s->bp_bp[bp] == NO_BP ? -1 : s->bp_realwid[s->bp_bp[bp]];
}

/* ngram_search_alloc_all_rc, ngram_search.c:541-588: every right context of the word's last phone has a channel */
static void
alloc_all_rc(s3o_psfwd_t *s, int32_t w)
{
    const s3o_psfwd_desc_t *d = &s->d;
    for (int32_t c = d->w_rc_off[w]; c < d->w_rc_off[w + 1]; c++) {
        s3o_pshmm_t *h = &s->hmm[s->rc_base + c];
        if (!h->alloc) { hmm_init(h, d, 0, d->rc_ssid[c], d->w_rc_tmat[w]); h->alloc = 1; }
    }
}

/* last_phone_transition :877-1030 */
static void
last_phone_transition(s3o_psfwd_t *s, int32_t frame_idx)
{
    const s3o_psfwd_desc_t *d = &s->d;
    const int32_t nf = frame_idx + 1;
    int32_t *nawl = s->awl[nf & 1];
    int32_t i, j, n_cand_sf = 0, bestscore, thresh;
    s->st_n_lastphn_cand_utt += s->n_cand;
    for (i = 0; i < s->n_cand; i++) {
        s3o_pscand_t *cp = &s->cand[i];
        int32_t ef;
        if (cp->bp == -1) continue;
        ef = s->bp_frame[cp->bp];
        cp->score = sub32(cp->score, s3o_psfwd_exit_score(s, cp->bp, d->w_first_ci[cp->wid]));
        if (s->ltrans[cp->wid].sf != ef + 1) {
            for (j = 0; j < n_cand_sf; j++)
                if (s->cand_sf_ef[j] == ef) break;
            if (j < n_cand_sf) cp->next = s->cand_sf_cand[j];
            else {
                if (n_cand_sf >= s->cand_sf_alloc) {
                    s->cand_sf_alloc += 32;
                    s->cand_sf_ef = realloc(s->cand_sf_ef, 4 * s->cand_sf_alloc);
                    s->cand_sf_cand = realloc(s->cand_sf_cand, 4 * s->cand_sf_alloc);
                }
                j = n_cand_sf++;
                cp->next = -1;
                s->cand_sf_ef[j] = ef;
            }
            s->cand_sf_cand[j] = i;
            s->ltrans[cp->wid].dscr = WORST;
            s->ltrans[cp->wid].sf = ef + 1;
        }
    }
    for (i = 0; i < n_cand_sf; i++) {
        int32_t bp = s->bp_table_idx[1 + s->cand_sf_ef[i]], bplast = s->bp_table_idx[1 + s->cand_sf_ef[i] + 1] - 1;
        for (; bp <= bplast; bp++) {
            if (!s->bp_valid[bp]) continue;
            for (j = s->cand_sf_cand[i]; j >= 0; j = s->cand[j].next) {
                s3o_pscand_t *cp = &s->cand[j];
                int32_t dscr = s3o_psfwd_exit_score(s, bp, d->w_first_ci[cp->wid]);
                if (dscr != WORST)
                    dscr = add32(dscr, s3o_psfwd_tg_score(s, d->w_basewid[cp->wid], s->bp_realwid[bp], prev_real_wid(s, bp))
                                 >> S3O_PS_SENSCR_SHIFT);
                if (dscr > s->ltrans[cp->wid].dscr) { s->ltrans[cp->wid].dscr = dscr; s->ltrans[cp->wid].bp = bp; }
            }
        }
    }
    bestscore = s->last_phone_best_score;
    for (i = 0; i < s->n_cand; i++) {
        s3o_pscand_t *cp = &s->cand[i];
        cp->score = add32(cp->score, s->ltrans[cp->wid].dscr);
        cp->bp = s->ltrans[cp->wid].bp;
        if (cp->score > bestscore) bestscore = cp->score;
    }
    s->last_phone_best_score = bestscore;
    thresh = add32(bestscore, d->lponlybeam);
    for (i = 0; i < s->n_cand; i++) {
        s3o_pscand_t *cp = &s->cand[i];
        if (cp->score > thresh) {
            int32_t w = cp->wid, k = 0;
            alloc_all_rc(s, w);
            for (int32_t c = d->w_rc_off[w]; c < d->w_rc_off[w + 1]; c++) {
                s3o_pshmm_t *h = &s->hmm[s->rc_base + c];
                if (h->frame < frame_idx || cp->score > h->score[0]) { hmm_enter(h, cp->score, cp->bp, nf); k++; }
            }
            if (k > 0) { *(nawl++) = w; s->word_active[w] = 1; }
        }
    }
    s->n_awl[nf & 1] = (int32_t)(nawl - s->awl[nf & 1]);
}

/* set_real_wid, ngram_search.c:343-358 */
static void
set_real_wid(s3o_psfwd_t *s, int32_t bp)
{
    const s3o_psfwd_desc_t *d = &s->d;
    const int32_t prev = s->bp_bp[bp];
    if (d->w_flags[s->bp_wid[bp]] & 2) { if (prev != NO_BP) s->bp_realwid[bp] = s->bp_realwid[prev]; }
    else s->bp_realwid[bp] = d->w_basewid[s->bp_wid[bp]];
}

/* ngram_search_save_bp, ngram_search.c:360-441 */
static void
save_bp(s3o_psfwd_t *s, int32_t frame_idx, int32_t w, int32_t score, int32_t path, int32_t rc)
{
    const s3o_psfwd_desc_t *d = &s->d;
    int32_t bp = s->word_lat_idx[w];
    if (bp != NO_BP) {
        if (s->bp_score[bp] < score) {
            if (s->bp_bp[bp] != path) { s->bp_bp[bp] = path; set_real_wid(s, bp); }
            s->bp_score[bp] = score;
        }
        s->bss[s->bp_sidx[bp] + rc] = score;
        return;
    }
    if (s->bpidx >= s->bp_cap) {
        s->bp_cap *= 2;
        s->bp_frame = realloc(s->bp_frame, 4 * s->bp_cap); s->bp_wid = realloc(s->bp_wid, 4 * s->bp_cap);
        s->bp_bp = realloc(s->bp_bp, 4 * s->bp_cap); s->bp_score = realloc(s->bp_score, 4 * s->bp_cap);
        s->bp_sidx = realloc(s->bp_sidx, 4 * s->bp_cap); s->bp_realwid = realloc(s->bp_realwid, 4 * s->bp_cap);
        s->bp_valid = realloc(s->bp_valid, s->bp_cap);
    }
    if (s->bss_head >= s->bss_cap - d->n_ci) {
        s->bss_cap *= 2;
        s->bss = realloc(s->bss, 4 * (size_t)s->bss_cap);
    }
    {
        const int32_t rcsize = (d->w_flags[w] & 1) ? 1 : d->w_rc_off[w + 1] - d->w_rc_off[w];
        bp = s->bpidx;
        s->word_lat_idx[w] = bp;
        s->bp_wid[bp] = w; s->bp_frame[bp] = frame_idx; s->bp_bp[bp] = path; s->bp_score[bp] = score;
        s->bp_sidx[bp] = s->bss_head; s->bp_valid[bp] = 1;
        for (int32_t i = 0; i < rcsize; i++) s->bss[s->bss_head + i] = WORST;
        s->bss[s->bss_head + rc] = score;
        set_real_wid(s, bp);
        s->bpidx++;
        s->bss_head += rcsize;
    }
}

/* prune_word_chan :1037-1122 */
static void
prune_word_chan(s3o_psfwd_t *s, int32_t frame_idx)
{
    const s3o_psfwd_desc_t *d = &s->d;
    const int32_t nf = frame_idx + 1;
    const int32_t newword_thresh = add32(s->last_phone_best_score, d->wbeam);
    const int32_t lastphn_thresh = add32(s->last_phone_best_score, d->lponlybeam);
    int32_t *nawl = s->awl[nf & 1] + s->n_awl[nf & 1];
    for (int32_t i = 0; i < s->n_awl[frame_idx & 1]; i++) {
        int32_t w = s->awl[frame_idx & 1][i], k = 0;
        for (int32_t c = d->w_rc_off[w]; c < d->w_rc_off[w + 1]; c++) {
            s3o_pshmm_t *h = &s->hmm[s->rc_base + c];
            if (!h->alloc) continue;
            if (h->bestscore > lastphn_thresh) {
                h->frame = nf;
                k++;
                if (h->out_score > newword_thresh) save_bp(s, frame_idx, w, h->out_score, h->out_hist, c - d->w_rc_off[w]);
            }
            else if (h->frame != nf) { h->alloc = 0; hmm_clear(h, d->n_emit); }     /* listelem_free */
        }
        if (k > 0 && !s->word_active[w]) { *(nawl++) = w; s->word_active[w] = 1; }
    }
    s->n_awl[nf & 1] = (int32_t)(nawl - s->awl[nf & 1]);
    for (int32_t i = 0; i < d->n_1ph; i++) {
        s3o_pshmm_t *h = &s->hmm[s->sp_base + i];
        if (h->frame < frame_idx) continue;
        if (h->bestscore > lastphn_thresh) {
            h->frame = nf;
            if (h->out_score > newword_thresh) save_bp(s, frame_idx, d->sp_wid[i], h->out_score, h->out_hist, 0);
        }
    }
}

/* prune_channels :1125-1177 */
static void
prune_channels(s3o_psfwd_t *s, int32_t frame_idx)
{
    const s3o_psfwd_desc_t *d = &s->d;
    s->n_cand = 0;
    s->dynamic_beam = d->beam;
    if (d->maxhmmpf != -1 && s->st_n_root_chan_eval + s->st_n_nonroot_chan_eval > d->maxhmmpf) {
        int32_t bins[256], bw = -d->beam / 256, nhmms, i;
        memset(bins, 0, sizeof(bins));
        for (i = 0; i < d->n_root; i++) {
            int32_t b = sub32(s->best_score, s->hmm[i].bestscore) / bw;
            if (b >= 256) b = 255;
            ++bins[b];
        }
        for (i = 0; i < s->n_acl[frame_idx & 1]; i++) {
            int32_t b = sub32(s->best_score, s->hmm[s->acl[frame_idx & 1][i]].bestscore) / bw;
            if (b >= 256) b = 255;
            ++bins[b];
        }
        for (i = nhmms = 0; i < 256; ++i) {
            nhmms += bins[i];
            if (nhmms > d->maxhmmpf) break;
        }
        s->dynamic_beam = -(i * bw);
    }
    prune_root_chan(s, frame_idx);
    prune_nonroot_chan(s, frame_idx);
    last_phone_transition(s, frame_idx);
    prune_word_chan(s, frame_idx);
}

/* bptable_maxwpf :1183-1233 */
static void
bptable_maxwpf(s3o_psfwd_t *s, int32_t frame_idx)
{
    const s3o_psfwd_desc_t *d = &s->d;
    int32_t bp, n = 0, bestscr = (int32_t)0x80000000, best = -1;
    if (d->maxwpf == -1 || d->maxwpf == d->n_words) return;
    for (bp = s->bp_table_idx[1 + frame_idx]; bp < s->bpidx; bp++)
        if (d->w_flags[s->bp_wid[bp]] & 2) {
            if (s->bp_score[bp] > bestscr) { bestscr = s->bp_score[bp]; best = bp; }
            s->bp_valid[bp] = 0;
            n++;
        }
    if (best >= 0) { s->bp_valid[best] = 1; --n; }
    n = (s->bpidx - s->bp_table_idx[1 + frame_idx]) - n;
    for (; n > d->maxwpf; --n) {
        int32_t worstscr = 0x7fffffff, worst = -1;
        for (bp = s->bp_table_idx[1 + frame_idx]; bp < s->bpidx; bp++)
            if (s->bp_valid[bp] && s->bp_score[bp] < worstscr) { worstscr = s->bp_score[bp]; worst = bp; }
        if (worst < 0) abort();
        s->bp_valid[worst] = 0;
    }
}

/* word_transition :1236-1405 */
static void
word_transition(s3o_psfwd_t *s, int32_t frame_idx)
{
    const s3o_psfwd_desc_t *d = &s->d;
    const int32_t nf = frame_idx + 1;
    int32_t i, k = 0, bp, rc, thresh, newscore;
    for (i = d->n_ci - 1; i >= 0; --i) s->bestrc[i].score = WORST;
    for (bp = s->bp_table_idx[1 + frame_idx]; bp < s->bpidx; bp++) {
        const int32_t w = s->bp_wid[bp];
        const int32_t *rcss = &s->bss[s->bp_sidx[bp]];
        s->word_lat_idx[w] = NO_BP;
        if (w == d->finish_wid) continue;
        k++;
        for (rc = 0; rc < d->n_ci; rc++) {
            const int32_t v = d->w_last2_ci[w] == -1 ? rcss[0] : rcss[d->rc_cimap[d->w_rc_row[w] * d->n_ci + rc]];
            if (v > s->bestrc[rc].score) { s->bestrc[rc].score = v; s->bestrc[rc].path = bp; s->bestrc[rc].lc = d->w_last_ci[w]; }
        }
    }
    if (k == 0) return;
    thresh = add32(s->best_score, s->dynamic_beam);
    for (i = 0; i < d->n_root; i++) {
        s3o_pshmm_t *h = &s->hmm[i];
        const s3o_psbestrc_t *b = &s->bestrc[d->root_ci[i]];
        newscore = add32(add32(add32(b->score, d->nwpen), d->pip), s->pl[d->root_ci[i]]);
        if (newscore > thresh && (h->frame < frame_idx || newscore > h->score[0])) {
            hmm_enter(h, newscore, b->path, nf);
            h->senid[0] = d->root_lc_ssid[i * d->n_ci + b->lc];
        }
    }
    for (i = 0; i < d->n_1ph_lm; i++) s->ltrans[d->sp_wid[i]].dscr = (int32_t)0x80000000;
    for (bp = s->bp_table_idx[1 + frame_idx]; bp < s->bpidx; bp++) {
        if (!s->bp_valid[bp]) continue;
        for (i = 0; i < d->n_1ph_lm; i++) {
            const int32_t w = d->sp_wid[i];
            newscore = s3o_psfwd_exit_score(s, bp, d->w_first_ci[w]);
            if (newscore != WORST)
                newscore = add32(newscore, s3o_psfwd_tg_score(s, d->w_basewid[w], s->bp_realwid[bp], prev_real_wid(s, bp))
                                 >> S3O_PS_SENSCR_SHIFT);
            if (newscore > s->ltrans[w].dscr) { s->ltrans[w].dscr = newscore; s->ltrans[w].bp = bp; }
        }
    }
    for (i = 0; i < d->n_1ph_lm; i++) {
        const int32_t w = d->sp_wid[i];
        s3o_pshmm_t *h = &s->hmm[s->sp_base + i];
        if (w == d->start_wid) continue;
        newscore = add32(add32(s->ltrans[w].dscr, d->pip), s->pl[d->sp_ci[i]]);
        if (newscore > thresh && (h->frame < frame_idx || newscore > h->score[0])) {
            hmm_enter(h, newscore, s->ltrans[w].bp, nf);
            h->senid[0] = d->sp_lc_ssid[i * d->n_ci + d->w_last_ci[s->bp_wid[s->ltrans[w].bp]]];
        }
    }
    {
        const s3o_psbestrc_t *b = &s->bestrc[d->sil_ci];
        s3o_pshmm_t *h = &s->hmm[s->sp_base + s->w_sp[d->silence_wid]];
        newscore = add32(add32(add32(b->score, d->silpen), d->pip), s->pl[d->sp_ci[s->w_sp[d->silence_wid]]]);
        if (newscore > thresh && (h->frame < frame_idx || newscore > h->score[0])) hmm_enter(h, newscore, b->path, nf);
        for (i = 0; i < d->n_fill; i++) {
            h = &s->hmm[s->sp_base + d->fill_sp[i]];
            newscore = add32(add32(add32(b->score, d->fillpen), d->pip), s->pl[d->sp_ci[d->fill_sp[i]]]);
            if (newscore > thresh && (h->frame < frame_idx || newscore > h->score[0])) hmm_enter(h, newscore, b->path, nf);
        }
    }
}

/* deactivate_channels :1421-1443 */
static void
deactivate_channels(s3o_psfwd_t *s, int32_t frame_idx)
{
    const s3o_psfwd_desc_t *d = &s->d;
    for (int32_t i = 0; i < d->n_root; i++)
        if (s->hmm[i].frame == frame_idx) hmm_clear_scores(&s->hmm[i], d->n_emit);
    for (int32_t i = 0; i < d->n_1ph; i++)
        if (s->hmm[s->sp_base + i].frame == frame_idx) hmm_clear_scores(&s->hmm[s->sp_base + i], d->n_emit);
}

/* ngram_search_mark_bptable, ngram_search.c:322-340; bp_table_idx[-1] is valid in the reference: stored at +1 */
static void
mark_bptable(s3o_psfwd_t *s, int32_t frame_idx)
{
    if (frame_idx >= s->n_frame_alloc) {
        while (frame_idx >= s->n_frame_alloc) s->n_frame_alloc *= 2;
        s->bp_table_idx = realloc(s->bp_table_idx, 4 * (size_t)(s->n_frame_alloc + 2));
    }
    s->bp_table_idx[1 + frame_idx] = s->bpidx;
}

/* ngram_fwdtree_search :1446-1488 (the senone scores arrive from the caller's acmod_score) */
int32_t
s3o_psfwd_step(s3o_psfwd_t *s, const int16_t *senscr, int32_t frame_idx, int32_t n_senone_active)
{
    const s3o_psfwd_desc_t *d = &s->d;
    s->senscr = senscr;
    s->st_n_senone_active_utt += n_senone_active;
    mark_bptable(s, frame_idx);
    if (s->best_score == WORST || s->best_score < WORST) return 0;
    if (add32(s->best_score, 2 * d->beam) < WORST) renormalize(s, frame_idx, s->best_score);
    evaluate_channels(s, frame_idx);
    prune_channels(s, frame_idx);
    bptable_maxwpf(s, frame_idx);
    word_transition(s, frame_idx);
    deactivate_channels(s, frame_idx);
    ++s->n_frame;
    return 1;
}

/* ngram_fwdtree_finish :1490-1551 */
void
s3o_psfwd_finish(s3o_psfwd_t *s, int32_t cf)
{
    const s3o_psfwd_desc_t *d = &s->d;
    int32_t i;
    mark_bptable(s, cf);
    for (i = 0; i < d->n_root; i++) hmm_clear(&s->hmm[i], d->n_emit);
    for (i = 0; i < s->n_acl[cf & 1]; i++) hmm_clear(&s->hmm[s->acl[cf & 1][i]], d->n_emit);
    for (i = 0; i < s->n_awl[cf & 1]; i++) {
        int32_t w = s->awl[cf & 1][i];
        if (d->w_flags[w] & 1) continue;
        s->word_active[w] = 0;
        for (int32_t c = d->w_rc_off[w]; c < d->w_rc_off[w + 1]; c++) {        /* ngram_search_free_all_rc */
            s->hmm[s->rc_base + c].alloc = 0;
            hmm_clear(&s->hmm[s->rc_base + c], d->n_emit);
        }
    }
}

/* ngram_search_find_exit, ngram_search.c:444-484 */
int32_t
s3o_psfwd_find_exit(const s3o_psfwd_t *s, int32_t frame_idx, int32_t *out_best_score)
{
    int32_t end_bpidx, best_exit = NO_BP, best_score = WORST, bp;
    if (s->n_frame == 0) return NO_BP;
    if (frame_idx == -1 || frame_idx >= s->n_frame) frame_idx = s->n_frame - 1;
    end_bpidx = s->bp_table_idx[1 + frame_idx];
    while (frame_idx >= 0 && s->bp_table_idx[1 + frame_idx] == end_bpidx) --frame_idx;
    if (frame_idx < 0) return NO_BP;
    for (bp = s->bp_table_idx[1 + frame_idx]; bp < end_bpidx; ++bp) {
        if (s->bp_wid[bp] == s->d.finish_wid || s->bp_score[bp] > best_score) { best_score = s->bp_score[bp]; best_exit = bp; }
        if (s->bp_wid[bp] == s->d.finish_wid) break;
    }
    if (out_best_score) *out_best_score = best_score;
    return best_exit;
}

/* ngram_search_bp_iter :862-903 + ngram_search_bp2itor :777-818, lwf = 1 */
int32_t
s3o_psfwd_backtrace(const s3o_psfwd_t *s, int32_t bpidx, int32_t *wid, int32_t *sf, int32_t *ef, int32_t *ascr,
                    int32_t *lscr, int32_t *bps, int32_t max)
{
    const s3o_psfwd_desc_t *d = &s->d;
    int32_t n = 0, bp, cur;
    for (bp = bpidx; bp != NO_BP; bp = s->bp_bp[bp]) n++;
    if (n > max) return -n;
    for (bp = bpidx, cur = n - 1; bp != NO_BP; bp = s->bp_bp[bp], cur--) {
        const int32_t pbe = s->bp_bp[bp], w = s->bp_wid[bp];
        wid[cur] = w; ef[cur] = s->bp_frame[bp]; sf[cur] = pbe == NO_BP ? 0 : s->bp_frame[pbe] + 1;
        if (bps) bps[cur] = bp;
        if (pbe == NO_BP) { ascr[cur] = s->bp_score[bp]; lscr[cur] = 0; }
        else {
            const int32_t start_score = s3o_psfwd_exit_score(s, pbe, d->w_first_ci[w]);
            if (w == d->silence_wid) lscr[cur] = d->silpen;
            else if (d->w_flags[w] & 2) lscr[cur] = d->fillpen;
            else {
                lscr[cur] = s3o_psfwd_tg_score(s, s->bp_realwid[bp], s->bp_realwid[pbe], prev_real_wid(s, pbe)) >> S3O_PS_SENSCR_SHIFT;
                lscr[cur] = (int32_t)(lscr[cur] * 1.0f);
            }
            ascr[cur] = sub32(sub32(s->bp_score[bp], start_score), lscr[cur]);
        }
    }
    return n;
}

/* accessors for the ctypes harness (tests/psfwd_synth.py) */
void
s3o_psfwd_scalars(const s3o_psfwd_t *s, int32_t *out)
{
    out[0] = s->n_frame; out[1] = s->bpidx; out[2] = s->bss_head; out[3] = s->best_score; out[4] = s->last_phone_best_score;
    out[5] = s->renormalized; out[6] = s->st_n_root_chan_eval; out[7] = s->st_n_nonroot_chan_eval; out[8] = s->st_n_last_chan_eval;
    out[9] = s->st_n_word_lastchan_eval; out[10] = s->st_n_lastphn_cand_utt; out[11] = s->st_n_senone_active_utt;
}
const int32_t *
s3o_psfwd_array(const s3o_psfwd_t *s, int32_t which)
{
    switch (which) {
    case 0: return s->bp_frame; case 1: return s->bp_wid; case 2: return s->bp_bp; case 3: return s->bp_score;
    case 4: return s->bp_sidx; case 5: return s->bp_realwid; case 6: return s->bss; case 7: return s->bp_table_idx + 1;
    }
    return 0;
}
const uint8_t *s3o_psfwd_valid(const s3o_psfwd_t *s) { return s->bp_valid; }
