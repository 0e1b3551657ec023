/*
 * oracle/s3o_wordlevel.c -- CPU ORACLE (test infrastructure only) for the WORD LEVEL of
 * sphinx3's mode-4 search: the trigram look-up, the Viterbi history
 * table and the word transitions that close a search frame.
 *
 * Restates, on plain arrays, the behaviour of
 *   sphinx3/src/libs3decoder/liblm/lm.c            lm_ug_score :983-995, lm_bg_score :1241-1312,
 *                                                  lm_tg_score :1661-1833 (find_bg/find_tg :1132-1178)
 *   sphinx3/src/libs3decoder/libsearch/vithist.c   vithist_utt_begin :300-335, vithist_enter :396-489,
 *                                                  vithist_rescore :492-574, vithist_frame_gc :580-645,
 *                                                  vithist_prune :649-718, vithist_frame_windup :748-763,
 *                                                  vithist_utt_end :766-860, vithist_backtrace :1066-1100
 *   sphinxbase/src/libsphinxbase/util/heap.c       subheap_insert :113-148, subheap_pop :159-200
 *                                                  (vithist_prune pops its entries from this heap; its
 *                                                  order among EQUAL scores is part of the behaviour)
 *   sphinx3/src/libs3decoder/libsearch/srch_time_switch_tree.c   srch_utt_word_trans :1086-1179
 *
 * The reference's caches (lm->tgcache, tginfo lists, membg) only memoise: lm_tg_score is a
 * pure function of (lw1, lw2, lw3), which is what is restated here.  LM word ids are int32,
 * "no word" (BAD_LMWID) is any negative value.
 *
 * Parity status: PINNED.  oracle/_ref/ref_s3owl_decode (integration/sphinx3/s3amd_tst.c built
 * with -DLT_ORACLE -DWL_ORACLE) runs the unmodified reference decoder with its whole word
 * level served from this file; tests/test_oracle_wordlevel.py checks that its -hyp / -hypseg
 * are byte-identical to the unmodified reference on tidigits and RM1, and the committed
 * per-frame traces (tests/golden/wordlevel_*.npz) replay bit-for-bit.
 */
#include <stdlib.h>
#include <string.h>
#include "s3o.h"

/* ------------------------------------------------------------------ */
/* the trigram                                                         */
/* ------------------------------------------------------------------ */
/* find_bg / find_tg (lm.c:1132-1178, 1529-1570): the index of w in a sorted run of n ids, or -1.
 * (The reference narrows by bisection to <= 16 entries and then scans; on sorted unique ids
 * that is the unique match, which plain bisection finds as well.) */
static int32_t
find_sorted(const int32_t *v, int32_t n, int32_t w)
{
    int32_t b = 0, e = n;
    while (e - b > 16) {
        const int32_t i = (b + e) >> 1;
        if (v[i] < w) b = i + 1;
        else if (v[i] > w) e = i;
        else return i;
    }
    for (; b < e; b++)
        if (v[b] == w) return b;
    return -1;
}

/* lm_ug_score, lm.c:983-995 */
static int32_t
ug_score(const s3o_lm3g_t *lm, int32_t lw, int32_t wid)
{
    int32_t s = lm->ug_prob[lw];
    if (lm->inclass) s += lm->inclass[wid];
    return s;
}

/* lm_bg_score, lm.c:1241-1312 */
int32_t
s3o_lm_bg_score(const s3o_lm3g_t *lm, int32_t lw1, int32_t lw2, int32_t wid)
{
    int32_t n, i, score;
    if (lm->n_bg == 0 || lw1 < 0)
        return ug_score(lm, lw2, wid);
    n = lm->ug_firstbg[lw1 + 1] - lm->ug_firstbg[lw1];
    i = n > 0 ? find_sorted(lm->bg_wid + lm->ug_firstbg[lw1], n, lw2) : -1;
    if (i >= 0)
        score = lm->bg_prob[lm->ug_firstbg[lw1] + i];
    else
        score = lm->ug_bowt[lw1] + lm->ug_prob[lw2];
    if (lm->inclass) score += lm->inclass[wid];
    return score;
}

/* lm_tg_score, lm.c:1661-1833 (load_tg :1363-1525 gives bowt and the trigram run of the bigram
 * (lw1, lw2); a missing bigram has bowt 0 and no trigrams) */
int32_t
s3o_lm_tg_score(const s3o_lm3g_t *lm, int32_t lw1, int32_t lw2, int32_t lw3, int32_t wid)
{
    int32_t nb, b, bowt = 0, t0 = 0, nt = 0, i;
    if (lm->n_tg == 0 || lw1 < 0)
        return s3o_lm_bg_score(lm, lw2, lw3, wid);
    nb = lm->ug_firstbg[lw1 + 1] - lm->ug_firstbg[lw1];
    b = nb > 0 ? find_sorted(lm->bg_wid + lm->ug_firstbg[lw1], nb, lw2) : -1;
    if (b >= 0) {
        b += lm->ug_firstbg[lw1];
        bowt = lm->bg_bowt[b];
        t0 = lm->bg_firsttg[b];
        nt = lm->bg_firsttg[b + 1] - t0;
    }
    i = nt > 0 ? find_sorted(lm->tg_wid + t0, nt, lw3) : -1;
    if (i >= 0) {
        int32_t s = lm->tg_prob[t0 + i];
        if (lm->inclass) s += lm->inclass[wid];
        return s;
    }
    return bowt + s3o_lm_bg_score(lm, lw2, lw3, wid);
}

/* ------------------------------------------------------------------ */
/* the Viterbi history                                                 */
/* ------------------------------------------------------------------ */
s3o_vithist_t *
s3o_vithist_init(int32_t cap, int32_t max_frames, int32_t wbeam, int32_t bghist)
{
    s3o_vithist_t *vh = calloc(1, sizeof(*vh));
    vh->cap = cap; vh->max_frames = max_frames; vh->wbeam = wbeam; vh->bghist = bghist;
    vh->score = calloc(cap, 4); vh->pred = calloc(cap, 4); vh->lw0 = calloc(cap, 4); vh->lw1 = calloc(cap, 4);
    vh->wid = calloc(cap, 4); vh->sf = calloc(cap, 4); vh->ef = calloc(cap, 4); vh->ascr = calloc(cap, 4);
    vh->lscr = calloc(cap, 4); vh->type = calloc(cap, 4); vh->valid = calloc(cap, 1);
    vh->frame_start = calloc(max_frames + 2, 4); vh->bestscore = calloc(max_frames + 2, 4);
    vh->bestvh = calloc(max_frames + 2, 4);
    return vh;
}

void
s3o_vithist_free(s3o_vithist_t *vh)
{
    if (!vh) return;
    free(vh->score); free(vh->pred); free(vh->lw0); free(vh->lw1); free(vh->wid); free(vh->sf); free(vh->ef);
    free(vh->ascr); free(vh->lscr); free(vh->type); free(vh->valid); free(vh->frame_start);
    free(vh->bestscore); free(vh->bestvh);
    free(vh);
}

/* vithist_utt_begin, vithist.c:300-335: entry 0 is the dummy <s> */
void
s3o_vithist_utt_begin(s3o_vithist_t *vh, int32_t startwid, int32_t start_lwid)
{
    vh->n_entry = 1;
    vh->wid[0] = startwid; vh->sf[0] = -1; vh->ef[0] = -1; vh->ascr[0] = 0; vh->lscr[0] = 0;
    vh->score[0] = 0; vh->pred[0] = -1; vh->type[0] = 0; vh->valid[0] = 1;
    vh->lw0[0] = start_lwid; vh->lw1[0] = -1;
    vh->n_frm = 0;
    vh->frame_start[0] = 1;
    vh->bestscore[0] = S3O_MAX_NEG_INT32;
    vh->bestvh[0] = -1;
    vh->overflow = 0;
}

typedef struct { int32_t wid, sf, ef, ascr, lscr, score, pred, type, lw0, lw1; } tve_t;

/* vithist_enter with comp_rc == -1 (composite triphones), vithist.c:396-489.  The reference
 * finds the frame's entry for an LM state through the per-frame lms2vh tree; a scan of the
 * frame's entries finds the same entry (there is at most one per LM state). */
static void
vithist_enter(s3o_vithist_t *vh, const tve_t *t)
{
    const int32_t fs = vh->frame_start[vh->n_frm];
    int32_t id;
    for (id = fs; id < vh->n_entry; id++)
        if (vh->lw0[id] == t->lw0 && vh->lw1[id] == t->lw1) break;
    if (id < vh->n_entry && !(vh->score[id] < t->score)) {
        /* known LM state, not better: only the frame's best is looked at below */
    }
    else {
        if (id == vh->n_entry) {
            if (vh->n_entry >= vh->cap) { vh->overflow = 1; return; }
            vh->n_entry++;
        }
        vh->wid[id] = t->wid; vh->sf[id] = t->sf; vh->ef[id] = t->ef; vh->ascr[id] = t->ascr;
        vh->lscr[id] = t->lscr; vh->score[id] = t->score; vh->pred[id] = t->pred; vh->type[id] = t->type;
        vh->lw0[id] = t->lw0; vh->lw1[id] = t->lw1; vh->valid[id] = 1;
    }
    if (vh->bestscore[vh->n_frm] < t->score) {
        vh->bestscore[vh->n_frm] = t->score;
        vh->bestvh[vh->n_frm] = id;
    }
}

/* vithist_rescore, vithist.c:492-574.  Returns -1 for pred == -1 (E_FATAL in the reference). */
int32_t
s3o_vithist_rescore(s3o_vithist_t *vh, const s3o_lm3g_t *lm, const s3o_wdict_t *d, int32_t wid,
                    int32_t ef, int32_t score, int32_t pred, int32_t type)
{
    tve_t t;
    int32_t se, fe, i, lwid;
    if (pred < 0) return -1;
    t.wid = wid; t.sf = vh->ef[pred] + 1; t.ef = ef; t.type = type;
    t.ascr = (int32_t)((uint32_t)score - (uint32_t)vh->score[pred]);
    t.lscr = 0;
    if (pred == 0) { se = 0; fe = 1; }
    else { se = vh->frame_start[vh->ef[pred]]; fe = vh->frame_start[vh->ef[pred] + 1]; }
    if (d->is_filler[wid]) {
        t.lscr = d->fillpen[wid];
        t.score = (int32_t)((uint32_t)score + (uint32_t)t.lscr);
        t.pred = pred;
        t.lw0 = vh->lw0[pred]; t.lw1 = vh->lw1[pred];
        vithist_enter(vh, &t);
        return 0;
    }
    lwid = d->lwid[wid];
    t.lw0 = lwid;
    for (i = se; i < fe; i++) {
        if (!vh->valid[i]) continue;
        t.score = (int32_t)((uint32_t)vh->score[i] + (uint32_t)t.ascr);
        t.lscr = s3o_lm_tg_score(lm, vh->lw1[i], vh->lw0[i], lwid, wid);
        t.score = (int32_t)((uint32_t)t.score + (uint32_t)t.lscr);
        if ((int32_t)((uint32_t)t.score - (uint32_t)vh->wbeam) >= vh->bestscore[vh->n_frm]) {
            t.pred = i;
            t.lw1 = vh->lw0[i];
            vithist_enter(vh, &t);
        }
    }
    return 0;
}

/* sphinxbase heap.c: a pointer-linked binary heap kept balanced by subtree sizes.  Restated on
 * index arrays; val = -score, data = entry id. */
typedef struct { int32_t *val, *data, *nl, *nr, *l, *r; int32_t n_alloc, top; } heap_t;

static int32_t
heap_ins(heap_t *h, int32_t root, int32_t data, int32_t val)          /* subheap_insert, heap.c:113-148 */
{
    if (root < 0) {
        const int32_t k = h->n_alloc++;
        h->data[k] = data; h->val[k] = val; h->l[k] = h->r[k] = -1; h->nl[k] = h->nr[k] = 0;
        return k;
    }
    if (h->val[root] > val) {
        const int32_t td = h->data[root], tv = h->val[root];
        h->data[root] = data; h->val[root] = val;
        data = td; val = tv;
    }
    if (h->nl[root] > h->nr[root]) { h->r[root] = heap_ins(h, h->r[root], data, val); h->nr[root]++; }
    else { h->l[root] = heap_ins(h, h->l[root], data, val); h->nl[root]++; }
    return root;
}

static int32_t
heap_pop_root(heap_t *h, int32_t root)                              /* subheap_pop, heap.c:159-200 */
{
    const int32_t l = h->l[root], r = h->r[root];
    if (l < 0) {
        if (r < 0) return -1;
        h->data[root] = h->data[r]; h->val[root] = h->val[r];
        h->r[root] = heap_pop_root(h, r); h->nr[root]--;
    }
    else if (r < 0 || h->val[l] < h->val[r]) {
        h->data[root] = h->data[l]; h->val[root] = h->val[l];
        h->l[root] = heap_pop_root(h, l); h->nl[root]--;
    }
    else {
        h->data[root] = h->data[r]; h->val[root] = h->val[r];
        h->r[root] = heap_pop_root(h, r); h->nr[root]--;
    }
    return root;
}

/* vithist_prune + vithist_frame_gc, vithist.c:580-718.  `order_out` (optional, >= the frame's
 * entry count) receives the ids in the order the heap popped them (for the tests). */
void
s3o_vithist_prune(s3o_vithist_t *vh, const s3o_wdict_t *d, int32_t frm, int32_t maxwpf, int32_t maxhist,
                  int32_t beam, int32_t *order_out)
{
    const int32_t se = vh->frame_start[frm], fe = vh->n_entry - 1, n = fe - se + 1;
    const int32_t th = (int32_t)((uint32_t)vh->bestscore[frm] + (uint32_t)beam);
    int32_t i, filler_done = 0, nw = 0, te, bs, bv, npop = 0;
    int32_t *wid = calloc(maxwpf > 0 ? maxwpf + 1 : 1, 4);
    heap_t h;
    h.val = malloc(4 * (n + 1)); h.data = malloc(4 * (n + 1)); h.nl = malloc(4 * (n + 1));
    h.nr = malloc(4 * (n + 1)); h.l = malloc(4 * (n + 1)); h.r = malloc(4 * (n + 1));
    h.n_alloc = 0; h.top = -1;
    for (i = se; i <= fe; i++) {
        h.top = heap_ins(&h, h.top, i, (int32_t)(0u - (uint32_t)vh->score[i]));
        vh->valid[i] = 0;
    }
    while (h.top >= 0) {
        const int32_t id = h.data[h.top];
        int32_t k;
        h.top = heap_pop_root(&h, h.top);
        if (order_out) order_out[npop++] = id;
        if (!(vh->score[id] >= th && maxhist > 0)) break;
        if (d->is_filler[vh->wid[id]]) {
            if (filler_done) continue;
            filler_done = 1;
        }
        for (k = 0; k < nw && wid[k] != vh->wid[id]; k++);
        if (k == nw) {
            if (maxwpf > 0) {
                wid[nw++] = vh->wid[id];
                --maxwpf; --maxhist;
                vh->valid[id] = 1;
            }
        }
        else if (!vh->bghist) {
            --maxhist;
            vh->valid[id] = 1;
        }
    }
    if (order_out) for (; npop < n; npop++) order_out[npop] = -1;
    free(wid); free(h.val); free(h.data); free(h.nl); free(h.nr); free(h.l); free(h.r);
    /* vithist_frame_gc */
    te = se; bs = S3O_MAX_NEG_INT32; bv = -1;
    for (i = se; i <= fe; i++) {
        if (!vh->valid[i]) continue;
        if (i != te) {
            vh->wid[te] = vh->wid[i]; vh->sf[te] = vh->sf[i]; vh->ef[te] = vh->ef[i]; vh->ascr[te] = vh->ascr[i];
            vh->lscr[te] = vh->lscr[i]; vh->score[te] = vh->score[i]; vh->pred[te] = vh->pred[i];
            vh->type[te] = vh->type[i]; vh->lw0[te] = vh->lw0[i]; vh->lw1[te] = vh->lw1[i]; vh->valid[te] = 1;
        }
        if (vh->score[i] > bs) { bs = vh->score[i]; bv = te; }
        te++;
    }
    vh->bestvh[frm] = bv;
    vh->n_entry = te;
}

/* vithist_frame_windup, vithist.c:748-763 */
void
s3o_vithist_frame_windup(s3o_vithist_t *vh, int32_t frm)
{
    (void)frm;
    vh->n_frm++;
    vh->frame_start[vh->n_frm] = vh->n_entry;
    vh->bestscore[vh->n_frm] = S3O_MAX_NEG_INT32;
    vh->bestvh[vh->n_frm] = -1;
}

/* srch_utt_word_trans, srch_time_switch_tree.c:1086-1179: the best exit per word-final CI
 * phone re-enters the unigram lextree of this transition, the frame's best exit the filler
 * lextree.  Returns the number of unigram-tree calls (lc / scr / hist), -1 when the frame has
 * no entry (then nothing is entered and the transition counter does not advance);
 * *fill_scr / *fill_hist = the filler tree's call. */
int32_t
s3o_word_trans(const s3o_vithist_t *vh, const s3o_wdict_t *d, int32_t cf, int32_t wordend_beam,
               int32_t *lc, int32_t *scr, int32_t *hist, int32_t *fill_scr, int32_t *fill_hist)
{
    int32_t p, id, n = 0, maxp = S3O_MAX_NEG_INT32;
    int32_t *bs, *bv;
    if (vh->bestvh[cf] < 0) return -1;
    bs = malloc(4 * d->n_ci); bv = malloc(4 * d->n_ci);
    for (p = 0; p < d->n_ci; p++) { bs[p] = S3O_MAX_NEG_INT32; bv[p] = -1; }
    for (id = vh->frame_start[cf]; id < vh->n_entry; id++) {
        if (!vh->valid[id]) continue;
        p = d->last_ci[vh->wid[id]];            /* filler phones already mapped to silence */
        if (vh->score[id] > bs[p]) {
            bs[p] = vh->score[id]; bv[p] = id;
            if (maxp < vh->score[id]) maxp = vh->score[id];
        }
    }
    for (p = 0; p < d->n_ci; p++)
        if (bv[p] >= 0 && (wordend_beam == 0 || bs[p] > (int32_t)((uint32_t)wordend_beam + (uint32_t)maxp))) {
            lc[n] = p; scr[n] = bs[p]; hist[n] = bv[p]; n++;
        }
    *fill_scr = vh->bestscore[cf]; *fill_hist = vh->bestvh[cf];
    free(bs); free(bv);
    return n;
}

/* vithist_utt_end, vithist.c:766-860: the final </s> entry; returns its id or -1.  (When the
 * last frame has no exit the reference adds a silence entry spanning the rest and retries.) */
int32_t
s3o_vithist_utt_end(s3o_vithist_t *vh, const s3o_lm3g_t *lm, const s3o_wdict_t *d)
{
    int32_t f, i, sv = 0, nsv = 0, best = S3O_MAX_NEG_INT32, bestvh = -1, id;
    for (f = vh->n_frm - 1; f >= 0; --f) {
        sv = vh->frame_start[f]; nsv = vh->frame_start[f + 1];
        if (sv < nsv) break;
    }
    if (f < 0) return -1;
    for (i = sv; i < nsv; i++) {
        const int32_t s = (int32_t)((uint32_t)vh->score[i]
                                    + (uint32_t)s3o_lm_tg_score(lm, vh->lw1[i], vh->lw0[i], d->finish_lwid, d->finishwid));
        if (best < s) { best = s; bestvh = i; }
    }
    if (f != vh->n_frm - 1) {
        vh->n_frm -= 1;
        s3o_vithist_rescore(vh, lm, d, d->silwid, vh->n_frm, vh->score[bestvh], bestvh, -1);
        vh->n_frm += 1;
        vh->frame_start[vh->n_frm] = vh->n_entry;
        return s3o_vithist_utt_end(vh, lm, d);
    }
    if (vh->n_entry >= vh->cap) { vh->overflow = 1; return -1; }
    id = vh->n_entry++;
    vh->wid[id] = d->finishwid;
    vh->sf[id] = vh->ef[bestvh] + 1;        /* (BAD_S3FRMID never occurs: entry 0 has ef -1 -> sf 0) */
    vh->ef[id] = vh->n_frm;
    vh->ascr[id] = 0;
    vh->lscr[id] = (int32_t)((uint32_t)best - (uint32_t)vh->score[bestvh]);
    vh->score[id] = best; vh->pred[id] = bestvh; vh->type[id] = 0; vh->valid[id] = 1;
    vh->lw0[id] = d->finish_lwid; vh->lw1[id] = d->finish_lwid;
    return id;
}

/* vithist_backtrace, vithist.c:1066-1100: ids from the first word to `id` (entry 0 excluded) */
int32_t
s3o_vithist_backtrace(const s3o_vithist_t *vh, int32_t id, int32_t *ids, int32_t max_ids)
{
    int32_t n = 0, i, k;
    for (i = id; i > 0; i = vh->pred[i]) n++;
    if (n > max_ids) return -1;
    for (i = id, k = n - 1; i > 0; i = vh->pred[i], k--) ids[k] = i;
    return n;
}
