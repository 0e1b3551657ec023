/*
 * oracle/ps_search_oracle.c -- TEST INFRASTRUCTURE: the unmodified pocketsphinx decoder with the first pass
 * (ps_searchfuncs_t start / step / finish, pocketsphinx_internal.h:68-81) served by the CPU restatement
 * oracle/s3o_psfwd.c.  Same shape as integration/pocketsphinx/ps_search_amd.c (the product binding), which this
 * file pins: if the restatement is right, every output of oracle/_ref/ref_ps_ofwd equals ref_ps_fwd's.
 *
 * The search object stays the reference's ngram_search_t: finish copies the restatement's backpointer table
 * into it and then runs the reference's own finish, so that fwdflat, the lattice, bestpath, hyp and seg_iter
 * are the reference's code on the restatement's table.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sphinxbase/ckd_alloc.h>
#include <sphinxbase/err.h>
#include <sphinxbase/bitvec.h>
#include "pocketsphinx_internal.h"
#include "ngram_search.h"
#include "phone_loop_search.h"
#include "s3o_psfwd.h"

#define PSAMD_DESC_T s3o_psfwd_desc_t
#include "psamd_export.h"

typedef struct {
    ps_searchfuncs_t vt;            /* the decoder's table with three slots re-pointed */
    ps_searchfuncs_t *orig;
    s3o_psfwd_t *o;
    s3o_psfwd_desc_t desc;
    psamd_pool_t pool;
    uint8 *flags;
} oracle_search_t;

/* one binding per decoder: found through the vt pointer (vt is the struct's first member) */
#define BINDING(search) ((oracle_search_t *)(search)->vt)

static int
o_start(ps_search_t *search)
{
    ngram_search_t *ngs = (ngram_search_t *)search;
    oracle_search_t *b = BINDING(search);
    int32 i, k;
    ngs->done = FALSE;
    ngram_model_flush(ngs->lmset);
    ckd_free(search->hyp_str);
    search->hyp_str = NULL;
    /* the single-phone words' channels are shared with fwdflat, which leaves its multiplexed ids in them */
    for (i = 0; i < b->desc.n_1ph; i++) {
        root_chan_t *r = (root_chan_t *)ngs->word_chan[b->desc.sp_wid[i]];
        for (k = 0; k < b->desc.n_emit; k++) b->o->hmm[b->o->sp_base + i].senid[k] = r->hmm.senid[k];
    }
    s3o_psfwd_start(b->o);
    return 0;
}

static int
o_step(ps_search_t *search, int frame_idx)
{
    oracle_search_t *b = BINDING(search);
    acmod_t *acmod = ps_search_acmod(search);
    int16 const *senscr;
    if (!acmod->compallsen) {
        int32 s;
        s3o_psfwd_sen_active(b->o, frame_idx, b->flags);
        acmod_clear_active(acmod);
        for (s = 0; s < b->desc.n_sen; s++)
            if (b->flags[s]) acmod_activate_sen(acmod, s);
    }
    if ((senscr = acmod_score(acmod, &frame_idx)) == NULL) return 0;
    if (ps_search_lookahead(search)) {
        /* -pl_window: the decoder's own phone loop says what this frame's transitions add (phone_loop_search.h:103-105) */
        phone_loop_search_t *pls = (phone_loop_search_t *)ps_search_lookahead(search);
        int32 *pl = ckd_calloc(b->desc.n_ci, sizeof(int32)), ci;
        for (ci = 0; ci < b->desc.n_ci; ci++) pl[ci] = phone_loop_search_score(pls, ci);
        s3o_psfwd_set_lookahead(b->o, pl);
        ckd_free(pl);
    }
    {
        extern FILE *g_trace;
        s3o_psfwd_t *o = b->o;
        int rv = s3o_psfwd_step(o, senscr, frame_idx, acmod->n_senone_active), nf = frame_idx + 1, i;
        if (g_trace) {
            fprintf(g_trace, "F %d rv %d best %d lpbest %d dyn %d bpidx %d nacl %d nawl %d ncand %d\n", frame_idx, rv, o->best_score,
                    o->last_phone_best_score, o->dynamic_beam, o->bpidx, o->n_acl[nf & 1], o->n_awl[nf & 1], o->n_cand);
            fprintf(g_trace, "W");
            for (i = 0; i < o->n_awl[nf & 1]; i++) fprintf(g_trace, " %d", o->awl[nf & 1][i]);
            fprintf(g_trace, "\nC");
            for (i = 0; i < o->n_cand; i++) fprintf(g_trace, " %d:%d:%d", o->cand[i].wid, o->cand[i].score, o->cand[i].bp);
            fprintf(g_trace, "\n");
        }
        return rv;
    }
}

static int
o_finish(ps_search_t *search)
{
    ngram_search_t *ngs = (ngram_search_t *)search;
    oracle_search_t *b = BINDING(search);
    s3o_psfwd_t *o = b->o;
    int32 cf = ps_search_acmod(search)->output_frame, i, k;
    s3o_psfwd_finish(o, cf);
    /* the table into the reference's structures */
    while (ngs->bp_table_size <= o->bpidx) {
        ngs->bp_table_size *= 2;
        ngs->bp_table = ckd_realloc(ngs->bp_table, ngs->bp_table_size * sizeof(*ngs->bp_table));
    }
    while (ngs->bscore_stack_size <= o->bss_head + b->desc.n_ci) {
        ngs->bscore_stack_size *= 2;
        ngs->bscore_stack = ckd_realloc(ngs->bscore_stack, ngs->bscore_stack_size * sizeof(*ngs->bscore_stack));
    }
    for (i = 0; i <= cf; i++) ngram_search_mark_bptable(ngs, i);       /* grows bp_table_idx as the reference does */
    for (i = 0; i < o->bpidx; i++) {
        bptbl_t *be = &ngs->bp_table[i];
        const int32 w = o->bp_wid[i];
        be->frame = o->bp_frame[i]; be->valid = o->bp_valid[i]; be->refcnt = 0; be->wid = w; be->bp = o->bp_bp[i];
        be->score = o->bp_score[i]; be->s_idx = o->bp_sidx[i]; be->real_wid = o->bp_realwid[i];
        be->last_phone = b->desc.w_last_ci[w]; be->last2_phone = b->desc.w_last2_ci[w];
    }
    memcpy(ngs->bscore_stack, o->bss, sizeof(int32) * o->bss_head);
    for (i = 0; i <= cf; i++) ngs->bp_table_idx[i] = o->bp_table_idx[1 + i];
    ngs->bpidx = o->bpidx; ngs->bss_head = o->bss_head; ngs->n_frame = o->n_frame;
    ngs->best_score = o->best_score; ngs->last_phone_best_score = o->last_phone_best_score;
    ngs->renormalized = o->renormalized;
    ngs->st.n_root_chan_eval = o->st_n_root_chan_eval; ngs->st.n_nonroot_chan_eval = o->st_n_nonroot_chan_eval;
    ngs->st.n_last_chan_eval = o->st_n_last_chan_eval; ngs->st.n_word_lastchan_eval = o->st_n_word_lastchan_eval;
    ngs->st.n_lastphn_cand_utt = o->st_n_lastphn_cand_utt; ngs->st.n_senone_active_utt = o->st_n_senone_active_utt;
    ngs->n_active_chan[0] = ngs->n_active_chan[1] = 0;
    ngs->n_active_word[0] = ngs->n_active_word[1] = 0;
    for (i = 0; i < b->desc.n_1ph; i++) {
        root_chan_t *r = (root_chan_t *)ngs->word_chan[b->desc.sp_wid[i]];
        for (k = 0; k < b->desc.n_emit; k++) r->hmm.senid[k] = o->hmm[o->sp_base + i].senid[k];
    }
    /* ngram_search_finish: ngram_fwdtree_finish (a no-op on the host's empty lists), fwdflat if enabled, done */
    return b->orig->finish(search);
}

static void
o_free(ps_search_t *search)
{
    oracle_search_t *b = BINDING(search);
    ps_searchfuncs_t *orig = b->orig;
    search->vt = orig;
    s3o_psfwd_free(b->o);
    psamd_pool_free(&b->pool);
    ckd_free(b->flags);
    ckd_free(b);
    orig->free(search);
}

int
ps_oracle_search_install(ps_decoder_t *ps)
{
    ngram_search_t *ngs = (ngram_search_t *)ps->search;
    oracle_search_t *b;
    if (ps->search == NULL || strcmp(ps_search_name(ps->search), "ngram") != 0) {
        E_ERROR("ps_oracle_search_install: the decoder's search is not the N-gram search\n");
        return -1;
    }
    b = ckd_calloc(1, sizeof(*b));
    if (psamd_export(ps, ngs, &b->desc, &b->pool) < 0) { ckd_free(b); return -1; }
    b->o = s3o_psfwd_init(&b->desc);
    b->flags = ckd_calloc(b->desc.n_sen, 1);
    if (getenv("PSO_LMCHECK")) {
        /* pin the flat trigram on the reference's own ngram_tg_score (tests/test_oracle_psfwd.py): every
         * PSO_LMCHECK-th bigram and trigram of the model (guaranteed hits, through the words' dictionary ids), the
         * same n-grams with the last word replaced (back-off paths), every word after <s> and with no history */
        const s3o_psfwd_desc_t *d = &b->desc;
        int32 step = atoi(getenv("PSO_LMCHECK")), nbad = 0, n = 0, n_used, nw = d->n_words, u, bi, t, w;
        int32 *rev = ckd_calloc(d->lm_n_ug, sizeof(int32));
        if (step < 1) step = 1;
        for (u = 0; u < d->lm_n_ug; u++) rev[u] = -1;
        for (w = nw - 1; w >= 0; w--)
            if (d->w_lmwid[w] >= 0 && d->w_basewid[w] == w) rev[d->w_lmwid[w]] = w;
#define CHECK3(w3, w2, w1) do { int32 x, y; if ((w3) >= 0 && ngram_model_set_known_wid(ngs->lmset, (w3))) { \
            x = ngram_tg_score(ngs->lmset, (w3), (w2), (w1), &n_used); y = s3o_psfwd_tg_score(b->o, (w3), (w2), (w1)); n++; \
            if (x != y && nbad++ < 10) E_ERROR("LM check: tg_score(%d | %d %d) reference %d restatement %d\n", (w3), (w2), (w1), x, y); } } while (0)
        for (w = 0; w < nw; w++) {
            CHECK3(d->w_basewid[w], dict_startwid(ps->dict), -1);
            CHECK3(d->w_basewid[w], -1, -1);
            CHECK3(d->w_basewid[w], d->w_basewid[(w * 7 + 3) % nw], dict_startwid(ps->dict));
        }
        for (u = 0; u < d->lm_n_ug; u++)
            for (bi = d->ug_firstbg[u]; bi < d->ug_firstbg[u + 1]; bi++) {
                if (rev[u] < 0 || rev[d->bg_wid[bi]] < 0) continue;
                if (bi % step == 0) {
                    CHECK3(rev[d->bg_wid[bi]], rev[u], -1);
                    CHECK3(rev[d->bg_wid[bi]], rev[u], rev[(bi * 31) % d->lm_n_ug] < 0 ? -1 : rev[(bi * 31) % d->lm_n_ug]);
                    CHECK3(rev[(bi * 17) % d->lm_n_ug], rev[d->bg_wid[bi]], rev[u]);
                }
                for (t = d->bg_firsttg[bi]; t < d->bg_firsttg[bi + 1]; t++)
                    if (t % step == 0 && rev[d->tg_wid[t]] >= 0) CHECK3(rev[d->tg_wid[t]], rev[d->bg_wid[bi]], rev[u]);
            }
#undef CHECK3
        ckd_free(rev);
        ngram_model_flush(ngs->lmset);
        if (nbad) E_FATAL("LM check: %d differences in %d scores\n", nbad, n);
        E_INFO("LM check: %d scores identical\n", n);
    }
    b->orig = ps->search->vt;
    b->vt = *b->orig;
    b->vt.start = o_start; b->vt.step = o_step; b->vt.finish = o_finish; b->vt.free = o_free;
    ps->search->vt = &b->vt;
    E_INFO("first pass served by oracle/s3o_psfwd.c: %d roots, %d interior channels, %d single-phone words, %d right-context channels\n",
           b->desc.n_root, b->desc.n_nonroot, b->desc.n_1ph, b->desc.w_rc_off[b->desc.n_words]);
    return 0;
}
