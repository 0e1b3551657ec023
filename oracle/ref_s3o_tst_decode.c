/*
 * oracle/ref_s3o_tst_decode.c -- TEST INFRASTRUCTURE: sphinx3_decode (mode 4, "fwdtree") with the per-frame lextree
 * operations -- and, with -DWL_ORACLE, the word level -- served by the CPU restatement (oracle/s3o_lextree.c,
 * oracle/s3o_wordlevel.c) through the reference's own srch_funcs_t table (sphinx3/include/srch.h:528-701).  Same slots
 * and the same flattening (integration/sphinx3/s3amd_flatten.h) as the product binding integration/sphinx3/s3amd_tst.c,
 * without any of the product: no GPU, no libcmusphinx_amd.
 *   (default)            oracle/_ref/ref_s3olt_decode : identical -hyp / -hypseg to the unmodified reference PINS the
 *                        lextree restatement (tests/test_oracle_lextree.py)
 *   -DWL_ORACLE          oracle/_ref/ref_s3owl_decode : pins the word level the same way
 *                        (tests/test_oracle_wordlevel.py) and records the per-frame word-level trace the device tests replay.
 *
 * What stays the reference's: kb_init (models, dictionary, LM, lextree_build, dict2pid), feature
 * computation, the utterance API, hypothesis output (and, frame-synchronous mode, vithist_* / lm_*).
 * The lextrees the reference built are FLATTENED once (flatten_tree) into the node/CSR arrays the
 * backends take, its lm_t / dict_t into the plain arrays of s3a_lm3g_init / s3a_wordlevel_init
 * (flatten_lm).  The replaced slots restate the control flow of srch_time_switch_tree.c:457-560
 * (begin/end), :776-907 (hmm_compute_lv2), :923-1007 (propagate_graph_ph_lv2), :1010-1210 (rescoring,
 * word transitions), :1213-1237 (frame_windup), :1262-1324 (select_active_gmm).
 */
#define main sphinx3_decode_reference_main
#include "main_decode.c"        /* the reference's file, in place (for its arg table) */
#undef main

/* srch_TST_graph_t (the mode's private graph structure) is defined INSIDE the reference's
 * srch_time_switch_tree.c:233-257, not in a header.  A maintainer would add these slots in
 * that file; here the file is #included in place (not copied) to obtain the definition. */
#include "srch_time_switch_tree.c"

#include <string.h>
#include "byteorder.h"
#include "s3_decode.h"
#include "srch.h"
#include "gmm_wrap.h"
#include "dict2pid.h"
#include "lextree.h"
#include "vithist.h"
#include "dag.h"

#include "s3o.h"

#include "s3amd_flatten.h"

/* ------------------------------------------------------------------ */
/* backend                                                             */
/* ------------------------------------------------------------------ */
static __thread int32 g_ntree;           /* 2 * n_lextree: unigram trees then filler trees */
static __thread flat_t **g_flat;
static __thread int32 *g_tp_flat;        /* tmat->tp flattened */
static __thread int16 *g_sseq_flat, *g_comsseq_flat, *g_comstate;
static __thread int32 *g_comstate_off, g_n_comstate;
static __thread int32 g_max_node;
static __thread int32 *g_best, *g_wbest, *g_nact;
static __thread int32 *g_exit_n, *g_exit_wid, *g_exit_scr, *g_exit_hist;
static __thread long g_frames, g_histframes;
#include <time.h>
static __thread double g_t_score, g_t_search, g_t_word, g_t_utt;
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

static __thread s3o_lextree_t **g_lt;

/* Optional trace of the FIRST utterance (env S3O_TRACE=file): the flattened trees and, per
 * frame, every input the lextree operations consumed and every result they produced.
 * tests/golden/make_golden.py turns it into the fixture the oracle-vs-HIP lextree parity
 * tests replay.  Record = {tag, n, n x int32}. */
static __thread FILE *g_trace;
static __thread int g_trace_utt;
#ifdef WL_ORACLE
/* the word level from oracle/s3o_wordlevel.c, and (env S3O_WLTRACE=file) a record of what it consumed and
 * produced in every frame of the FIRST utterance: tests/golden/make_golden.py turns it into the fixture the
 * device word level replays.  Record = {tag, n, n x int32}. */
static __thread wl_flat_t *g_wl;
static __thread s3o_lm3g_t g_olm;
static __thread s3o_wdict_t g_od;
static __thread s3o_vithist_t *g_ovh;
static __thread FILE *g_wltrace;
static void
wtr(int32 tag, int32 n, const void *data)
{
    if (!g_wltrace) return;
    fwrite(&tag, 4, 1, g_wltrace); fwrite(&n, 4, 1, g_wltrace);
    if (n) fwrite(data, 4, n, g_wltrace);
}
static void
wtr8(int32 tag, int32 n, const uint8 *d)
{
    int32 i, *w;
    if (!g_wltrace) return;
    w = ckd_calloc(n + 1, 4);
    for (i = 0; i < n; i++) w[i] = d[i];
    wtr(tag, n, w);
    ckd_free(w);
}
#endif
static void
tr(int32 tag, int32 n, const void *data)
{
    if (!g_trace) return;
    fwrite(&tag, 4, 1, g_trace); fwrite(&n, 4, 1, g_trace);
    if (n) fwrite(data, 4, n, g_trace);
}
static void
tr16(int32 tag, int32 n, const int16 *d)
{
    int32 i, *w;
    if (!g_trace) return;
    w = ckd_calloc(n + 1, 4);
    for (i = 0; i < n; i++) w[i] = d[i];
    tr(tag, n, w);
    ckd_free(w);
}
static void
tr8(int32 tag, int32 n, const uint8 *d)
{
    int32 i, *w;
    if (!g_trace) return;
    w = ckd_calloc(n + 1, 4);
    for (i = 0; i < n; i++) w[i] = d[i];
    tr(tag, n, w);
    ckd_free(w);
}
static void
trace_state(int32 tag_base)
{
    int32 t;
    if (!g_trace) return;
    for (t = 0; t < g_ntree; t++) {
        s3o_lextree_t *lt = g_lt[t];
        int32 n = lt->n_node, i, *buf = ckd_calloc(10 * n + 4, 4);
        tr(tag_base + 0, lt->n_active, lt->active);
        tr(tag_base + 1, lt->n_next_active, lt->next_active);
        for (i = 0; i < n; i++) {
            s3o_hmm_t *h = &lt->hmm[i];
            buf[10 * i + 0] = h->score[0]; buf[10 * i + 1] = h->score[1]; buf[10 * i + 2] = h->score[2];
            buf[10 * i + 3] = (int32)h->history[0]; buf[10 * i + 4] = (int32)h->history[1];
            buf[10 * i + 5] = (int32)h->history[2]; buf[10 * i + 6] = h->out_score;
            buf[10 * i + 7] = (int32)h->out_history; buf[10 * i + 8] = h->bestscore; buf[10 * i + 9] = h->frame;
        }
        tr(tag_base + 2, 10 * n, buf);
        ckd_free(buf);
    }
}

#define utt_decode_adapt utt_decode

static void
backend_init(kb_t *kb, srch_TST_graph_t *tstg)
{
    kbcore_t *kbc = kb->kbcore;
    mdef_t *mdef = kbcore_mdef(kbc);
    dict2pid_t *d2p = kbcore_dict2pid(kbc);
    tmat_t *tmat = kbcore_tmat(kbc);
    int32 ne = mdef_n_emit_state(mdef), i, j, k, n;

    if (!dict2pid_is_composite(d2p))
        E_FATAL("tst shim: full cross-word triphone expansion grows the lextree during search; "
                "only composite triphones (the reference's default) are supported\n");
    if (kb->pl->pheurtype != 0 && !getenv("S3A_UTT"))
        E_FATAL("tst shim: -pheurtype > 0 is served by the whole-utterance engine only (S3A_UTT=lanes)\n");
    if (kbcore_lmset(kbc)->n_lm != 1)
        E_FATAL("tst shim: exactly one LM is supported\n");

    g_ntree = 2 * tstg->n_lextree;
    g_flat = ckd_calloc(g_ntree, sizeof(*g_flat));
    for (i = 0; i < g_ntree; i++) {
        lextree_t *lt = (i < tstg->n_lextree) ? tstg->curugtree[i] : tstg->fillertree[i - tstg->n_lextree];
        g_flat[i] = flatten_tree(lt);
        if (g_flat[i]->n_node > g_max_node) g_max_node = g_flat[i]->n_node;
    }
    g_tp_flat = ckd_calloc(tmat->n_tmat * ne * (ne + 1), 4);
    for (i = 0; i < tmat->n_tmat; i++)
        for (j = 0; j < ne; j++)
            for (k = 0; k <= ne; k++)
                g_tp_flat[(i * ne + j) * (ne + 1) + k] = tmat->tp[i][j][k];
    g_sseq_flat = ckd_calloc(mdef_n_sseq(mdef) * ne, 2);
    for (i = 0; i < mdef_n_sseq(mdef); i++)
        for (j = 0; j < ne; j++)
            g_sseq_flat[i * ne + j] = mdef->sseq[i][j];
    g_comsseq_flat = ckd_calloc(d2p->n_comsseq * ne + 1, 2);
    for (i = 0; i < d2p->n_comsseq; i++)
        for (j = 0; j < ne; j++)
            g_comsseq_flat[i * ne + j] = d2p->comsseq[i][j];
    for (i = 0, n = 0; i < d2p->n_comstate; i++)
        for (j = 0; IS_S3SENID(d2p->comstate[i][j]); j++)
            n++;
    g_n_comstate = d2p->n_comstate;
    g_comstate_off = ckd_calloc(g_n_comstate + 1, 4);
    g_comstate = ckd_calloc(n + 1, 2);
    for (i = 0, n = 0; i < g_n_comstate; i++) {
        g_comstate_off[i] = n;
        for (j = 0; IS_S3SENID(d2p->comstate[i][j]); j++)
            g_comstate[n++] = d2p->comstate[i][j];
    }
    g_comstate_off[g_n_comstate] = n;

    g_best = ckd_calloc(g_ntree, 4); g_wbest = ckd_calloc(g_ntree, 4); g_nact = ckd_calloc(g_ntree, 4);
    g_exit_n = ckd_calloc(g_ntree, 4);
    g_exit_wid = ckd_calloc(g_ntree * g_max_node, 4);
    g_exit_scr = ckd_calloc(g_ntree * g_max_node, 4);
    g_exit_hist = ckd_calloc(g_ntree * g_max_node, 4);

    if (getenv("S3O_TRACE") && (g_trace = fopen(getenv("S3O_TRACE"), "wb")) != NULL) {
        int32 hdr[8] = { g_ntree, ne, tmat->n_tmat, mdef_n_sseq(mdef), d2p->n_comsseq, g_n_comstate,
                         mdef_n_sen(mdef), d2p->n_comstate };
        tr(1, 8, hdr);
        tr(2, tmat->n_tmat * ne * (ne + 1), g_tp_flat);
        tr16(3, mdef_n_sseq(mdef) * ne, g_sseq_flat);
        tr16(4, d2p->n_comsseq * ne, g_comsseq_flat);
        tr(5, g_n_comstate + 1, g_comstate_off);
        tr16(6, g_comstate_off[g_n_comstate], g_comstate);
        for (i = 0; i < g_ntree; i++) {
            flat_t *f = g_flat[i];
            int32 h2[4] = { f->n_node, f->n_lc, f->n_root, f->type };
            tr(10, 4, h2); tr(11, f->n_node, f->ssid); tr(12, f->n_node, f->tmatid);
            tr8(13, f->n_node, f->composite); tr(14, f->n_node, f->wid); tr(15, f->n_node, f->prob);
            tr(16, f->n_node + 1, f->child_off); tr(17, f->child_off[f->n_node], f->child);
            if (f->n_lc) { tr16(18, f->n_lc, f->lc); tr(19, f->n_lc + 1, f->lcroot_off); tr(20, f->lcroot_off[f->n_lc], f->lcroot); }
            tr(21, f->n_root, f->root);
        }
    }
    g_lt = ckd_calloc(g_ntree, sizeof(*g_lt));
    for (i = 0; i < g_ntree; i++) {
        flat_t *f = g_flat[i];
        g_lt[i] = s3o_lextree_init(f->n_node, f->ssid, f->tmatid, f->composite, f->wid, f->prob,
                                   f->child_off, f->child, f->n_lc, f->lc, f->lcroot_off, f->lcroot,
                                   f->n_root, f->root, ne, g_tp_flat, g_sseq_flat, g_comsseq_flat);
    }
#ifdef WL_ORACLE
    {
        wl_flat_t *w = g_wl = flatten_lm(kbc);
        vithist_t *vh = tstg->vithist;
        g_olm.n_ug = w->n_ug; g_olm.n_bg = w->n_bg; g_olm.n_tg = w->n_tg;
        g_olm.ug_prob = w->ug_prob; g_olm.ug_bowt = w->ug_bowt; g_olm.ug_firstbg = w->ug_firstbg;
        g_olm.bg_wid = w->bg_wid; g_olm.bg_prob = w->bg_prob; g_olm.bg_bowt = w->bg_bowt; g_olm.bg_firsttg = w->bg_firsttg;
        g_olm.tg_wid = w->tg_wid; g_olm.tg_prob = w->tg_prob; g_olm.inclass = w->inclass;
        g_od.n_word = w->n_word; g_od.n_ci = w->n_ci; g_od.lwid = w->lwid; g_od.is_filler = w->is_filler;
        g_od.fillpen = w->fillpen; g_od.last_ci = w->last_ci; g_od.startwid = w->startwid;
        g_od.finishwid = w->finishwid; g_od.silwid = w->silwid; g_od.start_lwid = w->start_lwid;
        g_od.finish_lwid = w->finish_lwid;
        g_ovh = s3o_vithist_init(1 << 22, S3_MAX_FRAMES, vh->wbeam, vh->bghist);
        if (getenv("S3O_WLTRACE") && (g_wltrace = fopen(getenv("S3O_WLTRACE"), "wb")) != NULL) {
            int32 hdr[17] = { w->n_ug, w->n_bg, w->n_tg, w->n_word, w->n_ci, w->startwid, w->finishwid, w->silwid,
                              w->start_lwid, w->finish_lwid, vh->wbeam, vh->bghist, tstg->histprune->maxwpf,
                              tstg->histprune->maxhistpf, tstg->n_lextree, tstg->epl, kb->beam->wordend };
            wtr(1, 17, hdr);
            wtr(2, w->n_ug, w->ug_prob); wtr(3, w->n_ug, w->ug_bowt); wtr(4, w->n_ug + 1, w->ug_firstbg);
            wtr(5, w->n_bg, w->bg_wid); wtr(6, w->n_bg, w->bg_prob); wtr(7, w->n_bg, w->bg_bowt);
            wtr(8, w->n_bg ? w->n_bg + 1 : 0, w->bg_firsttg); wtr(9, w->n_tg, w->tg_wid); wtr(10, w->n_tg, w->tg_prob);
            wtr(11, w->n_word, w->lwid); wtr8(12, w->n_word, w->is_filler); wtr(13, w->n_word, w->fillpen);
            wtr(14, w->n_word, w->last_ci);
        }
    }
#endif
    E_INFO("tst shim: %d lextrees flattened (largest %d nodes), backend %s\n", g_ntree, g_max_node,
           "CPU oracle (oracle/s3o_lextree.c)"
        );
}

/* one batch of lextree_enter calls into tree t */
static void
be_enter(int32 t, int32 n, int32 *lc, int32 *scr, int32 *hist, int32 cf, int32 thresh)
{
    int32 c;
    if (g_trace) {
        int32 hdr[4] = { t, n, cf, thresh };
        tr(30, 4, hdr); tr(31, n, lc); tr(32, n, scr); tr(33, n, hist);
    }
    for (c = 0; c < n; c++)
        s3o_lextree_enter(g_lt[t], lc[c], cf, scr[c], hist[c], thresh);
}

static void
be_swap(int32 cf)
{
    int32 t;
    (void)cf;
    for (t = 0; t < g_ntree; t++) s3o_lextree_active_swap(g_lt[t]);
}

/* ------------------------------------------------------------------ */
/* replaced srch_funcs_t slots                                         */
/* ------------------------------------------------------------------ */
static int
tst_begin(void *srch)
{
    srch_t *s = srch;
    srch_TST_graph_t *tstg = s->grh->graph_struct;
    kbcore_t *kbc = s->kbc;
    mgau_model_t *g = kbc->mgau;
    int32 pred, i, lc, zero = 0;

    g_t_utt -= now_s();
    vithist_utt_reset(tstg->vithist);
    histprune_zero_histbin(tstg->histprune);
    pred = vithist_utt_begin(tstg->vithist, kbc);
#ifdef WL_ORACLE
    s3o_vithist_utt_begin(g_ovh, g_od.startwid, g_od.start_lwid);
#endif
    if (g)
        for (i = 0; i < g->n_mgau; i++) { g->mgau[i].bstidx = NO_BSTIDX; g->mgau[i].updatetime = NOT_UPDATED; }
    lc = mdef_silphone(kbc->mdef);
    be_enter(0, 1, &lc, &zero, &pred, -1, s->beam->hmm);
    lc = BAD_S3CIPID;
    be_enter(tstg->n_lextree, 1, &lc, &zero, &pred, -1, s->beam->hmm);
    tstg->n_lextrans = 1;
    be_swap(-1);
    return SRCH_SUCCESS;
}

static int
tst_end(void *srch)
{
    srch_t *s = srch;
    srch_TST_graph_t *tstg = s->grh->graph_struct;
    int32 t;
    g_t_utt += now_s();
#ifdef WL_ORACLE
    if (g_ovh->overflow) E_FATAL("tst shim: the oracle's history table overflowed\n");
    vithist_fill(tstg->vithist, g_ovh->n_entry, g_ovh->n_frm, g_ovh->score, g_ovh->pred, g_ovh->lw0, g_ovh->lw1,
                 g_ovh->wid, g_ovh->sf, g_ovh->ef, g_ovh->ascr, g_ovh->lscr, g_ovh->type, g_ovh->frame_start,
                 g_ovh->bestscore, g_ovh->bestvh, kbcore_lm(s->kbc));
    if (g_wltrace) { int32 z = 0; wtr(99, 1, &z); fclose(g_wltrace); g_wltrace = NULL; }
#endif
    s->exit_id = vithist_utt_end(tstg->vithist, s->kbc);
    s->stat->utt_wd_exit = vithist_n_entry(tstg->vithist);
    histprune_showhistbin(tstg->histprune, s->stat->nfr, s->uttid);
    for (t = 0; t < g_ntree; t++) s3o_lextree_utt_end(g_lt[t]);
    if (g_trace) { int32 z = 0; tr(99, 1, &z); fclose(g_trace); g_trace = NULL; }
    g_trace_utt++;
    lm_cache_stats_dump(kbcore_lm(s->kbc));
    lm_cache_reset(kbcore_lm(s->kbc));
    return (s->exit_id >= 0) ? SRCH_SUCCESS : SRCH_FAILURE;
}

static int
tst_select_active(void *srch)
{
    srch_t *s = srch;
    ascr_t *ascr = s->ascr;
    mdef_t *mdef = kbcore_mdef(s->kbc);
    dict2pid_t *d2p = kbcore_dict2pid(s->kbc);
    int32 t;
    if (!ascr->sen_active) return SRCH_SUCCESS;
    ascr_clear_ssid_active(ascr);
    ascr_clear_comssid_active(ascr);
    for (t = 0; t < g_ntree; t++)
        s3o_lextree_ssid_active(g_lt[t], ascr->ssid_active, ascr->comssid_active);
    ascr_clear_sen_active(ascr);
    s3o_sseq2sen_active(g_sseq_flat, mdef_n_sseq(mdef), mdef_n_emit_state(mdef), ascr->ssid_active,
                        ascr->sen_active);
    s3o_comsseq2sen_active(g_comsseq_flat, d2p->n_comsseq, mdef_n_emit_state(mdef), g_comstate_off,
                           g_comstate, ascr->comssid_active, ascr->sen_active);
    return SRCH_SUCCESS;
}

static int
tst_hmm_compute_lv2(void *srch, int32 frmno)
{
    srch_t *s = srch;
    srch_TST_graph_t *tstg = s->grh->graph_struct;
    histprune_t *hp = tstg->histprune;
    beam_t *bm = s->beam;
    int32 besthmmscr = MAX_NEG_INT32, bestwordscr = MAX_NEG_INT32, frm_nhmm = 0, t, hb, pb, wb;

    if (g_trace) {
        int32 fr = frmno;
        tr(40, 1, &fr);
        tr(41, mdef_n_sen(kbcore_mdef(s->kbc)), s->ascr->senscr);
        tr(42, kbcore_dict2pid(s->kbc)->n_comstate, s->ascr->comsen);
        tr8(43, mdef_n_sen(kbcore_mdef(s->kbc)), s->ascr->sen_active);
    }
    for (t = 0; t < g_ntree; t++) {
        s3o_lextree_hmm_eval(g_lt[t], s->ascr->senscr, s->ascr->comsen, frmno);
        g_best[t] = g_lt[t]->best; g_wbest[t] = g_lt[t]->wbest; g_nact[t] = g_lt[t]->n_active;
    }
    for (t = 0; t < g_ntree; t++) {
        if (besthmmscr < g_best[t]) besthmmscr = g_best[t];
        if (bestwordscr < g_wbest[t]) bestwordscr = g_wbest[t];
        s->stat->utt_hmm_eval += g_nact[t];
        frm_nhmm += g_nact[t];
    }
    if (besthmmscr > 0)
        E_ERROR("***ERROR*** Fr %d, best HMM score > 0 (%d); int32 wraparound?\n", frmno, besthmmscr);
    if (frm_nhmm / hp->hmm_hist_binsize > hp->hmm_hist_bins - 1)
        hp->hmm_hist[hp->hmm_hist_bins - 1]++;
    else
        hp->hmm_hist[frm_nhmm / hp->hmm_hist_binsize]++;

    if (frm_nhmm > (hp->maxhmmpf + (hp->maxhmmpf >> 1))) {
        int32 nbin = 1000, bw = -(bm->hmm) / nbin, i, j;
        int32 *bin = ckd_calloc(nbin, sizeof(int32));
        for (t = 0; t < g_ntree; t++)
            s3o_lextree_hmm_histbin(g_lt[t], besthmmscr, bin, nbin, bw);
        for (i = 0, j = 0; (i < nbin) && (j < hp->maxhmmpf); i++, j += bin[i]);
        ckd_free(bin);
        g_histframes++;
        hb = -(i * bw);
        pb = (hb > bm->ptrans) ? hb : bm->ptrans;
        wb = (hb > bm->word) ? hb : bm->word;
    }
    else {
        hb = bm->hmm; pb = bm->ptrans; wb = bm->word;
    }
    bm->bestscore = besthmmscr;
    bm->bestwordscore = bestwordscr;
    bm->thres = bm->bestscore + hb;
    bm->phone_thres = bm->bestscore + pb;
    bm->word_thres = bm->bestwordscore + wb;
    g_frames++;
    if (g_trace) {
        int32 r[8] = { bm->bestscore, bm->bestwordscore, frm_nhmm, bm->thres, bm->phone_thres, bm->word_thres,
                       bm->hmm, bm->ptrans };
        tr(44, 8, r); tr(45, g_ntree, g_best); tr(46, g_ntree, g_wbest); tr(47, g_ntree, g_nact);
        trace_state(50);
    }
    return SRCH_SUCCESS;
}

static int
tst_propagate_ph_lv2(void *srch, int32 frmno)
{
    srch_t *s = srch;
    beam_t *bm = s->beam;
    int32 pth = bm->phone_thres;
    if (bm->ptranskip != 0 && (frmno % bm->ptranskip) == 0)
        pth = bm->word_thres;           /* srch_time_switch_tree.c:975-1003 */
    {
        int32 t;
        for (t = 0; t < g_ntree; t++)
            s3o_lextree_hmm_propagate_non_leaves(g_lt[t], frmno, bm->thres, pth, bm->word_thres);
        if (g_trace) { int32 r[3] = { bm->thres, pth, bm->word_thres }; tr(60, 3, r); trace_state(61); }
    }
    return SRCH_SUCCESS;
}

/* srch_utt_word_trans, srch_time_switch_tree.c:1087-1179 */
static void
tst_word_trans(srch_t *s, int32 cf)
{
    srch_TST_graph_t *tstg = s->grh->graph_struct;
    vithist_t *vh = tstg->vithist;
    beam_t *bm = s->beam;
    dict_t *dict = kbcore_dict(s->kbc);
    mdef_t *mdef = kbcore_mdef(s->kbc);
    int32 n_ci = mdef_n_ciphone(mdef), th = bm->bestscore + bm->hmm;
    int32 *bs = bm->wordbestscores, *bv = bm->wordbestexits;
    int32 p, vhid, le, k, n, maxpscore = MAX_NEG_INT32;
    static __thread int32 *c_lc, *c_scr, *c_hist;

    if (vh->bestvh[cf] < 0)
        return;
    if (!c_lc) { c_lc = ckd_calloc(n_ci + 1, 4); c_scr = ckd_calloc(n_ci + 1, 4); c_hist = ckd_calloc(n_ci + 1, 4); }
    for (p = 0; p < n_ci; p++) { bs[p] = MAX_NEG_INT32; bv[p] = -1; }
    vhid = vithist_first_entry(vh, cf);
    le = vithist_n_entry(vh) - 1;
    for (; vhid <= le; vhid++) {
        vithist_entry_t *ve = vithist_id2entry(vh, vhid);
        int32 score;
        if (!vithist_entry_valid(ve))
            continue;
        p = dict_last_phone(dict, vithist_entry_wid(ve));
        if (mdef_is_fillerphone(mdef, p))
            p = mdef_silphone(mdef);
        score = vithist_entry_score(ve);
        if (score > bs[p]) {
            bs[p] = score;
            bv[p] = vhid;
            if (maxpscore < score) maxpscore = score;
        }
    }
    k = tstg->n_lextrans++;
    k = (k % (tstg->n_lextree * tstg->epl)) / tstg->epl;
    for (p = 0, n = 0; p < n_ci; p++)
        if (bv[p] >= 0 && (bm->wordend == 0 || bs[p] > bm->wordend + maxpscore)) {
            c_lc[n] = p; c_scr[n] = bs[p]; c_hist[n] = bv[p]; n++;
        }
    be_enter(k, n, c_lc, c_scr, c_hist, cf, th);
    c_lc[0] = BAD_S3CIPID; c_scr[0] = vh->bestscore[cf]; c_hist[0] = vh->bestvh[cf];
    be_enter(tstg->n_lextree + k, 1, c_lc, c_scr, c_hist, cf, th);
}

#ifdef WL_ORACLE
static void
tst_word_trans_oracle(srch_t *s, int32 cf)
{
    srch_TST_graph_t *tstg = s->grh->graph_struct;
    beam_t *bm = s->beam;
    int32 th = bm->bestscore + bm->hmm, n, k, fscr, fhist, lcb = BAD_S3CIPID;
    static __thread int32 *c_lc, *c_scr, *c_hist;
    const int32 fs = g_ovh->frame_start[cf], ne = g_ovh->n_entry - fs;
    if (!c_lc) { c_lc = ckd_calloc(g_od.n_ci + 1, 4); c_scr = ckd_calloc(g_od.n_ci + 1, 4); c_hist = ckd_calloc(g_od.n_ci + 1, 4); }
    n = s3o_word_trans(g_ovh, &g_od, cf, bm->wordend, c_lc, c_scr, c_hist, &fscr, &fhist);
    {   /* the frame's surviving entries and its lextree_enter calls */
        int32 hdr[6] = { cf, ne, n, g_ovh->bestscore[cf], g_ovh->bestvh[cf], th };
        wtr(30, 6, hdr);
        wtr(31, ne, g_ovh->wid + fs); wtr(32, ne, g_ovh->score + fs); wtr(33, ne, g_ovh->pred + fs);
        wtr(34, ne, g_ovh->lw0 + fs); wtr(35, ne, g_ovh->lw1 + fs); wtr(36, ne, g_ovh->ascr + fs);
        wtr(37, ne, g_ovh->lscr + fs); wtr(38, ne, g_ovh->sf + fs); wtr(39, ne, g_ovh->type + fs);
        if (n > 0) { wtr(40, n, c_lc); wtr(41, n, c_scr); wtr(42, n, c_hist); }
    }
    if (n < 0) return;
    k = tstg->n_lextrans++;
    k = (k % (tstg->n_lextree * tstg->epl)) / tstg->epl;
    be_enter(k, n, c_lc, c_scr, c_hist, cf, th);
    be_enter(tstg->n_lextree + k, 1, &lcb, &fscr, &fhist, cf, th);
}
#endif

static int
tst_propagate_wd_lv2(void *srch, int32 frmno)
{
    srch_t *s = srch;
    srch_TST_graph_t *tstg = s->grh->graph_struct;
    histprune_t *hp = tstg->histprune;
    vithist_t *vh = tstg->vithist;
    int32 t, i;

    /* srch_TST_rescoring: word exits of every tree, in tree then active-list order */
    for (t = 0; t < g_ntree; t++) {
        g_exit_n[t] = s3o_lextree_hmm_propagate_leaves(g_lt[t], s->beam->word_thres,
                                                       g_exit_wid + t * g_max_node, g_exit_scr + t * g_max_node,
                                                       g_exit_hist + t * g_max_node, g_max_node);
        if (g_exit_n[t] < 0) { E_ERROR("out.history==-1, error\n"); return SRCH_FAILURE; }
        if (g_trace) {
            tr(70, 1, &g_exit_n[t]); tr(71, g_exit_n[t], g_exit_wid + t * g_max_node);
            tr(72, g_exit_n[t], g_exit_scr + t * g_max_node); tr(73, g_exit_n[t], g_exit_hist + t * g_max_node);
        }
    }
#ifdef WL_ORACLE
    {
        int32 hdr[4] = { frmno, 0, s->beam->word_thres - s->beam->bestwordscore, s->beam->bestscore + s->beam->hmm };
        for (t = 0; t < g_ntree; t++) hdr[1] += g_exit_n[t];
        wtr(20, 4, hdr);
        for (t = 0; t < g_ntree; t++) {
            int32 ty = g_flat[t]->type;
            wtr(21, 1, &ty); wtr(22, g_exit_n[t], g_exit_wid + t * g_max_node);
            wtr(23, g_exit_n[t], g_exit_scr + t * g_max_node); wtr(24, g_exit_n[t], g_exit_hist + t * g_max_node);
        }
    }
    for (t = 0; t < g_ntree; t++)
        for (i = 0; i < g_exit_n[t]; i++)
            if (s3o_vithist_rescore(g_ovh, &g_olm, &g_od, g_exit_wid[t * g_max_node + i], frmno,
                                    g_exit_scr[t * g_max_node + i], g_exit_hist[t * g_max_node + i],
                                    g_flat[t]->type) < 0)
                E_FATAL("Hmm->out.history equals to -1 with score %d, some active phone was not computed?\n",
                        g_exit_scr[t * g_max_node + i]);
#else
    for (t = 0; t < g_ntree; t++)
        for (i = 0; i < g_exit_n[t]; i++)
            vithist_rescore(vh, s->kbc, g_exit_wid[t * g_max_node + i], frmno,
                            g_exit_scr[t * g_max_node + i], g_exit_hist[t * g_max_node + i],
                            g_flat[t]->type, -1);
#endif
    {
        double t1 = now_s();
#ifdef WL_ORACLE
        (void)vh;
        s3o_vithist_prune(g_ovh, &g_od, frmno, hp->maxwpf, hp->maxhistpf,
                          s->beam->word_thres - s->beam->bestwordscore, NULL);
        tst_word_trans_oracle(s, frmno);
#else
        vithist_prune(vh, kbcore_dict(s->kbc), frmno, hp->maxwpf, hp->maxhistpf,
                      s->beam->word_thres - s->beam->bestwordscore);
        tst_word_trans(s, frmno);
#endif
        g_t_word += now_s() - t1;
    }
    return SRCH_SUCCESS;
}

static int
tst_frame_windup(void *srch, int32 frmno)
{
    srch_t *s = srch;
    srch_TST_graph_t *tstg = s->grh->graph_struct;
    vithist_frame_windup(tstg->vithist, frmno, NULL, s->kbc);
#ifdef WL_ORACLE
    s3o_vithist_frame_windup(g_ovh, frmno);
#endif
    be_swap(frmno);
    if (g_trace) trace_state(80);
    return SRCH_SUCCESS;
}

/* ------------------------------------------------------------------ */
/* driver                                                              */
/* ------------------------------------------------------------------ */
#include <pthread.h>

static void
install_slots(srch_t *s)
{
    s->funcs->utt_begin = tst_begin;
    s->funcs->utt_end = tst_end;
    s->funcs->select_active_gmm = tst_select_active;
    s->funcs->hmm_compute_lv2 = tst_hmm_compute_lv2;
    s->funcs->propagate_graph_ph_lv2 = tst_propagate_ph_lv2;
    s->funcs->propagate_graph_wd_lv2 = tst_propagate_wd_lv2;
    s->funcs->frame_windup = tst_frame_windup;
}

/*
 * One decoder = one kb_t + one set of device objects + one HIP stream.  With
 * S3A_STREAMS=N (device build) N decoders run in N host threads of ONE process, each on
 * its contiguous shard of the control file (-ctloffset/-ctlcount, the reference's own
 * sharding device), sharing the GPU through their streams: utterances are independent, so
 * there is no cross-stream communication; the per-shard -hyp/-hypseg files are
 * concatenated in control-file order at the end.
 */
typedef struct {
    int id, n, argc;
    char **argv;
    int32 off, cnt;
    long frames, histframes;
    double t_utt, t_search, t_word;
} worker_t;

static pthread_mutex_t g_init_lock = PTHREAD_MUTEX_INITIALIZER;
static pthread_barrier_t g_start;
static double g_t_start;
static char g_hyp[2][4096];

static void *
worker_main(void *vp)
{
    worker_t *w = vp;
    kb_t kb;
    cmd_ln_t *config;
    char part[2][4200], so[32], sc[32];
    char **av = ckd_calloc(w->argc + 16, sizeof(char *));
    int ac = 0, i;

    for (i = 0; i < w->argc; i++) {
        if (i > 0 && (!strcmp(w->argv[i], "-hyp") || !strcmp(w->argv[i], "-hypseg")
                      || !strcmp(w->argv[i], "-ctloffset") || !strcmp(w->argv[i], "-ctlcount"))) { i++; continue; }
        av[ac++] = w->argv[i];
    }
    if (g_hyp[0][0]) { snprintf(part[0], sizeof part[0], "%s.part%03d", g_hyp[0], w->id); av[ac++] = "-hyp"; av[ac++] = part[0]; }
    if (g_hyp[1][0]) { snprintf(part[1], sizeof part[1], "%s.part%03d", g_hyp[1], w->id); av[ac++] = "-hypseg"; av[ac++] = part[1]; }
    snprintf(so, sizeof so, "%d", w->off); snprintf(sc, sizeof sc, "%d", w->cnt);
    av[ac++] = "-ctloffset"; av[ac++] = so; av[ac++] = "-ctlcount"; av[ac++] = sc;

    pthread_mutex_lock(&g_init_lock);           /* model / dictionary / LM loading is not re-entrant */
    config = cmd_ln_parse_r(NULL, arg, ac, av, TRUE);
    kb_init(&kb, config);
    if (((srch_t *)kb.srch)->op_mode != 4)
        E_FATAL("tst shim: -op_mode 4 (fwdtree) only\n");
    backend_init(&kb, (srch_TST_graph_t *)((srch_t *)kb.srch)->grh->graph_struct);
    install_slots(kb.srch);
    pthread_mutex_unlock(&g_init_lock);
    if (pthread_barrier_wait(&g_start) == PTHREAD_BARRIER_SERIAL_THREAD)
        g_t_start = now_s();        /* every decoder is loaded: the decode clock starts here */

    if (w->cnt > 0)
        kb.stat->tm = ctl_process(cmd_ln_str_r(config, "-ctl"), cmd_ln_str_r(config, "-ctl_lm"),
                                  cmd_ln_str_r(config, "-ctl_mllr"), w->off, w->cnt, utt_decode_adapt, &kb);
    if (kb.matchsegfp) fclose(kb.matchsegfp);
    if (kb.matchfp) fclose(kb.matchfp);
    w->frames = g_frames; w->histframes = g_histframes; w->t_utt = g_t_utt; w->t_search = g_t_search; w->t_word = g_t_word;
    return NULL;
}

static void
concat_parts(const char *dst, int n)
{
    FILE *out = fopen(dst, "w");
    char path[4200], buf[65536];
    int i;
    size_t k;
    if (!out) E_FATAL("cannot write %s\n", dst);
    for (i = 0; i < n; i++) {
        FILE *in;
        snprintf(path, sizeof path, "%s.part%03d", dst, i);
        if ((in = fopen(path, "r")) == NULL) continue;
        while ((k = fread(buf, 1, sizeof buf, in)) > 0) fwrite(buf, 1, k, out);
        fclose(in);
        remove(path);
    }
    fclose(out);
}


/* ------------------------------------------------------------------ */
/* S3A_LIVE=gpu|cpu: the reference's LIVE API (libAPI/s3_decode.c)     */
/* ------------------------------------------------------------------ */
/*
 * s3_decode_init / _begin_utt / _process (cepstra in blocks) / _end_utt / _hypothesis, as an application embedding
 * the decoder calls them.  s3_decode_init runs kb_init; an integrator installs the replacement slots right after it
 * (two calls), everything else is the reference's: feat_s2mfc2feat_live, utt_decode_block -> srch_utt_decode_blk ->
 * the per-frame slots.  S3A_LIVE=cpu leaves the table alone (the expected output of the test).
 */
static float32 **
read_cep(const char *path, int32 ceplen, int32 *n_out)
{
    FILE *fp = fopen(path, "rb");
    int32 n, i, swap = 0;
    long sz;
    float32 **c;
    if (!fp) E_FATAL("cannot read %s\n", path);
    fseek(fp, 0, SEEK_END); sz = ftell(fp); fseek(fp, 0, SEEK_SET);
    if (fread(&n, 4, 1, fp) != 1) E_FATAL("%s: empty\n", path);
    if ((long)n * 4 + 4 != sz) { SWAP_INT32(&n); swap = 1; }
    if ((long)n * 4 + 4 != sz || n % ceplen) E_FATAL("%s: not a cepstrum file\n", path);
    c = (float32 **)ckd_calloc_2d(n / ceplen + 1, ceplen, sizeof(float32));
    if (fread(c[0], 4, n, fp) != (size_t)n) E_FATAL("%s: short read\n", path);
    if (swap) for (i = 0; i < n; i++) SWAP_FLOAT32(&c[0][i]);
    fclose(fp);
    *n_out = n / ceplen;
    return c;
}

static int
live_mode_main(int argc, char *argv[], int use_gpu)
{
    s3_decode_t d;
    cmd_ln_t *config;
    const char *cepdir, *cepext;
    arg_t *defs;
    int n_live = 0, n_dec = 0, a, b2, n = 0;
    char line[4096], uttid[4096], path[8192];
    FILE *ctl;
    int32 ceplen, block = getenv("S3A_LIVE_BLOCK") ? atoi(getenv("S3A_LIVE_BLOCK")) : 37, n_utt = 0;

    /* the live decoder's own argument table + what sphinx3_decode's table has on top of it (-cepdir, -cepext, ...) */
    while (S3_DECODE_ARG_DEFS[n_live].name) n_live++;
    while (arg[n_dec].name) n_dec++;
    defs = (arg_t *)ckd_calloc(n_live + n_dec + 1, sizeof(arg_t));
    for (a = 0; a < n_live; a++) defs[n++] = S3_DECODE_ARG_DEFS[a];
    for (a = 0; a < n_dec; a++) {
        for (b2 = 0; b2 < n_live && strcmp(arg[a].name, S3_DECODE_ARG_DEFS[b2].name); b2++) ;
        if (b2 == n_live) defs[n++] = arg[a];
    }
    cmd_ln_appl_enter(argc, argv, "default.arg", defs);     /* (parts of the reference read the global table) */
    config = cmd_ln_get();
    if (!config) E_FATAL("bad arguments\n");
    memset(&d, 0, sizeof d);
    if (s3_decode_init(&d, config) != S3_DECODE_SUCCESS) E_FATAL("s3_decode_init failed\n");
    if (use_gpu) {
        srch_t *s = (srch_t *)d.kb.srch;
        if (s->op_mode != 4) E_FATAL("tst shim: -op_mode 4 (fwdtree) only\n");
        backend_init(&d.kb, (srch_TST_graph_t *)s->grh->graph_struct);
        install_slots(s);
    }
    cepdir = cmd_ln_str_r(config, "-cepdir"); cepext = cmd_ln_str_r(config, "-cepext");
    ceplen = feat_cepsize(kbcore_fcb(d.kbcore));
    if ((ctl = fopen(cmd_ln_str_r(config, "-ctl"), "r")) == NULL) E_FATAL("cannot read the control file\n");
    while (fgets(line, sizeof line, ctl)) {
        float32 **cep;
        int32 nfr, f;
        char *hyp = NULL, *id = NULL;
        hyp_t **segs = NULL;
        if (sscanf(line, "%4095s", uttid) != 1 || uttid[0] == '#') continue;
        snprintf(path, sizeof path, "%s/%s%s", cepdir ? cepdir : ".", uttid, cepext ? cepext : ".mfc");
        cep = read_cep(path, ceplen, &nfr);
        if (s3_decode_begin_utt(&d, uttid) != S3_DECODE_SUCCESS) E_FATAL("s3_decode_begin_utt failed\n");
        for (f = 0; f < nfr; f += block)
            if (s3_decode_process(&d, cep + f, nfr - f < block ? nfr - f : block) != S3_DECODE_SUCCESS)
                E_FATAL("s3_decode_process failed\n");
        s3_decode_end_utt(&d);
        if (s3_decode_hypothesis(&d, &id, &hyp, &segs) != S3_DECODE_SUCCESS) E_FATAL("s3_decode_hypothesis failed\n");
        printf("LIVE %s:", id ? id : uttid);
        for (; segs && *segs; segs++)
            printf(" %s(%d,%d,%d,%d)", dict_wordstr(kbcore_dict(d.kbcore), (*segs)->id), (*segs)->sf, (*segs)->ef, (*segs)->ascr, (*segs)->lscr);
        printf(" | %s\n", hyp ? hyp : "");
        ckd_free_2d((void **)cep);
        n_utt++;
    }
    fclose(ctl);
    E_INFO("tst shim live mode: %d utterances through s3_decode_process in blocks of %d frames, %s slots, %ld frames searched by the replacement backend\n",
           n_utt, block, use_gpu ? "replacement" : "reference", g_frames);
    if (d.kb.matchsegfp) { fclose(d.kb.matchsegfp); d.kb.matchsegfp = NULL; }
    if (d.kb.matchfp) { fclose(d.kb.matchfp); d.kb.matchfp = NULL; }
    fflush(stdout);
    return 0;
}

int
main(int argc, char *argv[])
{
    int n_streams = getenv("S3A_STREAMS") ? atoi(getenv("S3A_STREAMS")) : 1, i;
    cmd_ln_t *config;
    worker_t *w;
    pthread_t *th;
    int32 n_utt = 0, base, extra, off;
    long frames = 0, histframes = 0;
    double t_utt = 0, t_search = 0, t_word = 0, wall;
    char line[16384];
    FILE *fp;

    if (getenv("S3A_LIVE"))
        return live_mode_main(argc, argv, strcmp(getenv("S3A_LIVE"), "cpu") != 0);
    cmd_ln_appl_enter(argc, argv, "default.arg", arg);      /* `arg`: the reference's own table */
    unlimit();
    config = cmd_ln_get();
    if (!cmd_ln_str_r(config, "-ctl"))
        E_FATAL("-ctl is required\n");
    if (n_streams < 1) n_streams = 1;
    if (cmd_ln_str_r(config, "-hyp")) snprintf(g_hyp[0], sizeof g_hyp[0], "%s", cmd_ln_str_r(config, "-hyp"));
    if (cmd_ln_str_r(config, "-hypseg")) snprintf(g_hyp[1], sizeof g_hyp[1], "%s", cmd_ln_str_r(config, "-hypseg"));
    if ((fp = fopen(cmd_ln_str_r(config, "-ctl"), "r")) == NULL)
        E_FATAL("cannot read the control file\n");
    while (fgets(line, sizeof line, fp))
        if (line[0] != '\n' && line[0] != '#') n_utt++;     /* as ctl_process counts lines */
    fclose(fp);
    if (cmd_ln_int32_r(config, "-ctlcount") < n_utt) n_utt = cmd_ln_int32_r(config, "-ctlcount");

    w = ckd_calloc(n_streams, sizeof(*w));
    th = ckd_calloc(n_streams, sizeof(*th));
    base = n_utt / n_streams; extra = n_utt % n_streams;
    off = cmd_ln_int32_r(config, "-ctloffset");
    wall = now_s();
    pthread_barrier_init(&g_start, NULL, n_streams);
    for (i = 0; i < n_streams; i++) {
        w[i].id = i; w[i].n = n_streams; w[i].argc = argc; w[i].argv = argv;
        w[i].off = off; w[i].cnt = base + (i < extra ? 1 : 0);
        off += w[i].cnt;
        pthread_create(&th[i], NULL, worker_main, &w[i]);
    }
    for (i = 0; i < n_streams; i++) {
        pthread_join(th[i], NULL);
        frames += w[i].frames; histframes += w[i].histframes; t_utt += w[i].t_utt; t_search += w[i].t_search; t_word += w[i].t_word;
    }
    {
        double t_end = now_s();
        E_INFO("tst shim decode-only wall (all decoders loaded -> last one done): %.3f s\n", t_end - g_t_start);
        t_word += 0;
        wall = t_end - wall;
        g_t_start = t_end - g_t_start;
    }
    if (g_hyp[0][0]) concat_parts(g_hyp[0], n_streams);
    if (g_hyp[1][0]) concat_parts(g_hyp[1], n_streams);
    if (frames == 0)
        E_FATAL("tst shim: the replaced slots were never called\n");
    E_INFO("tst shim: %ld frames searched by the replacement backend in %d stream(s)\n", frames, n_streams);
    E_INFO("tst shim: histogram pruning (lextree_hmm_histbin) applied in %ld frames\n", histframes);
    E_INFO("tst shim timing: %.1f us/frame inside utterances per stream (%.0f x real time per stream); of which "
           "frame_search (enqueue + the one sync) %.1f us, vithist_prune + word transitions %.1f us\n",
           1e6 * t_utt / frames, 0.01 * frames / (t_utt / n_streams) / n_streams, 1e6 * t_search / frames,
           1e6 * t_word / frames);
    E_INFO("tst shim throughput: %ld frames, decode-only %.3f s = %.0f x real time aggregate "
           "(%.3f s incl. loading %d decoders one after another)\n",
           frames, g_t_start, 0.01 * frames / g_t_start, wall, n_streams);
    return 0;
}
