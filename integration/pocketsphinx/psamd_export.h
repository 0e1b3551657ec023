/*
 * psamd_export.h -- what a pocketsphinx maintainer adds next to ngram_search_fwdtree.c: flatten the
 * search's read-only structures into the plain arrays of s3a_psfwd_desc_t (include/cmusphinx_amd.h).
 *
 * Reads, through pocketsphinx's own accessors,
 *   bin_mdef_t   sseq, n_ciphone, sil                        (bin_mdef.h:119-146)
 *   tmat_t       tp                                          (tmat.h:71-77)
 *   dict_t       pronunciations, base ids, filler range      (dict.h:66-93)
 *   dict2pid_t   rssid (right contexts), ldiph_lc            (dict2pid.h:130-158)
 *   ngram_search_t  root_chan[], the channel tree (->next / ->alt), homophone_set, single_phone_wid,
 *                   rhmm_1ph, the beams                      (ngram_search.h:62-108, 188-338)
 *   ngram_model_t   the current LM of ngs->lmset (ngram_model_set_lookup, sphinxbase/ngram_model.h:618;
 *                   _current_wid :687): its lm3g arrays (sphinxbase/src/libsphinxbase/lm/lm3g_model.h:131-148) --
 *                   the public iterators (ngram_model_mgrams) lose trigrams on some DMP files
 *                   ("Trigram %d has no valid bigram parent", lm3g_templates.c:526), the score functions do not
 * The including file defines PSAMD_DESC_T (the struct type to fill) before including this header.
 */
#ifndef PSAMD_EXPORT_H
#define PSAMD_EXPORT_H

#include <string.h>
#include <sphinxbase/ckd_alloc.h>
#include <sphinxbase/err.h>
#include <sphinxbase/ngram_model.h>
#include "ngram_model_internal.h"       /* sphinxbase/src/libsphinxbase/lm: ngram_model_t's layout */
#include "lm3g_model.h"                 /* ... and lm3g_model_t's */
#include "pocketsphinx_internal.h"
#include "ngram_search.h"
#include "phone_loop_search.h"

#ifndef PSAMD_DESC_T
#error "define PSAMD_DESC_T before including psamd_export.h"
#endif

typedef struct {
    void *ptr[96];
    int n;
} psamd_pool_t;

static void *
psamd_alloc(psamd_pool_t *pool, size_t n, size_t sz)
{
    void *p = ckd_calloc(n ? n : 1, sz);
    if (pool->n >= 96) E_FATAL("psamd_export: pool exhausted\n");
    pool->ptr[pool->n++] = p;
    return p;
}

static void
psamd_pool_free(psamd_pool_t *pool)
{
    int i;
    for (i = 0; i < pool->n; i++) ckd_free(pool->ptr[i]);
    pool->n = 0;
}

/* number the interior channels in depth-first (->next before ->alt) order */
static int32
psamd_count_tree(chan_t *hmm)
{
    int32 n = 0;
    for (; hmm; hmm = hmm->alt) n += 1 + psamd_count_tree(hmm->next);
    return n;
}

typedef struct {
    chan_t **chan;          /* interior channel by number (minus n_root) */
    int32 n;
} psamd_numbering_t;

static void
psamd_number_tree(psamd_numbering_t *nb, chan_t *hmm)
{
    for (; hmm; hmm = hmm->alt) {
        nb->chan[nb->n++] = hmm;
        psamd_number_tree(nb, hmm->next);
    }
}

static int32
psamd_chan_index(psamd_numbering_t *nb, chan_t *hmm)
{
    int32 i;
    for (i = 0; i < nb->n; i++) if (nb->chan[i] == hmm) return i;
    E_FATAL("psamd_export: channel not numbered\n");
    return -1;
}

/* returns 0, or -1 with a message where the device search does not serve the configuration */
static int
psamd_export(ps_decoder_t *ps, ngram_search_t *ngs, PSAMD_DESC_T *d, psamd_pool_t *pool)
{
    bin_mdef_t *mdef = ps->acmod->mdef;
    dict_t *dict = ps_search_dict(ngs);
    dict2pid_t *d2p = ps_search_dict2pid(ngs);
    ngram_model_t *lm;
    int32 n_ci = bin_mdef_n_ciphone(mdef), n_emit = bin_mdef_n_emit_state(mdef);
    int32 n_words = ps_search_n_words(ngs), w, i, j, k, n_ch;
    psamd_numbering_t nb;

    memset(d, 0, sizeof(*d));
    pool->n = 0;
    if (!ngs->fwdtree) { E_ERROR("psamd_export: -fwdtree is off\n"); return -1; }
    if (n_emit != 3 && n_emit != 5) { E_ERROR("psamd_export: %d emitting states (3 or 5 served)\n", n_emit); return -1; }
    if ((lm = ngram_model_set_lookup(ngs->lmset, NULL)) == NULL) {
        E_ERROR("psamd_export: interpolated LM sets are not served\n");
        return -1;
    }

    /* ---- acoustic model ---- */
    d->n_ci = n_ci; d->sil_ci = bin_mdef_silphone(mdef); d->n_emit = n_emit;
    d->n_sen = bin_mdef_n_sen(mdef); d->n_sseq = bin_mdef_n_sseq(mdef); d->n_tmat = ps->acmod->tmat->n_tmat;
    {
        uint16 *sseq = psamd_alloc(pool, (size_t)d->n_sseq * n_emit, sizeof(uint16));
        uint8 *tp = psamd_alloc(pool, (size_t)d->n_tmat * n_emit * (n_emit + 1), 1);
        for (i = 0; i < d->n_sseq; i++)
            for (j = 0; j < n_emit; j++) sseq[i * n_emit + j] = mdef->sseq[i][j];
        for (i = 0; i < d->n_tmat; i++)
            for (j = 0; j < n_emit; j++)
                for (k = 0; k <= n_emit; k++)
                    tp[(i * n_emit + j) * (n_emit + 1) + k] = ps->acmod->tmat->tp[i][j][k];
        d->sseq = sseq; d->tp = tp;
    }

    /* ---- dictionary, right contexts ---- */
    d->n_words = n_words;
    d->start_wid = dict_startwid(dict); d->finish_wid = dict_finishwid(dict); d->silence_wid = dict_silwid(dict);
    {
        int32 *basewid = psamd_alloc(pool, n_words, sizeof(int32)), *lmwid = psamd_alloc(pool, n_words, sizeof(int32)), *lmcw = NULL;
        int16 *fci = psamd_alloc(pool, n_words, sizeof(int16)), *lci = psamd_alloc(pool, n_words, sizeof(int16));
        int16 *l2ci = psamd_alloc(pool, n_words, sizeof(int16)), *rctm = psamd_alloc(pool, n_words, sizeof(int16));
        uint8 *flags = psamd_alloc(pool, n_words, 1);
        int32 *rcoff = psamd_alloc(pool, n_words + 1, sizeof(int32)), *rcrow = psamd_alloc(pool, n_words, sizeof(int32));
        int32 *rowof = psamd_alloc(pool, (size_t)n_ci * n_ci, sizeof(int32));
        int32 n_rc = 0, n_rows = 0;
        uint16 *rcssid;
        int16 *cimap;
        for (i = 0; i < n_ci * n_ci; i++) rowof[i] = -1;
        for (w = 0; w < n_words; w++) {
            basewid[w] = dict_basewid(dict, w);
            lmwid[w] = ngram_model_set_current_wid(ngs->lmset, w);
            if (lmwid[w] != NGRAM_INVALID_WID && NGRAM_IS_CLASSWID(lmwid[w])) {
                /* a word of a class (ngram_ng_score, ngram_model.c:505-516): scored -- and looked up as history -- as the class's tag
                 * word, plus its in-class weight */
                ngram_class_t *cl = lm->classes[NGRAM_CLASSID(lmwid[w])];
                if (lmcw == NULL) lmcw = psamd_alloc(pool, n_words, sizeof(int32));
                lmcw[w] = ngram_class_prob(cl, lmwid[w]);
                lmwid[w] = cl->tag_wid;
            }
            fci[w] = dict_first_phone(dict, w);
            lci[w] = dict_last_phone(dict, w);
            flags[w] = (dict_is_single_phone(dict, w) ? 1 : 0) | (dict_filler_word(dict, w) ? 2 : 0)
                | (dict_real_word(dict, w) ? 4 : 0);
            rcoff[w] = n_rc;
            if (dict_is_single_phone(dict, w)) { l2ci[w] = -1; rcrow[w] = -1; rctm[w] = -1; continue; }
            l2ci[w] = dict_second_last_phone(dict, w);
            rctm[w] = bin_mdef_pid2tmatid(mdef, lci[w]);
            n_rc += dict2pid_rssid(d2p, lci[w], l2ci[w])->n_ssid;
            if (rowof[lci[w] * n_ci + l2ci[w]] < 0) rowof[lci[w] * n_ci + l2ci[w]] = n_rows++;
            rcrow[w] = rowof[lci[w] * n_ci + l2ci[w]];
        }
        rcoff[n_words] = n_rc;
        rcssid = psamd_alloc(pool, n_rc, sizeof(uint16));
        cimap = psamd_alloc(pool, (size_t)n_rows * n_ci, sizeof(int16));
        for (w = 0; w < n_words; w++) {
            xwdssid_t *rs;
            if (dict_is_single_phone(dict, w)) continue;
            rs = dict2pid_rssid(d2p, lci[w], l2ci[w]);
            for (i = 0; i < rs->n_ssid; i++) rcssid[rcoff[w] + i] = rs->ssid[i];
            for (i = 0; i < n_ci; i++) cimap[rcrow[w] * n_ci + i] = rs->cimap[i];
        }
        d->w_lmcw = lmcw;
        d->w_basewid = basewid; d->w_lmwid = lmwid; d->w_first_ci = fci; d->w_last_ci = lci; d->w_last2_ci = l2ci;
        d->w_flags = flags; d->w_rc_off = rcoff; d->rc_ssid = rcssid; d->w_rc_row = rcrow; d->n_rc_rows = n_rows;
        d->rc_cimap = cimap; d->w_rc_tmat = rctm;
    }

    /* ---- the channel tree ---- */
    d->n_root = ngs->n_root_chan;
    nb.n = 0;
    for (i = 0, k = 0; i < ngs->n_root_chan; i++) k += psamd_count_tree(ngs->root_chan[i].next);
    nb.chan = psamd_alloc(pool, k, sizeof(chan_t *));
    for (i = 0; i < ngs->n_root_chan; i++) psamd_number_tree(&nb, ngs->root_chan[i].next);
    if (nb.n != ngs->n_nonroot_chan) E_FATAL("psamd_export: %d interior channels numbered, the search has %d\n", nb.n, ngs->n_nonroot_chan);
    d->n_nonroot = nb.n;
    n_ch = d->n_root + d->n_nonroot;
    {
        int16 *rci = psamd_alloc(pool, d->n_root, sizeof(int16)), *rci2 = psamd_alloc(pool, d->n_root, sizeof(int16));
        int16 *rtm = psamd_alloc(pool, d->n_root, sizeof(int16));
        uint16 *rss = psamd_alloc(pool, d->n_root, sizeof(uint16));
        uint16 *rlc = psamd_alloc(pool, (size_t)d->n_root * n_ci, sizeof(uint16));
        int32 *coff = psamd_alloc(pool, n_ch + 1, sizeof(int32)), *poff = psamd_alloc(pool, n_ch + 1, sizeof(int32));
        int32 *child = psamd_alloc(pool, d->n_nonroot, sizeof(int32)), *pen = psamd_alloc(pool, n_words, sizeof(int32));
        uint16 *nss = psamd_alloc(pool, d->n_nonroot, sizeof(uint16));
        int16 *ntm = psamd_alloc(pool, d->n_nonroot, sizeof(int16)), *nci = psamd_alloc(pool, d->n_nonroot, sizeof(int16));
        int32 nc = 0, np = 0, c;
        /* the interior numbering is depth-first, so a channel's children are NOT contiguous in it; the child lists
         * are written out explicitly in sibling order */
        for (c = 0; c < n_ch; c++) {
            chan_t *first;
            int32 pw;
            coff[c] = nc; poff[c] = np;
            if (c < d->n_root) {
                root_chan_t *r = &ngs->root_chan[c];
                rci[c] = r->ciphone; rci2[c] = r->ci2phone; rtm[c] = r->hmm.tmatid; rss[c] = hmm_mpx_ssid(&r->hmm, 0);
                for (j = 0; j < n_ci; j++) rlc[c * n_ci + j] = dict2pid_ldiph_lc(d2p, r->ciphone, r->ci2phone, j);
                first = r->next; pw = r->penult_phn_wid;
            }
            else {
                chan_t *h = nb.chan[c - d->n_root];
                nss[c - d->n_root] = hmm_nonmpx_ssid(&h->hmm); ntm[c - d->n_root] = h->hmm.tmatid;
                nci[c - d->n_root] = h->ciphone;
                first = h->next; pw = h->info.penult_phn_wid;
            }
            for (; first; first = first->alt) child[nc++] = d->n_root + psamd_chan_index(&nb, first);
            for (; pw >= 0; pw = ngs->homophone_set[pw]) pen[np++] = pw;
        }
        coff[n_ch] = nc; poff[n_ch] = np;
        d->root_ci = rci; d->root_ci2 = rci2; d->root_tmat = rtm; d->root_ssid0 = rss; d->root_lc_ssid = rlc;
        d->ch_child_off = coff; d->ch_child = child; d->ch_pen_off = poff; d->ch_pen_wid = pen;
        d->nr_ssid = nss; d->nr_tmat = ntm; d->nr_ci = nci;
    }

    /* ---- single-phone words ---- */
    d->n_1ph = ngs->n_1ph_words; d->n_1ph_lm = ngs->n_1ph_LMwords;
    {
        int32 *spw = psamd_alloc(pool, d->n_1ph, sizeof(int32)), *fill = psamd_alloc(pool, d->n_1ph, sizeof(int32));
        uint16 *sps = psamd_alloc(pool, d->n_1ph, sizeof(uint16)), *splc = psamd_alloc(pool, (size_t)d->n_1ph * n_ci, sizeof(uint16));
        int16 *sptm = psamd_alloc(pool, d->n_1ph, sizeof(int16)), *spci = psamd_alloc(pool, d->n_1ph, sizeof(int16));
        int32 nf = 0;
        for (i = 0; i < d->n_1ph; i++) {
            root_chan_t *r;
            w = ngs->single_phone_wid[i];
            r = (root_chan_t *)ngs->word_chan[w];
            spw[i] = w; sptm[i] = r->hmm.tmatid; spci[i] = r->ciphone;
            /* ngram_fwdtree_start clears these channels but not their multiplexed ids; a fresh channel has the
             * CI phone's (init_search_tree) */
            sps[i] = bin_mdef_pid2ssid(mdef, r->ciphone);
            for (j = 0; j < n_ci; j++) splc[i * n_ci + j] = dict2pid_ldiph_lc(d2p, r->ciphone, r->ci2phone, j);
        }
        /* word_transition's last loop: the filler range minus <sil>, <s> and words without a channel */
        for (w = dict_filler_start(dict); w <= dict_filler_end(dict); w++) {
            if (w == ps_search_silence_wid(ngs) || w == dict_startwid(dict) || ngs->word_chan[w] == NULL) continue;
            for (i = 0; i < d->n_1ph && spw[i] != w; i++) ;
            if (i == d->n_1ph) { E_ERROR("psamd_export: filler word %d has a channel but is not a listed single-phone word\n", w); return -1; }
            fill[nf++] = i;
        }
        d->sp_wid = spw; d->sp_ssid0 = sps; d->sp_lc_ssid = splc; d->sp_tmat = sptm; d->sp_ci = spci;
        d->n_fill = nf; d->fill_sp = fill;
    }

    /* ---- the language model: the lm3g arrays lm3g_tg_score / lm3g_bg_score search (lm3g_templates.c:73-195), with the
     * 16-bit indirections (prob2 / bo_wt2 / prob3 tables, segmented trigram offsets: lm3g_model.h:101-113) resolved.
     * Both in-memory kinds (read from a DMP file, read from an ARPA file) are {ngram_model_t base; lm3g_model_t lm3g; ..};
     * they differ in the width of the word id inside bigram_t / trigram_t (ngram_model_dmp.h:52-68,
     * ngram_model_arpa.h:51-68).  ngram_model_dmp_build returns its argument for a DMP model. ---- */
    {
        typedef struct { uint16 wid, prob2, bo_wt2, trigrams; } bg16_t;
        typedef struct { uint16 wid, prob3; } tg16_t;
        typedef struct { uint32 wid; uint16 prob2, bo_wt2, trigrams; } bg32_t;
        typedef struct { uint32 wid; uint16 prob3; } tg32_t;
        extern ngram_model_t *ngram_model_dmp_build(ngram_model_t *base);   /* sphinxbase ngram_model_dmp.h:90 */
        ngram_model_t *as_dmp = ngram_model_dmp_build(lm);
        const int is_dmp = (as_dmp == lm);
        lm3g_model_t *g = (lm3g_model_t *)((char *)lm + sizeof(ngram_model_t));
        int32 const *cnt = ngram_model_get_counts(lm);
        int32 order = ngram_model_get_size(lm), n_ug = cnt[0], n_bg = order > 1 ? cnt[1] : 0, n_tg = order > 2 ? cnt[2] : 0;
        int32 *ugp = psamd_alloc(pool, n_ug, sizeof(int32)), *ugb = psamd_alloc(pool, n_ug, sizeof(int32));
        int32 *ugf = psamd_alloc(pool, n_ug + 1, sizeof(int32));
        int32 *bgw = psamd_alloc(pool, n_bg, sizeof(int32)), *bgp = psamd_alloc(pool, n_bg, sizeof(int32));
        int32 *bgb = psamd_alloc(pool, n_bg, sizeof(int32)), *bgf = psamd_alloc(pool, n_bg + 1, sizeof(int32));
        int32 *tgw = psamd_alloc(pool, n_tg, sizeof(int32)), *tgp = psamd_alloc(pool, n_tg, sizeof(int32));
        int32 u, bi, t;
        ngram_model_free(as_dmp);
        for (u = 0; u < n_ug; u++) {
            ugp[u] = g->unigrams[u].prob1.l; ugb[u] = g->unigrams[u].bo_wt1.l;
            ugf[u] = n_bg > 0 ? g->unigrams[u].bigrams : 0;
        }
        ugf[n_ug] = n_bg > 0 ? g->unigrams[n_ug].bigrams : 0;
        for (bi = 0; bi <= n_bg && n_bg > 0; bi++) {
            int32 wid, p2, b2, tr;
            if (is_dmp) { bg16_t *x = (bg16_t *)g->bigrams + bi; wid = x->wid; p2 = x->prob2; b2 = x->bo_wt2; tr = x->trigrams; }
            else { bg32_t *x = (bg32_t *)g->bigrams + bi; wid = x->wid; p2 = x->prob2; b2 = x->bo_wt2; tr = x->trigrams; }
            bgf[bi] = n_tg > 0 ? g->tseg_base[bi >> LOG_BG_SEG_SZ] + tr : 0;
            if (bi == n_bg) break;
            bgw[bi] = wid; bgp[bi] = g->prob2[p2].l; bgb[bi] = (n_tg > 0 && g->bo_wt2) ? g->bo_wt2[b2].l : 0;
        }
        if (n_bg == 0) bgf[0] = 0;
        for (t = 0; t < n_tg; t++) {
            if (is_dmp) { tg16_t *x = (tg16_t *)g->trigrams + t; tgw[t] = x->wid; tgp[t] = g->prob3[x->prob3].l; }
            else { tg32_t *x = (tg32_t *)g->trigrams + t; tgw[t] = x->wid; tgp[t] = g->prob3[x->prob3].l; }
        }
        d->lm_order = order; d->lm_n_ug = n_ug; d->lm_n_bg = n_bg; d->lm_n_tg = n_tg; d->lm_zero = ngram_zero(lm);
        d->ug_prob = ugp; d->ug_bowt = ugb; d->ug_firstbg = ugf;
        d->bg_wid = bgw; d->bg_prob = bgp; d->bg_bowt = bgb; d->bg_firsttg = bgf;
        d->tg_wid = tgw; d->tg_prob = tgp;
    }

    d->beam = ngs->beam; d->pbeam = ngs->pbeam; d->wbeam = ngs->wbeam; d->lpbeam = ngs->lpbeam;
    d->lponlybeam = ngs->lponlybeam; d->fillpen = ngs->fillpen; d->silpen = ngs->silpen; d->nwpen = ngs->nwpen;
    d->pip = ngs->pip; d->maxwpf = ngs->maxwpf; d->maxhmmpf = ngs->maxhmmpf;

    /* ---- phone loop look-ahead (-pl_window; phone_loop_search.c:69-108): the loop's HMMs and beams as the decoder holds them ---- */
    if (ps_search_lookahead(ngs)) {
        phone_loop_search_t *pls = (phone_loop_search_t *)ps_search_lookahead(ngs);
        uint16 *cs = psamd_alloc(pool, n_ci, sizeof(uint16));
        int16 *ct = psamd_alloc(pool, n_ci, sizeof(int16));
        for (i = 0; i < n_ci; i++) { cs[i] = bin_mdef_pid2ssid(mdef, i); ct[i] = bin_mdef_pid2tmatid(mdef, i); }
        d->pl_window = ps->pl_window; d->pl_beam = pls->beam; d->pl_pbeam = pls->pbeam; d->pl_pip = pls->pip;
        d->ci_ssid = cs; d->ci_tmat = ct;
    }
    return 0;
}

#endif /* PSAMD_EXPORT_H */
