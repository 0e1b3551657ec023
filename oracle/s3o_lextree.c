/*
 * oracle/s3o_lextree.c -- CPU ORACLE (test infrastructure only; see s3o.h).
 *
 * The per-frame operations of sphinx3's lexical-tree search on a FLATTENED
 * lextree (node arrays + CSR child lists instead of lextree_node_t / glist),
 * restating, sequentially and in the reference's order,
 *   sphinx3/src/libs3decoder/libsearch/lextree.c:1093-1236  lextree_enter
 *   lextree.c:1240-1249  lextree_active_swap
 *   lextree.c:1253-1310  lextree_hmm_eval
 *   lextree.c:1314-1358  lextree_hmm_histbin
 *   lextree.c:1365-1597  lextree_hmm_propagate_non_leaves  (composite-triphone mode,
 *                        -pheurtype 0: the only mode kbcore.c:626 allows)
 *   lextree.c:1600-1663  lextree_hmm_propagate_leaves
 *   lextree.c:910-932    lextree_ssid_active
 *   lextree.c:936-961    lextree_utt_end
 *   libam/mdef.c:857-869 mdef_sseq2sen_active, libsearch/dict2pid.c:1055-1075
 *                        dict2pid_comsseq2sen_active
 *
 * The order of the active list is part of the semantics (it decides which
 * parent wins an exact tie, whether a child is cleared before or after its
 * parent enters it, and the order in which word exits reach vithist), so this
 * restatement walks the lists exactly as the reference does.
 */
#include <stdlib.h>
#include <string.h>
#include "s3o.h"

s3o_lextree_t *
s3o_lextree_init(int32_t n_node, const int32_t *ssid, const int32_t *tmatid,
                 const uint8_t *composite, const int32_t *wid, const int32_t *prob,
                 const int32_t *child_off, const int32_t *child,
                 int32_t n_lc, const int16_t *lc, const int32_t *lcroot_off, const int32_t *lcroot,
                 int32_t n_root, const int32_t *root,
                 int32_t n_emit, const int32_t *tp, const int16_t *sseq, const int16_t *comsseq)
{
    s3o_lextree_t *lt = (s3o_lextree_t *)calloc(1, sizeof *lt);
    int32_t i;
    lt->n_node = n_node;
    lt->ssid = ssid; lt->tmatid = tmatid; lt->composite = composite; lt->wid = wid; lt->prob = prob;
    lt->child_off = child_off; lt->child = child;
    lt->n_lc = n_lc; lt->lc = lc; lt->lcroot_off = lcroot_off; lt->lcroot = lcroot;
    lt->n_root = n_root; lt->root = root;
    lt->ctx.n_emit_state = n_emit; lt->ctx.tp = tp; lt->ctx.sseq = sseq; lt->ctx.senscore = NULL;
    lt->comctx = lt->ctx;
    lt->comctx.sseq = comsseq;
    lt->hmm = (s3o_hmm_t *)calloc(n_node, sizeof(s3o_hmm_t));
    lt->active = (int32_t *)calloc(n_node, sizeof(int32_t));
    lt->next_active = (int32_t *)calloc(n_node, sizeof(int32_t));
    for (i = 0; i < n_node; i++)        /* lextree_node_alloc -> hmm_init (non-mpx) */
        s3o_hmm_init(composite[i] ? &lt->comctx : &lt->ctx, &lt->hmm[i], 0, ssid[i], tmatid[i]);
    return lt;
}

void
s3o_lextree_free(s3o_lextree_t *lt)
{
    if (!lt) return;
    free(lt->hmm); free(lt->active); free(lt->next_active);
    free(lt);
}

void
s3o_lextree_enter(s3o_lextree_t *lt, int32_t lc, int32_t cf, int32_t inscore, int32_t inhist,
                  int32_t thresh)
{
    const int32_t *list;
    int32_t n_list, i, n, nf = cf + 1;
    if (lt->n_lc == 0) {
        list = lt->root;
        n_list = lt->n_root;
    }
    else {
        for (i = 0; i < lt->n_lc && lt->lc[i] != lc; i++);
        if (i >= lt->n_lc) abort();         /* assert(i < lextree->n_lc) */
        list = lt->lcroot + lt->lcroot_off[i];
        n_list = lt->lcroot_off[i + 1] - lt->lcroot_off[i];
    }
    n = lt->n_next_active;
    for (i = 0; i < n_list; i++) {
        int32_t ln = list[i];
        s3o_hmm_t *h = &lt->hmm[ln];
        int32_t scr = (int32_t)((uint32_t)inscore + (uint32_t)lt->prob[ln]);
        if (scr >= thresh && h->score[0] < scr) {
            h->score[0] = scr;
            h->history[0] = inhist;
            if (h->frame != nf) {
                h->frame = nf;
                lt->next_active[n++] = ln;
            }
        }
    }
    lt->n_next_active = n;
}

void
s3o_lextree_active_swap(s3o_lextree_t *lt)
{
    int32_t *t = lt->active;
    lt->active = lt->next_active;
    lt->next_active = t;
    lt->n_active = lt->n_next_active;
    lt->n_next_active = 0;
}

int32_t
s3o_lextree_hmm_eval(s3o_lextree_t *lt, const int32_t *senscr, const int32_t *comsen, int32_t frm)
{
    int32_t best = S3O_MAX_NEG_INT32, wbest = S3O_MAX_NEG_INT32, i, k;
    lt->ctx.senscore = senscr;
    lt->comctx.senscore = comsen;
    for (i = 0; i < lt->n_active; i++) {
        int32_t ln = lt->active[i];
        if (lt->hmm[ln].frame != frm) abort();     /* assert(hmm_frame(ln) == frm) */
        k = s3o_hmm_vit_eval(lt->composite[ln] ? &lt->comctx : &lt->ctx, &lt->hmm[ln]);
        if (best < k) best = k;
        if (lt->wid[ln] >= 0 && wbest < k) wbest = k;
    }
    lt->best = best;
    lt->wbest = wbest;
    return best;
}

void
s3o_lextree_hmm_histbin(s3o_lextree_t *lt, int32_t bestscr, int32_t *bin, int32_t nbin, int32_t bw)
{
    /* glist_add_ptr prepends: within a bin the nodes come out in REVERSE insertion order */
    int32_t *head = (int32_t *)malloc(sizeof(int32_t) * nbin);
    int32_t *next = (int32_t *)malloc(sizeof(int32_t) * (lt->n_active > 0 ? lt->n_active : 1));
    int32_t *nodes = (int32_t *)malloc(sizeof(int32_t) * (lt->n_active > 0 ? lt->n_active : 1));
    int32_t i, k, j;
    for (i = 0; i < nbin; i++) head[i] = -1;
    for (i = 0; i < lt->n_active; i++) {
        int32_t ln = lt->active[i];
        k = (bestscr - lt->hmm[ln].bestscore) / bw;
        if (k >= nbin) k = nbin - 1;
        bin[k]++;
        nodes[i] = ln;
        next[i] = head[k];
        head[k] = i;
    }
    k = 0;
    for (i = 0; i < nbin; i++)
        for (j = head[i]; j >= 0; j = next[j])
            lt->active[k++] = nodes[j];
    free(head); free(next); free(nodes);
}

void
s3o_lextree_hmm_propagate_non_leaves(s3o_lextree_t *lt, int32_t cf, int32_t th, int32_t pth,
                                     int32_t wth)
{
    int32_t nf = cf + 1, i, j, n = lt->n_next_active;
    (void)wth;
    for (i = 0; i < lt->n_active; i++) {
        int32_t ln = lt->active[i];
        s3o_hmm_t *h = &lt->hmm[ln];
        if (h->frame < nf) {
            if (h->bestscore >= th) {           /* active in next frame */
                h->frame = nf;
                lt->next_active[n++] = ln;
            }
            else
                s3o_hmm_clear(&lt->ctx, h);     /* deactivate */
        }
        if (lt->wid[ln] < 0) {                  /* not a leaf */
            if (h->out_score < pth)
                continue;
            for (j = lt->child_off[ln]; j < lt->child_off[ln + 1]; j++) {
                int32_t ln2 = lt->child[j];
                s3o_hmm_t *h2 = &lt->hmm[ln2];
                int32_t newscore = (int32_t)((uint32_t)h->out_score
                                             + ((uint32_t)lt->prob[ln2] - (uint32_t)lt->prob[ln]));
                if (newscore >= th && h2->score[0] < newscore) {
                    h2->score[0] = newscore;
                    h2->history[0] = h->out_history;
                    if (h2->frame != nf) {
                        h2->frame = nf;
                        lt->next_active[n++] = ln2;
                    }
                }
            }
        }
    }
    lt->n_next_active = n;
}

int32_t
s3o_lextree_hmm_propagate_leaves(const s3o_lextree_t *lt, int32_t wth, int32_t *out_wid,
                                 int32_t *out_score, int32_t *out_hist, int32_t max_out)
{
    int32_t i, n = 0;
    for (i = 0; i < lt->n_active; i++) {
        int32_t ln = lt->active[i];
        const s3o_hmm_t *h = &lt->hmm[ln];
        if (lt->wid[ln] < 0 || h->out_score < wth)
            continue;
        if (h->out_history == -1)
            return -1;                          /* LEXTREE_OPERATION_FAILURE */
        if (n < max_out) {
            out_wid[n] = lt->wid[ln];
            out_score[n] = (int32_t)((uint32_t)h->out_score - (uint32_t)lt->prob[ln]);
            out_hist[n] = (int32_t)h->out_history;
        }
        n++;
    }
    return n;
}

void
s3o_lextree_ssid_active(const s3o_lextree_t *lt, uint8_t *ssid, uint8_t *comssid)
{
    int32_t i;
    for (i = 0; i < lt->n_active; i++) {
        int32_t ln = lt->active[i];
        if (lt->composite[ln]) comssid[lt->ssid[ln]] = 1;
        else ssid[lt->ssid[ln]] = 1;
    }
}

void
s3o_sseq2sen_active(const int16_t *sseq, int32_t n_sseq, int32_t n_emit, const uint8_t *ssid,
                    uint8_t *sen)
{
    int32_t ss, i;
    for (ss = 0; ss < n_sseq; ss++)
        if (ssid[ss])
            for (i = 0; i < n_emit; i++)
                sen[sseq[ss * n_emit + i]] = 1;
}

void
s3o_comsseq2sen_active(const int16_t *comsseq, int32_t n_comsseq, int32_t n_emit,
                       const int32_t *comstate_off, const int16_t *comstate,
                       const uint8_t *comssid, uint8_t *sen)
{
    int32_t ss, i, j;
    for (ss = 0; ss < n_comsseq; ss++)
        if (comssid[ss])
            for (i = 0; i < n_emit; i++) {
                int32_t cs = comsseq[ss * n_emit + i];
                for (j = comstate_off[cs]; j < comstate_off[cs + 1]; j++)
                    sen[comstate[j]] = 1;
            }
}

void
s3o_lextree_utt_end(s3o_lextree_t *lt)
{
    int32_t i;
    for (i = 0; i < lt->n_active; i++)
        s3o_hmm_clear(&lt->ctx, &lt->hmm[lt->active[i]]);
    lt->n_active = 0;
    lt->n_next_active = 0;
}
