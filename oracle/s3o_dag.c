/*
 * s3o_dag.c -- TEST INFRASTRUCTURE (never linked into the product): a plain-C restatement of sphinx3's SECOND PASS
 * over the first pass's Viterbi history:
 *
 *   vithist_dag_build          libsearch/vithist.c:1100-1311   word lattice (DAG) from the history table
 *   dag_link / dag_update_link libsearch/dag.c:186-300         link lists (head insertion: the ORDER is semantics, ties)
 *   dag_bypass_filler_nodes    libsearch/dag.c:1037-1075       transitive links around filler words
 *   dag_search / dag_bestpath  libsearch/dag.c:397-484, 893-965  best path under the trigram, link by link
 *   dag_backtrace              libsearch/dag.c:590-671         the hypothesis, bypassed fillers restored
 *   srch_TST_bestpath_impl     libsearch/srch_time_switch_tree.c:1391-1440   the driver (filler coercion of the end node,
 *                              linksilences: <s> / </s> get their LM ids back for the search)
 *
 * The reference keeps nodes and links in singly linked lists built by HEAD insertion and walks them in list order; which
 * of two equally good predecessors wins (strict >) is therefore decided by creation order.  The restatement keeps the
 * same lists (arrays of `next` indices) so that the walks visit in the same order.
 *
 * Pinned: tests/test_oracle_dag.py -- oracle/_ref/ref_s3odag_decode (the unmodified reference decoder whose gen_dag /
 * bestpath_impl slots call THIS file on the reference's own vithist_t) writes -hyp / -hypseg byte-identical to the
 * unmodified reference with -bestpath 1 on tidigits and RM1, and the node / link counts equal the reference's dag_t.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "s3o.h"

typedef struct {
    int32_t wid, sf, fef, lef, seqid, reachable;
    int32_t hook;               /* head of the node's entry list (hk_*), -1 */
    int32_t succ, pred;         /* heads of the link lists, -1 */
    int32_t alloc_next;         /* dag->list chain */
    int32_t frame_next;         /* sfwid[sf] chain */
} dnode_t;

typedef struct {
    int32_t node;               /* the far end (succlist: destination; predlist: predecessor), -1: none (root's stop link) */
    int32_t src;
    int32_t ascr, lscr, pscr, ef;
    int32_t next, history, bypass;
    int32_t pscr_valid;
} dlink_t;

typedef struct {
    dnode_t *nd; int32_t n_node, cap_node;
    dlink_t *lk; int32_t n_lk, cap_lk;
    int32_t *hk_ent, *hk_next; int32_t n_hk, cap_hk;
    int32_t nlink, nbypass, maxedge, lmop, maxlmop;
    int32_t list, root, end;
    dlink_t final;
} dag_t;

static int32_t
new_link(dag_t *d)
{
    if (d->n_lk == d->cap_lk) {
        d->cap_lk = d->cap_lk ? 2 * d->cap_lk : 4096;
        d->lk = realloc(d->lk, sizeof(dlink_t) * (size_t)d->cap_lk);
    }
    return d->n_lk++;
}

/* dag_link, dag.c:186-238 */
static int32_t
dag_link(dag_t *d, int32_t pd, int32_t dn, int32_t ascr, int32_t lscr, int32_t ef, int32_t byp)
{
    int32_t l;
    if (ascr > 0) return 0;             /* "silently refuse to create positive edges" */
    if (pd >= 0) {
        l = new_link(d);
        d->lk[l].node = dn; d->lk[l].src = pd; d->lk[l].ascr = ascr; d->lk[l].lscr = lscr;
        d->lk[l].pscr = (int32_t)0x80000000; d->lk[l].pscr_valid = 0; d->lk[l].history = -1; d->lk[l].ef = ef;
        d->lk[l].next = d->nd[pd].succ; d->lk[l].bypass = byp;
        d->nd[pd].succ = l;
    }
    l = new_link(d);
    d->lk[l].node = pd; d->lk[l].src = dn; d->lk[l].ascr = ascr; d->lk[l].lscr = lscr;
    d->lk[l].pscr = (int32_t)0x80000000; d->lk[l].pscr_valid = 0; d->lk[l].history = -1; d->lk[l].ef = ef;
    d->lk[l].bypass = byp;
    d->lk[l].next = d->nd[dn].pred;
    d->nd[dn].pred = l;
    if (byp >= 0) d->nbypass++;
    d->nlink++;
    return d->nlink > d->maxedge ? -1 : 0;
}

static int32_t
find_link(const dag_t *d, int32_t head, int32_t dst, int32_t bypass)      /* find_succlink / find_predlink, dag.c:240-270 */
{
    int32_t l;
    for (l = head; l >= 0; l = d->lk[l].next)
        if (d->lk[l].node == dst) {
            if (bypass && d->lk[l].bypass < 0) continue;
            break;
        }
    return l;
}

/* dag_update_link, dag.c:277-300 */
static int32_t
dag_update_link(dag_t *d, int32_t pd, int32_t dn, int32_t ascr, int32_t ef, int32_t byp)
{
    const int32_t l = find_link(d, d->nd[pd].succ, dn, byp >= 0);
    if (l < 0) return dag_link(d, pd, dn, ascr, 0, ef, byp);
    if (d->lk[l].ascr < ascr) {
        const int32_t r = find_link(d, d->nd[dn].pred, pd, byp >= 0);
        d->lk[l].ascr = d->lk[r].ascr = ascr;
        d->lk[l].ef = d->lk[r].ef = ef;
        d->lk[l].bypass = d->lk[r].bypass = byp;
    }
    return 0;
}

static int32_t
lmid(const s3o_dagcfg_t *c, int32_t w)
{
    const int32_t b = c->basewid[w];
    if (b == c->startwid) return c->start_lwid;         /* linksilences, kbcore.c:191-206 */
    if (b == c->finishwid) return c->finish_lwid;
    return c->lwid[b];
}

/* dag_bestpath, dag.c:397-484: l is a BACKWARD link out of src */
static int32_t
dag_bestpath(dag_t *d, int32_t l, int32_t src, const s3o_dagcfg_t *c, const s3o_lm3g_t *lm)
{
    const int32_t dn = d->lk[l].node;
    int32_t pl;
    if (dn < 0) {                       /* no destination: src is the root */
        d->lk[l].lscr = 0; d->lk[l].pscr = 0; d->lk[l].pscr_valid = 1; d->lk[l].history = -1;
        return 0;
    }
    for (pl = d->nd[dn].pred; pl >= 0; pl = d->lk[pl].next) {
        const int32_t pd = d->lk[pl].node;
        int32_t score, lscr;
        if (pd >= 0 && c->is_filler[d->nd[pd].wid]) continue;
        if (!d->lk[pl].pscr_valid && dag_bestpath(d, pl, dn, c, lm) < 0) return -1;
        score = (int32_t)((uint32_t)d->lk[pl].pscr + (uint32_t)d->lk[l].ascr);
        if (score > d->lk[l].pscr) {
            if (pd >= 0)
                lscr = (int32_t)(c->lwf * s3o_lm_tg_score(lm, lmid(c, d->nd[pd].wid), lmid(c, d->nd[dn].wid), lmid(c, d->nd[src].wid),
                                                          c->basewid[d->nd[src].wid]));
            else
                lscr = (int32_t)(c->lwf * s3o_lm_bg_score(lm, lmid(c, d->nd[dn].wid), lmid(c, d->nd[src].wid), c->basewid[d->nd[src].wid]));
            score = (int32_t)((uint32_t)score + (uint32_t)lscr);
            if (d->lmop++ >= d->maxlmop) return -1;
            if (score > d->lk[l].pscr) { d->lk[l].lscr = lscr; d->lk[l].pscr = score; d->lk[l].history = pl; }
        }
    }
    d->lk[l].pscr_valid = 1;
    return 0;
}

int32_t
s3o_dag_bestpath(const s3o_dagcfg_t *c, const s3o_lm3g_t *lm, int32_t n_entry, const int32_t *wid, const int32_t *sf,
                 const int32_t *ef, const int32_t *ascr, const int32_t *lscr, const int32_t *score, const uint8_t *valid,
                 int32_t n_frm, int32_t endid, int32_t n_hyp, const int32_t *hyp_wid, const int32_t *hyp_sf, int32_t *out,
                 int32_t max_out, int32_t *stats)
{
    dag_t D;
    int32_t *sfw, i, f, k, rc = -1, n_out = 0;
    memset(&D, 0, sizeof D);
    D.maxedge = c->maxedge; D.list = D.root = D.end = -1;
    D.maxlmop = c->maxlmop;
    if (c->maxlpf > 0 && (int64_t)c->maxlpf * n_frm < D.maxlmop) D.maxlmop = c->maxlpf * n_frm;  /* vithist.c:1300-1305 (k *= nfrm) */
    D.cap_node = n_entry + 4; D.nd = calloc((size_t)D.cap_node, sizeof(dnode_t));
    D.cap_hk = n_entry + 4; D.hk_ent = malloc(sizeof(int32_t) * (size_t)D.cap_hk); D.hk_next = malloc(sizeof(int32_t) * (size_t)D.cap_hk);
    sfw = malloc(sizeof(int32_t) * (size_t)(n_frm + 2));
    for (f = 0; f <= n_frm; f++) sfw[f] = -1;

    /* ---- vithist_dag_build, vithist.c:1100-1311 ---- */
    for (i = 0; i < n_entry; i++) {
        int32_t s, e, dn, g;
        if (valid && !valid[i]) continue;
        if (sf[i] == -1) s = e = 0;                             /* the dummy <s> entry: "MAJOR HACK" */
        else if (sf[i] == 0) { s = 1; e = ef[i]; }
        else { s = sf[i]; e = ef[i]; }
        if (s < 0 || s > n_frm) goto done;
        for (dn = sfw[s]; dn >= 0; dn = D.nd[dn].frame_next)
            if (D.nd[dn].wid == wid[i]) break;
        if (dn < 0) {
            dn = D.n_node++;
            D.nd[dn].wid = wid[i]; D.nd[dn].sf = s; D.nd[dn].fef = e; D.nd[dn].lef = e; D.nd[dn].seqid = -1;
            D.nd[dn].hook = -1; D.nd[dn].succ = D.nd[dn].pred = -1; D.nd[dn].alloc_next = -1;
            D.nd[dn].frame_next = sfw[s];                       /* glist_add_ptr: head insertion */
            sfw[s] = dn;
        }
        else D.nd[dn].lef = e;
        if (i == endid) D.end = dn;
        for (g = D.nd[dn].hook; g >= 0; g = D.hk_next[g])
            if (ef[D.hk_ent[g]] == ef[i]) break;
        if (g >= 0) { if (score[i] > score[D.hk_ent[g]]) D.hk_ent[g] = i; }
        else { g = D.n_hk++; D.hk_ent[g] = i; D.hk_next[g] = D.nd[dn].hook; D.nd[dn].hook = g; }
    }
    for (k = 0; k < n_hyp; k++) {                               /* keep the first pass's own words */
        const int32_t hs = hyp_sf[k] == 0 ? 1 : hyp_sf[k];
        int32_t dn;
        if (hs < 0 || hs > n_frm) continue;
        for (dn = sfw[hs]; dn >= 0; dn = D.nd[dn].frame_next)
            if (D.nd[dn].wid == hyp_wid[k]) D.nd[dn].seqid = 0;
    }
    if (sfw[0] < 0 || D.nd[sfw[0]].wid != c->startwid || sfw[n_frm] < 0 || D.nd[sfw[n_frm]].wid != c->finishwid) goto done;
    D.nd[sfw[0]].seqid = 0; D.root = sfw[0];
    D.nd[sfw[n_frm]].seqid = 0;
    if (D.end < 0) D.end = sfw[n_frm];
    D.nd[D.end].seqid = 0;
    memset(&D.final, 0, sizeof D.final);
    D.final.node = D.end; D.final.next = -1; D.final.bypass = -1; D.final.history = -1;
    {
        int32_t g;
        for (g = D.nd[D.end].hook; g >= 0; g = D.hk_next[g])
            if (ef[D.hk_ent[g]] == n_frm) D.final.ascr = ascr[D.hk_ent[g]];
    }
    i = 0;
    for (f = 0; f <= n_frm; f++) {
        int32_t dn;
        for (dn = sfw[f]; dn >= 0; dn = D.nd[dn].frame_next) {
            if (D.nd[dn].lef - D.nd[dn].fef > c->min_endfr || D.nd[dn].seqid >= 0) {
                D.nd[dn].seqid = i++;
                D.nd[dn].alloc_next = D.list;
                D.list = dn;
            }
            else D.nd[dn].seqid = -1;
        }
    }
    for (f = 0; f < n_frm; f++) {
        int32_t dn;
        for (dn = sfw[f]; dn >= 0; dn = D.nd[dn].frame_next) {
            int32_t g;
            if (D.nd[dn].seqid < 0) continue;
            for (g = D.nd[dn].hook; g >= 0; g = D.hk_next[g]) {
                const int32_t ve = D.hk_ent[g], s2 = ef[ve] < 0 ? 1 : ef[ve] + 1;
                int32_t dn2;
                if (s2 > n_frm) continue;
                for (dn2 = sfw[s2]; dn2 >= 0; dn2 = D.nd[dn2].frame_next)
                    if (D.nd[dn2].seqid >= 0) (void)dag_link(&D, dn, dn2, ascr[ve], lscr[ve], s2 - 1, -1);
            }
        }
    }
    if (stats) { stats[0] = i; stats[1] = D.nlink; }

    /* ---- srch_TST_bestpath_impl, srch_time_switch_tree.c:1391-1440 ---- */
    if (c->is_filler[D.nd[D.end].wid]) D.nd[D.end].wid = c->finishwid;
    {   /* dag_bypass_filler_nodes, dag.c:1037-1075 */
        int32_t dn, stop = 0;
        for (dn = D.list; dn >= 0 && !stop; dn = D.nd[dn].alloc_next) {
            int32_t pl;
            if (!c->is_filler[D.nd[dn].wid]) continue;
            for (pl = D.nd[dn].pred; pl >= 0 && !stop; pl = D.lk[pl].next) {
                const int32_t pn = D.lk[pl].node;
                int32_t a = D.lk[pl].ascr, sl;
                a = (int32_t)((double)a + ((double)(c->fillpen[c->basewid[D.nd[dn].wid]] - c->wip) * c->lwf + (double)c->wip));
                for (sl = D.nd[dn].succ; sl >= 0; sl = D.lk[sl].next) {
                    const int32_t sn = D.lk[sl].node;
                    if (!c->is_filler[D.nd[sn].wid]
                        && dag_update_link(&D, pn, sn, (int32_t)((uint32_t)a + (uint32_t)D.lk[sl].ascr), D.lk[pl].ef, sl) < 0) { stop = 1; break; }
                }
            }
        }
    }
    if (stats) { stats[2] = D.nlink; stats[3] = D.nbypass; }
    {   /* dag_search, dag.c:893-965 */
        int32_t dn, l, bestl = -1, bestscore = (int32_t)0x80000000, fl, hist, n = 0, h0 = -1;
        for (dn = D.list; dn >= 0; dn = D.nd[dn].alloc_next)            /* dag_chk_linkscr */
            for (l = D.nd[dn].succ; l >= 0; l = D.lk[l].next)
                if (D.lk[l].ascr > 0) goto done;
        if (D.nd[D.end].pred < 0) goto done;
        if (D.nd[D.root].pred < 0) (void)dag_link(&D, -1, D.root, 0, 0, -1, -1);
        for (l = D.nd[D.end].pred; l >= 0; l = D.lk[l].next) {
            const int32_t pn = D.lk[l].node;
            if (pn < 0 || c->is_filler[D.nd[pn].wid]) continue;
            if (dag_bestpath(&D, l, D.end, c, lm) < 0) { bestl = -1; break; }
            if (D.lk[l].pscr > bestscore) { bestscore = D.lk[l].pscr; bestl = l; }
        }
        D.nd[D.root].pred = -1;
        if (stats) stats[4] = D.lmop;
        if (bestl < 0) goto done;
        /* the final node's own acoustic score; then dag_backtrace, dag.c:590-671.  The hypothesis is built back to front:
         * out rows are filled from the end of a scratch list and reversed */
        fl = new_link(&D);
        D.lk[fl] = D.final;
        D.lk[fl].history = bestl;
        D.lk[fl].pscr = (int32_t)((uint32_t)D.lk[bestl].pscr + (uint32_t)D.final.ascr);
        D.lk[fl].ef = n_frm - 1;
        {
            /* hyp as a linked list in scratch arrays: id sf ef ascr lscr next */
            const int32_t cap = 6 * (D.n_node + 8);
            int32_t *H = calloc((size_t)cap * 6, sizeof(int32_t)), nh = 0;
#define HN(x) H[6 * (x) + 5]
            for (l = fl; l >= 0; l = hist) {
                hist = D.lk[l].history;
                if (h0 >= 0) H[6 * h0 + 4] = D.lk[l].lscr;
                if (D.lk[l].node < 0) break;
                if (D.lk[l].bypass < 0) {
                    const int32_t h = nh++;
                    if (h >= cap) { free(H); goto done; }
                    H[6 * h] = D.nd[D.lk[l].node].wid; H[6 * h + 1] = D.nd[D.lk[l].node].sf; H[6 * h + 2] = D.lk[l].ef;
                    H[6 * h + 3] = D.lk[l].ascr; H[6 * h + 4] = 0; HN(h) = h0;
                    h0 = h;
                }
                else {
                    int32_t hh = -1, ht = -1, src = D.lk[l].node, dst = -1, bl, ll;
                    for (ll = l; ll >= 0; ll = D.lk[ll].bypass) {
                        const int32_t h = nh++;
                        if (h >= cap) { free(H); goto done; }
                        H[6 * h] = D.nd[src].wid; H[6 * h + 1] = D.nd[src].sf; H[6 * h + 4] = 0; HN(h) = -1;
                        if (hh >= 0) H[6 * h + 4] = (int32_t)(c->lwf * c->fillpen[c->basewid[D.nd[src].wid]]);
                        if (D.lk[ll].bypass >= 0) {
                            dst = D.lk[D.lk[ll].bypass].src;
                            bl = find_link(&D, D.nd[src].succ, dst, 0);
                            if (bl < 0) { free(H); goto done; }
                        }
                        else bl = ll;
                        H[6 * h + 2] = D.lk[bl].ef; H[6 * h + 3] = D.lk[bl].ascr;
                        if (ht >= 0) HN(ht) = h; else hh = h;
                        ht = h;
                        src = dst;
                    }
                    HN(ht) = h0;
                    h0 = hh;
                }
            }
            for (k = h0; k >= 0; k = HN(k)) n++;
            if (n > max_out) { free(H); rc = -2; goto done; }
            for (k = h0, n_out = 0; k >= 0; k = HN(k), n_out++) {
                out[n_out] = H[6 * k]; out[max_out + n_out] = H[6 * k + 1]; out[2 * max_out + n_out] = H[6 * k + 2];
                out[3 * max_out + n_out] = H[6 * k + 3]; out[4 * max_out + n_out] = H[6 * k + 4];
            }
#undef HN
            free(H);
            rc = n_out;
        }
    }
done:
    free(D.nd); free(D.lk); free(D.hk_ent); free(D.hk_next); free(sfw);
    return rc;
}
