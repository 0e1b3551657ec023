/*
 * s3a_nbest.hip -- N-best lists on the lattice the second pass leaves: what srch_TST_nbest_impl
 * (sphinx3/src/libs3decoder/libsearch/srch_time_switch_tree.c:1442-1492) does between vithist_dag_build and the list's file,
 * in the library, on the arrays s3a_uttdec_lattice / s3a_dagpass_lattice hand out:
 *
 *   dag_remove_unreachable     libsearch/dag.c:303-365
 *   dag_bypass_filler_nodes    libsearch/dag.c:1037-1075   (dag_update_link / dag_link :186-300)
 *   dag_compute_hscr           libsearch/dag.c:521-587
 *   dag_remove_bypass_links    libsearch/dag.c:1077-1112
 *   astar_init / astar_next_ppath / nbest_search   libsearch/astar.c:466-716 (aheap_insert / aheap_pop :231-296, ppath_dup / ppath_insert :304-395,
 *                                                   nbest_hyp_write / ppath_seg_write :417-462)
 *
 * Host code of the library (like the lattice files' formatters in s3a_host.c): A* is ONE best-first chain -- pop the best partial path, extend
 * it over its node's links, insert -- whose every step depends on the last; what decides the list's bytes is the ORDER of equals, and that
 * order is the shape of the reference's own unbalanced-by-value, balanced-by-count binary heap (insert goes to the lighter side, the left on
 * ties; pop pulls up the better child, the left on ties).  So the heap, the duplicate table (200 003 chains by history hash) and the
 * arithmetic (a double product truncated per link, as `int32 = lwf * lm_tg_score`) are restated as they are; the lists (succlist / predlist:
 * built by head insertion) are index chains in the reference's order, which the lattice's arrays carry (include/cmusphinx_amd.h).
 * One difference in form: a link is ONE record that sits in its source's successor chain and its target's predecessor chain (the reference
 * keeps two records and updates both).
 * Parity: tests/test_gpu_dag.py (lists byte-identical to the unmodified reference's -nbestdir files, tidigits + RM1, -bestpathlw), and
 * tests/test_nbest_host.py (no device: recorded lattices against recorded lists).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <stdarg.h>
#include <string>
#include <vector>

#include "s3a_internal.h"
#include "s3a_lm3g.h"

#pragma clang fp contract(off)

namespace {

inline int32_t nb_add32(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }

struct Edge { int32_t from, to, ascr, hscr, ef, byp, snext, pnext; };          /* snext / pnext: the next link of the source's successor chain / the target's predecessor chain */
struct PPath { int32_t hist, lmhist, node, lscr, pscr, tscr, pruned, hashnext; uint32_t histhash; };
struct HeapNode { int32_t pp, nl, nr, left, right; };

#define NB_HISTHASH_MOD 200003

struct Search {
    const s3a_dag_cfg_t *cfg;
    const s3a_lm3g_s *lm;
    std::vector<int32_t> wid, sf, shead, phead;
    std::vector<char> reach, listed;
    std::vector<Edge> E;
    int32_t n_node, root, end, final_ascr, nlink, lmop, maxlmop;
    /* A* */
    std::vector<PPath> pp;
    std::vector<HeapNode> hp;
    std::vector<int32_t> hfree, hash;
    int32_t heap_root, beam, besttscr, n_pop, n_exp, maxppath;

    bool filler(int32_t w) const { return cfg->is_filler[w] != 0; }
    int32_t base(int32_t w) const { return cfg->basewid[w]; }
    /* lm->dict2lmwid[] with linksilences in force (kbcore.c:191-206): <s> and </s> have LM ids while the second pass runs */
    int32_t lmid(int32_t bw) const { return bw < 0 ? -1 : bw == cfg->startwid ? cfg->start_lwid : bw == cfg->finishwid ? cfg->finish_lwid : cfg->lwid[bw]; }
    /* lm_tg_score (lm.c:1661-1833) on the handle's host copy */
    int32_t tg(int32_t bw0, int32_t bw1, int32_t bw2) const { return s3a_lm3g_tg_score(lm, lmid(bw0), lmid(bw1), lmid(bw2), bw2); }

    /* dag_link (dag.c:186-237): both chains by head insertion; -1: -maxedge exceeded */
    int32_t link(int32_t pd, int32_t d, int32_t ascr, int32_t ef, int32_t byp)
    {
        if (ascr > 0) return 0;                             /* ("silently refuse to create positive edges") */
        Edge e = { pd, d, ascr, 0, ef, byp, shead[pd], phead[d] };
        E.push_back(e);
        shead[pd] = phead[d] = (int32_t)E.size() - 1;
        nlink++;
        return nlink > cfg->maxedge ? -1 : 0;
    }
    /* dag_update_link (dag.c:277-300) for a bypass link: the pair's bypass link is replaced when the new score is better */
    int32_t update_link(int32_t pd, int32_t d, int32_t ascr, int32_t ef)
    {
        int32_t l = shead[pd];
        for (; l >= 0; l = E[l].snext) if (E[l].to == d && E[l].byp) break;
        if (l < 0) return link(pd, d, ascr, ef, 1);
        if (E[l].ascr < ascr) { E[l].ascr = ascr; E[l].ef = ef; }
        return 0;
    }
    void drop_from_chains(bool (*gone)(const Search &, const Edge &))
    {
        for (int32_t k = 0; k < n_node; k++) {
            int32_t *at = &shead[k];
            while (*at >= 0) { if (gone(*this, E[*at])) *at = E[*at].snext; else at = &E[*at].snext; }
            at = &phead[k];
            while (*at >= 0) { if (gone(*this, E[*at])) *at = E[*at].pnext; else at = &E[*at].pnext; }
        }
    }

    /* aheap_insert (astar.c:231-263): the better of (root, new) stays, the other goes down the lighter side -- the left one on ties */
    int32_t heap_insert(int32_t root, int32_t nw)
    {
        if (root < 0) {
            int32_t h;
            if (!hfree.empty()) { h = hfree.back(); hfree.pop_back(); } else { h = (int32_t)hp.size(); hp.push_back(HeapNode()); }
            hp[h].pp = nw; hp[h].left = hp[h].right = -1; hp[h].nl = hp[h].nr = 0;
            return h;
        }
        const int32_t old = hp[root].pp;
        if (pp[old].tscr < pp[nw].tscr) { hp[root].pp = nw; nw = old; }
        if (hp[root].nl > hp[root].nr) { const int32_t c = heap_insert(hp[root].right, nw); hp[root].right = c; hp[root].nr++; }
        else { const int32_t c = heap_insert(hp[root].left, nw); hp[root].left = c; hp[root].nl++; }
        return root;
    }
    /* aheap_pop (astar.c:270-296) */
    int32_t heap_pop(int32_t root)
    {
        const int32_t l = hp[root].left, r = hp[root].right;
        if (l < 0) {
            if (r < 0) { hfree.push_back(root); return -1; }
            hp[root].pp = hp[r].pp; hp[root].right = heap_pop(r); hp[root].nr--;
        }
        else if (r < 0 || pp[hp[l].pp].tscr >= pp[hp[r].pp].tscr) { hp[root].pp = hp[l].pp; hp[root].left = heap_pop(l); hp[root].nl--; }
        else { hp[root].pp = hp[r].pp; hp[root].right = heap_pop(r); hp[root].nr--; }
        return root;
    }
    /* ppath_dup (astar.c:304-342) */
    bool dup(int32_t hl, int32_t lmhist, int32_t node, uint32_t hval, int32_t pscr)
    {
        for (; hl >= 0; hl = pp[hl].hashnext) {
            if (pp[hl].node != node || pp[hl].histhash != hval) continue;
            int32_t h1 = pp[hl].lmhist, h2 = lmhist;
            for (; h1 >= 0 && h2 >= 0; h1 = pp[h1].lmhist, h2 = pp[h2].lmhist)
                if (h1 == h2 || base(wid[pp[h1].node]) != base(wid[pp[h2].node])) break;
            if (h1 == h2) {                                 /* the same history exists */
                if (pp[hl].pscr >= pscr) return true;
                pp[hl].pruned = 1;
                return false;
            }
        }
        return false;
    }
    /* ppath_insert (astar.c:350-395); false: -maxppath exceeded */
    void insert(int32_t top, int32_t l, int32_t lscr)
    {
        const int32_t pscr = nb_add32(nb_add32(pp[top].pscr, E[l].ascr), lscr);
        const int32_t lmhist = filler(wid[pp[top].node]) ? pp[top].lmhist : top;
        if (lmhist < 0) return;                             /* (a filler as the lattice's first node: the reference dereferences NULL here) */
        const int32_t w = wid[pp[lmhist].node];
        uint32_t h = pp[lmhist].histhash - (uint32_t)w + (uint32_t)base(w);
        h = (h >> 5) | (h << 27);
        h += (uint32_t)wid[E[l].to];
        const uint32_t hmod = h % NB_HISTHASH_MOD;
        if (dup(hash[hmod], lmhist, E[l].to, h, pscr)) return;
        PPath p;
        p.node = E[l].to; p.hist = top; p.lmhist = lmhist; p.lscr = lscr; p.pscr = pscr; p.tscr = nb_add32(pscr, E[l].hscr);
        p.histhash = h; p.hashnext = hash[hmod]; p.pruned = 0;
        pp.push_back(p);
        hash[hmod] = (int32_t)pp.size() - 1;
        heap_root = heap_insert(heap_root, (int32_t)pp.size() - 1);
    }
    /* astar_next_ppath (astar.c:531-621): the next complete path, -1: none (heap empty, or a limit: *limit says which) */
    int32_t next_path(const char **limit)
    {
        while (heap_root >= 0) {
            const int32_t top = hp[heap_root].pp;
            heap_root = heap_pop(heap_root);
            n_pop++;
            if (pp[top].pruned) continue;
            if (pp[top].node == end) return top;
            int32_t bw0 = -1, bw1 = -1, q = filler(wid[pp[top].node]) ? pp[top].lmhist : top;
            if (q >= 0) {
                bw1 = base(wid[pp[q].node]);
                q = pp[q].lmhist;
                bw0 = q >= 0 ? base(wid[pp[q].node]) : -1;
            }
            for (int32_t l = shead[pp[top].node]; l >= 0; l = E[l].snext) {
                const int32_t bw2 = base(wid[E[l].to]);
                const int32_t lscr = filler(bw2) ? cfg->fillpen[bw2] : (int32_t)(cfg->lwf * (double)tg(bw0, bw1, bw2));
                if (lmop++ > maxlmop) { *limit = "Max LM ops exceeded"; return -1; }
                const int32_t pscr = nb_add32(nb_add32(pp[top].pscr, E[l].ascr), lscr), tscr = nb_add32(pscr, E[l].hscr);
                if (nb_add32(tscr, -beam) >= besttscr) {
                    insert(top, l, lscr);
                    if ((int32_t)pp.size() - 1 > maxppath) { *limit = "Max PPATH limit exceeded"; return -1; }
                    if (tscr > besttscr) besttscr = tscr;
                }
            }
            n_exp++;
        }
        return -1;
    }
};

bool edge_bypass(const Search &, const Edge &e) { return e.byp != 0; }
bool edge_dead(const Search &s, const Edge &e) { return !s.reach[e.to] || !s.reach[e.from]; }

void
out(std::string &o, const char *fmt, ...)
{
    char b[512];
    va_list ap;
    va_start(ap, fmt);
    const int n = vsnprintf(b, sizeof b, fmt, ap);
    va_end(ap);
    if (n < (int)sizeof b) { o.append(b, n > 0 ? (size_t)n : 0); return; }
    std::vector<char> big((size_t)n + 1);
    va_start(ap, fmt);
    vsnprintf(big.data(), big.size(), fmt, ap);
    va_end(ap);
    o.append(big.data(), (size_t)n);
}

}   /* namespace */

struct s3a_nbest_s {
    std::string text;
    int32_t status, n_hyp, n_pop, n_exp, n_ppath, n_bypass, beam;
};

extern "C" void
s3a_nbest_free(s3a_nbest_t *nb)
{
    delete nb;
}

extern "C" int32_t
s3a_nbest_result(const s3a_nbest_t *nb, const char **text, int64_t *len, int32_t *n_hyp, int32_t *counts4)
{
    if (!nb) return S3A_EINVAL;
    if (text) *text = nb->text.c_str();
    if (len) *len = (int64_t)nb->text.size();
    if (n_hyp) *n_hyp = nb->n_hyp;
    if (counts4) { counts4[0] = nb->n_pop; counts4[1] = nb->n_exp; counts4[2] = nb->n_ppath; counts4[3] = nb->n_bypass; }
    return nb->status;
}

extern "C" s3a_nbest_t *
s3a_lattice_nbest(const s3a_lm3g_t *lm, const s3a_dag_cfg_t *cfg, const s3a_nbest_opts_t *o, const s3a_lat_info_t *info,
                  const s3a_lat_node_t *nodes, const s3a_lat_link_t *links, const char *const *wordstr)
{
    if (!lm || !cfg || !o || !info || !nodes || (info->n_links > 0 && !links) || !wordstr || info->n_nodes <= 0 || info->initial < 0
        || info->initial >= info->n_nodes || info->final < 0 || info->final >= info->n_nodes || !cfg->basewid || !cfg->is_filler || !cfg->lwid
        || !cfg->fillpen) {
        s3a_set_error("s3a_lattice_nbest: bad arguments");
        return NULL;
    }
    const int32_t N = info->n_nodes, M = info->n_links;
    for (int32_t k = 0; k < N; k++)
        if (nodes[k].wid < 0 || nodes[k].wid >= cfg->n_word) { s3a_set_error("s3a_lattice_nbest: node %d has word id %d", k, nodes[k].wid); return NULL; }
    for (int32_t j = 0; j < M; j++)
        if (links[j].from < 0 || links[j].from >= N || links[j].to < 0 || links[j].to >= N || (j > 0 && links[j].from < links[j - 1].from)) {
            s3a_set_error("s3a_lattice_nbest: link %d outside the lattice's nodes or out of source order", j);
            return NULL;
        }
    Search S;
    S.cfg = cfg; S.lm = lm; S.n_node = N; S.root = info->initial; S.end = info->final; S.final_ascr = info->final_ascr;
    S.wid.resize(N); S.sf.resize(N); S.shead.assign(N, -1); S.phead.assign(N, -1); S.reach.assign(N, 0); S.listed.assign(N, 1);
    for (int32_t k = 0; k < N; k++) { S.wid[k] = nodes[k].wid; S.sf[k] = nodes[k].sf; }
    /* the chains: a source's links in the arrays' order; a node's predecessors by source, ascending (head insertion from the last source
     * made to the first: vithist_dag_build's order, cmusphinx_amd.h) */
    S.E.resize(M);
    for (int32_t j = M - 1; j >= 0; j--) {
        Edge &e = S.E[j];
        e.from = links[j].from; e.to = links[j].to; e.ascr = links[j].ascr; e.hscr = 0; e.ef = links[j].ef; e.byp = 0;
        e.snext = S.shead[e.from]; S.shead[e.from] = j;
        e.pnext = S.phead[e.to]; S.phead[e.to] = j;
    }
    S.nlink = M; S.lmop = 0;
    S.maxlmop = cfg->maxlmop;
    { const long long k = (long long)cfg->maxlpf * info->n_frames; if (k > 0 && S.maxlmop > k) S.maxlmop = (int32_t)k; }
    s3a_nbest_t *nb = new s3a_nbest_s();
    nb->status = S3A_OK; nb->n_hyp = 0; nb->n_pop = nb->n_exp = nb->n_ppath = nb->n_bypass = 0; nb->beam = o->beam_logs3;
    /* (srch_time_switch_tree.c:1467-1469: a search that ended in a filler word has its final node turned into </s>) */
    if (S.filler(S.wid[S.end])) S.wid[S.end] = cfg->finishwid;
    /* dag_remove_unreachable: what has no path to the final node leaves the lists -- except the list's first node, which
     * dag.c:352-361 never unlinks (its chains are empty then) */
    {
        std::vector<int32_t> st(1, S.end);
        S.reach[S.end] = 1;
        while (!st.empty()) {
            const int32_t d = st.back(); st.pop_back();
            for (int32_t l = S.phead[d]; l >= 0; l = S.E[l].pnext) if (!S.reach[S.E[l].from]) { S.reach[S.E[l].from] = 1; st.push_back(S.E[l].from); }
        }
        for (int32_t k = 0; k < N; k++) {
            for (int32_t l = S.shead[k]; l >= 0; l = S.E[l].snext) if (!S.reach[k] || !S.reach[S.E[l].to]) S.nlink--;
            if (!S.reach[k] && k > 0) S.listed[k] = 0;
        }
        S.drop_from_chains(edge_dead);
    }
    /* dag_bypass_filler_nodes: in list order (later fillers first), every predecessor of a filler linked to the filler's non-filler
     * successors -- the links made for later fillers among them */
    bool over = false;
    for (int32_t d = 0; d < N && !over; d++) {
        if (!S.listed[d] || !S.filler(S.wid[d])) continue;
        for (int32_t pl = S.phead[d]; pl >= 0 && !over; pl = S.E[pl].pnext) {
            const int32_t pnode = S.E[pl].from, pef = S.E[pl].ef;
            const int32_t ascr = (int32_t)((double)S.E[pl].ascr + ((double)(cfg->fillpen[S.base(S.wid[d])] - cfg->wip) * cfg->lwf + (double)cfg->wip));
            for (int32_t sl = S.shead[d]; sl >= 0; sl = S.E[sl].snext) {
                const int32_t snode = S.E[sl].to;
                if (S.filler(S.wid[snode])) continue;
                const size_t before = S.E.size();
                if (S.update_link(pnode, snode, nb_add32(ascr, S.E[sl].ascr), pef) < 0) { over = true; break; }
                if (S.E.size() > before) nb->n_bypass++;
            }
        }
    }
    if (over) {             /* "maxedge limit (%d) exceeded": srch_TST_nbest_impl returns without a list */
        nb->status = S3A_ENOMEM; nb->n_hyp = -1;
        s3a_set_error("s3a_lattice_nbest: maxedge limit (%d) exceeded", cfg->maxedge);
        return nb;
    }
    /* dag_compute_hscr: per link the best score from its END to the end of the utterance, nodes in list order, a successor's value as it
     * stands when it is read; (bw0 is NOT restored after a filler successor: dag.c:549-552 overwrites the node's own word for the links
     * that follow) */
    for (int32_t d = 0; d < N; d++) {
        if (!S.listed[d]) continue;
        int32_t bw0 = S.filler(S.wid[d]) ? -1 : S.base(S.wid[d]);
        for (int32_t l1 = S.shead[d]; l1 >= 0; l1 = S.E[l1].snext) {
            const int32_t d1 = S.E[l1].to;
            if (d1 == S.end) { S.E[l1].hscr = 0; continue; }
            int32_t bw1 = S.filler(S.wid[d1]) ? -1 : S.base(S.wid[d1]);
            if (bw1 < 0) { bw1 = bw0; bw0 = -1; }
            int32_t best = INT_MIN;
            for (int32_t l2 = S.shead[d1]; l2 >= 0; l2 = S.E[l2].snext) {
                const int32_t d2 = S.E[l2].to;
                if (S.filler(S.wid[d2])) continue;
                const int32_t bw2 = S.base(S.wid[d2]);
                const int32_t h = (int32_t)((double)nb_add32(S.E[l2].hscr, S.E[l2].ascr) + cfg->lwf * (double)S.tg(bw0, bw1, bw2));
                if (h > best) best = h;
            }
            S.E[l1].hscr = best;
        }
    }
    /* dag_remove_bypass_links: the search itself walks the fillers */
    S.drop_from_chains(edge_bypass);
    /* nbest_search (astar.c:656-716) */
    std::string &t = nb->text;
    out(t, "# %s\n", o->uttid ? o->uttid : "");
    out(t, "# frames %d\n", info->n_frames);
    out(t, "# logbase %e\n", o->logbase);
    out(t, "# langwt %e\n", o->lw * cfg->lwf);
    out(t, "# inspen %e\n", o->wip);
    out(t, "# beam %e\n", o->beam);
    S.beam = o->beam_logs3; S.besttscr = INT_MIN; S.n_pop = S.n_exp = 0; S.maxppath = o->maxppath; S.heap_root = -1;
    S.hash.assign(NB_HISTHASH_MOD, -1);
    {
        PPath p;
        p.node = S.root; p.hist = -1; p.lmhist = -1; p.lscr = 0; p.pscr = 0; p.tscr = 0; p.histhash = (uint32_t)S.wid[S.root]; p.hashnext = -1; p.pruned = 0;
        S.pp.push_back(p);
        S.heap_root = S.heap_insert(-1, 0);
        S.hash[p.histhash % NB_HISTHASH_MOD] = 0;
    }
    auto rawscore = [&](int32_t s) { s -= o->lm_wip; return (int32_t)((float)s / o->lm_lw); };     /* lm_rawscore, lm.c:2172-2178 */
    int32_t besthyp = INT_MIN, worsthyp = INT_MAX, n_hyp = 0;
    const char *limit = NULL;
    std::vector<int32_t> chain;
    while (n_hyp < o->nbest) {
        const int32_t top = S.next_path(&limit);
        if (top < 0) break;
        /* nbest_hyp_write */
        const int32_t pscr = nb_add32(S.pp[top].pscr, S.final_ascr);
        int32_t lscr = 0, lscr_base = 0;
        chain.clear();
        for (int32_t q = top; q >= 0; q = S.pp[q].hist) {
            if (S.pp[q].hist >= 0) lscr_base += rawscore(S.pp[q].lscr);
            lscr += S.pp[q].lscr;
            chain.push_back(q);
        }
        out(t, "T %d A %d L %d", pscr, pscr - lscr, lscr_base);
        for (size_t c = chain.size(); c-- > 0;) {           /* ppath_seg_write: from the start node on */
            const int32_t q = chain[c];
            const int32_t ascr = c == 0 ? pscr - S.pp[top].pscr : S.pp[chain[c - 1]].pscr - S.pp[q].pscr - S.pp[chain[c - 1]].lscr;
            out(t, " %d %d %d %s", S.sf[S.pp[q].node], ascr, S.pp[q].hist >= 0 ? rawscore(S.pp[q].lscr) : 0, wordstr[S.wid[S.pp[q].node]]);
        }
        out(t, " %d\n", info->n_frames);
        n_hyp++;
        if (besthyp < S.pp[top].pscr) besthyp = S.pp[top].pscr;
        if (worsthyp > S.pp[top].pscr) worsthyp = S.pp[top].pscr;
    }
    out(t, "End; best %d worst %d diff %d beam %d\n", nb_add32(besthyp, S.final_ascr), nb_add32(worsthyp, S.final_ascr), nb_add32(worsthyp, -besthyp), S.beam);
    nb->n_hyp = n_hyp; nb->n_pop = S.n_pop; nb->n_exp = S.n_exp; nb->n_ppath = (int32_t)S.pp.size() - 1;
    if (limit) s3a_set_error("s3a_lattice_nbest: %s (the list so far is kept, as the reference keeps it)", limit);
    return nb;
}
