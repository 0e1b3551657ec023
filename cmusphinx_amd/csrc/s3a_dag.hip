/*
 * s3a_dag.hip -- sphinx3's SECOND PASS on the device: the word lattice from the first pass's
 * Viterbi history, the links around filler words, the best path under the trigram, its backtrace -- one workgroup
 * per utterance (decoder lane), every lane of an engine at once, the history table never leaving HBM.
 *
 * Replaces
 *   vithist_utt_end            libsearch/vithist.c:766-860      (device tables: d_dag_utt_end)
 *   vithist_dag_build          libsearch/vithist.c:1100-1311
 *   dag_bypass_filler_nodes    libsearch/dag.c:1037-1075  (dag_link / dag_update_link :186-300)
 *   dag_search / dag_bestpath  libsearch/dag.c:893-965, 397-484
 *   dag_backtrace              libsearch/dag.c:590-671
 *   srch_TST_bestpath_impl     libsearch/srch_time_switch_tree.c:1391-1440
 *
 * The reference keeps nodes and links in singly linked lists built by head insertion and decides ties by list order
 * (strict >: the first of equals wins).  Nothing is walked here; the list orders are restated as sort keys:
 *   - a lattice node = the history entries with one (start frame, word); the nodes that start in a frame, in list order
 *     = by FIRST entry id, descending (glist_add_ptr prepends); nidx = rank by (start frame, that order);
 *   - a node's exits = per end frame its best entry (earliest on ties); every exit links to every kept node that starts
 *     in the next frame, so ALL real links into the nodes of frame f+1 come from the same exit list A_f, and a node's
 *     predecessor list is A_f by nidx of the source, descending -- one sorted list per frame, no per-link storage
 *     beyond the path scores;
 *   - dag_bypass_filler_nodes visits fillers latest first and keeps ONE bypass link per (predecessor, successor) pair
 *     with the best score, the first of equals: a 64-bit atomicMax of (score, nidx of the filler, continues-in-a-
 *     bypass-link) per level of start frames; the pair's place in its successor's predecessor list (bypass links are
 *     prepended as they are made: the most recent first) follows from the LATEST filler that links the pair;
 *   - dag_bestpath's recursion = levels of start frames, a thread per link, the candidates walked in list order with
 *     the reference's own lazy rule (the LM is consulted only for a candidate that can still win), so the count of LM
 *     operations -- the reference gives up beyond -maxlmop / -maxlpf -- is the reference's.
 * Parity: tests/test_gpu_dag.py (against the test infrastructure's CPU restatement on recorded tables, and whole decodes
 * with -bestpath 1 against the unmodified reference).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <vector>

#include "s3a_device.h"
#include "s3a_wordlevel.h"
#include "s3a_lm3g.h"
#include "s3a_dag.h"

#pragma clang fp contract(off)

#define DG_T WL_THREADS

__device__ __forceinline__ uint32_t
dg_hash(unsigned long long k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return (uint32_t)k;
}

/* insert / find key (non-zero) in an open-addressing table of size mask + 1; returns the slot, -1 when full;
 * *fresh = this call made the slot */
__device__ __forceinline__ int32_t
dg_insert(unsigned long long *keys, int32_t mask, unsigned long long key, bool *fresh)
{
    uint32_t h = dg_hash(key) & (uint32_t)mask;
    for (int32_t probe = 0; probe <= mask; probe++) {
        const unsigned long long old = atomicCAS(&keys[h], 0ull, key);
        if (old == 0ull) { if (fresh) *fresh = true; return (int32_t)h; }
        if (old == key) { if (fresh) *fresh = false; return (int32_t)h; }
        h = (h + 1) & (uint32_t)mask;
    }
    return -1;
}

__device__ __forceinline__ int32_t
dg_find(const unsigned long long *keys, int32_t mask, unsigned long long key)
{
    uint32_t h = dg_hash(key) & (uint32_t)mask;
    for (int32_t probe = 0; probe <= mask; probe++) {
        const unsigned long long k = WL_ALOAD(&keys[h]);
        if (k == key) return (int32_t)h;
        if (k == 0ull) return -1;
        h = (h + 1) & (uint32_t)mask;
    }
    return -1;
}

/* the entry / exit hashes are allocated for the engine's largest possible table (millions of entries); an utterance uses -- and
 * k_dag_reset clears -- the power of two that its own entries need (monotone in n, so a bound on n bounds the slots) */
__device__ __forceinline__ int32_t
dg_eff_mask(int32_t h1mask, int32_t n_entry)
{
    int32_t m = 1024;
    while (m < 2 * (n_entry + 4) && m - 1 < h1mask) m <<= 1;
    return m - 1 < h1mask ? m - 1 : h1mask;
}

/* phases of one workgroup exchange data through global memory, much of it written by atomics (which live in L2):
 * release, barrier, acquire (the agent-scope fences write back / invalidate the CU's vector L1) */
#define DG_BAR() do { __threadfence(); __syncthreads(); __threadfence(); } while (0)

__device__ __forceinline__ int32_t
dg_lmid(const DagShared &G, int32_t w)
{
    const int32_t b = G.basewid[w];
    if (b == G.startwid) return G.start_lwid;            /* linksilences, kbcore.c:191-206 */
    if (b == G.finishwid) return G.finish_lwid;
    return G.lwid[b];
}

/* descending insertion sort of a short segment by key[] (one thread) */
__device__ __forceinline__ void
dg_sort_desc(int32_t *v, const int32_t *key, int32_t n)
{
    for (int32_t a = 1; a < n; a++) {
        const int32_t x = v[a], kx = key ? key[x] : x;
        int32_t b = a - 1;
        while (b >= 0 && (key ? key[v[b]] : v[b]) < kx) { v[b + 1] = v[b]; b--; }
        v[b + 1] = x;
    }
}

/*
 * vithist_utt_end (vithist.c:766-860) on the lane's device table: the best transition into </s> from the last frame
 * that has entries (the earliest of equals), a silence entry over the rest when the search died early, the </s>
 * entry -- appended to the table (the second pass reads them like any other entry) -- and the first pass's own
 * hypothesis (vithist_backtrace), whose (word, start frame) pairs the lattice must keep.
 * One workgroup; returns the </s> entry's id (endid) through L.io[], -1: no word exit at all.
 */
__device__ void
d_dag_utt_end(const DagShared &G, const DagLane &L, const WLm &lm)
{
    __shared__ unsigned long long s_best;
    __shared__ int32_t s_f, s_nent;
    const int32_t tid = threadIdx.x;
    const int32_t n_frm = L.tab.st[1];
    if (tid == 0) {
        int32_t f;
        for (f = n_frm - 1; f >= 0; --f)
            if (L.tab.frame_start[f] < L.tab.frame_start[f + 1]) break;
        s_f = f; s_best = 0ull; s_nent = L.tab.st[0];
    }
    __syncthreads();
    const int32_t f = s_f;
    if (f < 0) { if (tid == 0) { L.io[DG_IO_ENDID] = -1; L.io[DG_IO_NENT] = s_nent; L.io[DG_IO_NHYP] = 0; } __syncthreads(); return; }
    const int32_t sv = L.tab.frame_start[f], nsv = L.tab.frame_start[f + 1];
    for (int32_t i = sv + tid; i < nsv; i += DG_T) {
        const int32_t s = add32(L.tab.score[i], wl_tg_score(lm, L.tab.lw1[i], L.tab.lw0[i], G.finish_lwid, G.finishwid));
        atomicMax(&s_best, wl_pack(s, (uint32_t)i));        /* bestscore < scr: the earliest of equals */
    }
    __syncthreads();
    if (tid == 0) {
        int32_t n = s_nent, bestvh = (int32_t)(0xffffffffu - (uint32_t)(s_best & 0xffffffffull));
        int32_t best = (int32_t)((uint32_t)(s_best >> 32) ^ 0x80000000u);
        int32_t last = bestvh;
        if (n + 2 > L.tab.cap) { L.io[DG_IO_ENDID] = -1; L.io[DG_IO_NENT] = n; L.io[DG_IO_NHYP] = 0; L.io[DG_IO_STATUS] = DG_E_CAP; }
        else {
            if (f != n_frm - 1) {
                /* the dummy silence entry (vithist_rescore with the silence word, vithist.c:817-826) */
                const int32_t ps = L.tab.score[bestvh], pen = G.fillpen[G.silwid];
                L.tab.wid[n] = G.silwid; L.tab.sf[n] = L.tab.ef[bestvh] + 1; L.tab.ef[n] = n_frm - 1;
                L.tab.ascr[n] = add32(ps, -ps); L.tab.lscr[n] = pen; L.tab.score[n] = add32(ps, pen); L.tab.pred[n] = bestvh;
                L.tab.lw0[n] = L.tab.lw0[bestvh]; L.tab.lw1[n] = L.tab.lw1[bestvh]; L.tab.type[n] = -1;
                best = add32(L.tab.score[n], wl_tg_score(lm, L.tab.lw1[bestvh], L.tab.lw0[bestvh], G.finish_lwid, G.finishwid));
                last = n;
                n++;
            }
            L.tab.wid[n] = G.finishwid; L.tab.sf[n] = L.tab.ef[last] < 0 ? 0 : L.tab.ef[last] + 1; L.tab.ef[n] = n_frm;
            L.tab.ascr[n] = 0; L.tab.lscr[n] = add32(best, -L.tab.score[last]); L.tab.score[n] = best; L.tab.pred[n] = last;
            L.tab.lw0[n] = G.finish_lwid; L.tab.lw1[n] = G.finish_lwid; L.tab.type[n] = 0;
            L.io[DG_IO_ENDID] = n;
            L.io[DG_IO_NENT] = n + 1;
            /* vithist_backtrace, vithist.c:1066-1100: id > 0 */
            int32_t k = 0;
            int32_t id = n;
            for (; id > 0 && k < G.hyp_cap; id = L.tab.pred[id], k++) { L.hyp_wid[k] = L.tab.wid[id]; L.hyp_sf[k] = L.tab.sf[id]; }
            if (id > 0) L.io[DG_IO_STATUS] = DG_E_CAP;         /* (cannot happen: hyp_cap >= the frames + 4, a word takes a frame) */
            L.io[DG_IO_NHYP] = k;
            L.io[DG_IO_FIRSTSCORE] = best;
        }
    }
    DG_BAR();
}

__global__ void __launch_bounds__(DG_T)
k_dag_pass(DagShared G, const DagLane *__restrict__ lanes, WLm lm, int32_t do_utt_end, int32_t use_active, const int32_t *__restrict__ lane_ids)
{
    /* (lane_ids: the lanes that have ended at a refill event of a queue; NULL: lane = workgroup) */
    const DagLane &L = lanes[lane_ids ? lane_ids[blockIdx.x] : (int32_t)blockIdx.x];
    const int32_t tid = threadIdx.x;
    __shared__ int32_t s_err, s_cnt, s_any, s_lmop;
    __shared__ unsigned long long s_max;
    if (use_active && !L.io[DG_IO_ACTIVE]) return;
    if (tid == 0) { s_err = 0; s_lmop = 0; }
    if (tid == 0) { L.io[DG_IO_STATUS] = 0; L.io[DG_IO_NWORDS] = 0; L.io[DG_IO_NNODE] = 0; L.io[DG_IO_NLINK] = 0; }
    __syncthreads();
    if (do_utt_end) d_dag_utt_end(G, L, lm);
    const int32_t E = L.io[DG_IO_NENT], endid = L.io[DG_IO_ENDID], n_frm = L.tab.st[1], n_hyp = L.io[DG_IO_NHYP];
    if (endid < 0 || E <= 0 || n_frm <= 0 || n_frm + 2 > G.F) {
        if (tid == 0 && L.io[DG_IO_STATUS] == 0) L.io[DG_IO_STATUS] = endid < 0 ? DG_E_NOEXIT : DG_E_CAP;
        return;
    }
    const int32_t F1 = n_frm + 1;           /* start frames 0 .. n_frm */
    const int32_t hmask = dg_eff_mask(G.h1mask, E);
    const int32_t *wid = L.tab.wid, *sf = L.tab.sf, *ef = L.tab.ef, *ascr = L.tab.ascr, *score = L.tab.score;

    /* ---- P1: nodes = distinct (start frame', word) ---- */
    for (int32_t i = tid; i <= G.F; i += DG_T) { L.ncnt[i] = 0; L.nfill[i] = 0; L.kcnt[i] = 0; L.acnt[i] = 0; L.afill[i] = 0; }
    __syncthreads();
    for (int32_t i = tid; i < E; i += DG_T) {
        const int32_t s = sf[i] < 0 ? 0 : (sf[i] == 0 ? 1 : sf[i]), e = sf[i] < 0 ? 0 : ef[i];     /* "MAJOR HACK", vithist.c:1129-1146 */
        L.sfp[i] = s; L.efp[i] = e;
        if (s > n_frm || e > n_frm || e < 0) { s_err = DG_E_TABLE; continue; }
        bool fresh;
        const int32_t slot = dg_insert(L.h1key, hmask, (unsigned long long)s * (unsigned long long)G.n_word + (unsigned long long)wid[i] + 1ull, &fresh);
        if (slot < 0) { s_err = DG_E_CAP; continue; }
        L.eslot[i] = slot;
        atomicMin(&L.h1first[slot], i);
        atomicMax(&L.h1last[slot], i);
    }
    DG_BAR();
    if (s_err) { if (tid == 0) L.io[DG_IO_STATUS] = s_err; return; }
    for (int32_t i = tid; i < E; i += DG_T)
        if (WL_ALOAD(&L.h1first[L.eslot[i]]) == i) atomicAdd(&L.ncnt[L.sfp[i]], 1);
    DG_BAR();
    const int32_t NN = wl_scan<false>(L.ncnt, L.nbase, F1, 0);
    if (tid == 0) L.nbase[F1] = NN;
    DG_BAR();
    for (int32_t i = tid; i < E; i += DG_T)
        if (WL_ALOAD(&L.h1first[L.eslot[i]]) == i) L.nfirst[L.nbase[L.sfp[i]] + atomicAdd(&L.nfill[L.sfp[i]], 1)] = i;
    DG_BAR();
    for (int32_t f = tid; f < F1; f += DG_T) dg_sort_desc(L.nfirst + L.nbase[f], (const int32_t *)NULL, L.ncnt[f]);   /* list order */
    DG_BAR();
    for (int32_t p = tid; p < NN; p += DG_T) {
        const int32_t i = L.nfirst[p], slot = L.eslot[i];
        L.h1node[slot] = p;
        L.nwid[p] = wid[i]; L.nsf[p] = L.sfp[i]; L.nfef[p] = L.efp[i]; L.nlef[p] = L.efp[WL_ALOAD(&L.h1last[slot])];
        L.nkeep[p] = (L.nlef[p] - L.nfef[p] > G.min_endfr) ? 1 : 0;
        L.nhk[p] = 0; L.phead[p] = -1; L.shead[p] = -1; L.reach[p] = 0; L.bpcnt[p] = 0;
    }
    DG_BAR();
    /* ---- P2: what is kept whatever its duration ---- */
    for (int32_t k = tid; k < n_hyp; k += DG_T) {
        const int32_t hs = L.hyp_sf[k] == 0 ? 1 : L.hyp_sf[k];
        if (hs < 0 || hs > n_frm) continue;
        const int32_t slot = dg_find(L.h1key, hmask, (unsigned long long)hs * (unsigned long long)G.n_word + (unsigned long long)L.hyp_wid[k] + 1ull);
        if (slot >= 0) L.nkeep[L.h1node[slot]] = 1;
    }
    __shared__ int32_t s_root, s_end, s_fin;
    if (tid == 0) {
        s_root = L.ncnt[0] > 0 ? L.nbase[0] : -1;
        s_fin = L.ncnt[n_frm] > 0 ? L.nbase[n_frm] : -1;
        s_end = L.h1node[L.eslot[endid]];
        if (s_root < 0 || s_fin < 0 || L.nwid[s_root] != G.startwid || L.nwid[s_fin] != G.finishwid) s_err = DG_E_TABLE;
    }
    DG_BAR();
    if (s_err) { if (tid == 0) L.io[DG_IO_STATUS] = s_err; return; }
    const int32_t root = s_root, endn = s_end;
    if (tid == 0) {
        L.nkeep[root] = 1; L.nkeep[s_fin] = 1; L.nkeep[endn] = 1;
        if (G.is_filler[L.nwid[endn]]) L.nwid[endn] = G.finishwid;        /* srch_time_switch_tree.c:1406-1408 */
    }
    DG_BAR();
    for (int32_t p = tid; p < NN; p += DG_T) {
        L.nfil[p] = G.is_filler[L.nwid[p]];
        if (L.nkeep[p]) atomicAdd(&L.kcnt[L.nsf[p]], 1);
    }
    DG_BAR();
    const int32_t NK = wl_scan<false>(L.kcnt, L.kbase, F1, 0);
    if (tid == 0) L.kbase[F1] = NK;
    DG_BAR();
    for (int32_t f = tid; f < F1; f += DG_T) {          /* the kept nodes of a frame, in list order */
        int32_t k = L.kbase[f];
        for (int32_t p = L.nbase[f]; p < L.nbase[f] + L.ncnt[f]; p++)
            if (L.nkeep[p]) { L.knode[k] = p; L.nkpos[p] = k - L.kbase[f]; k++; }
    }
    DG_BAR();
    /* ---- P3: a node's exits: per end frame its best entry, the earliest of equals (vithist.c:1176-1189) ---- */
    for (int32_t i = tid; i < E; i += DG_T) {
        const int32_t p = L.h1node[L.eslot[i]];
        L.enode[i] = p;
        const int32_t slot = dg_insert(L.h2key, hmask, (unsigned long long)p * (unsigned long long)(G.F + 1) + (unsigned long long)L.efp[i] + 1ull, (bool *)NULL);
        if (slot < 0) { s_err = DG_E_CAP; continue; }
        L.eslot[i] = slot;                               /* (now the exit's slot) */
        atomicMax(&L.h2best[slot], wl_pack(score[i], (uint32_t)i));
    }
    DG_BAR();
    if (s_err) { if (tid == 0) L.io[DG_IO_STATUS] = s_err; return; }
    /* ---- P4: A_e = the exits with end frame e of kept nodes (that may link at all), by nidx of the node, descending ---- */
    for (int32_t i = tid; i < E; i += DG_T) {
        const unsigned long long b = WL_ALOAD(&L.h2best[L.eslot[i]]);
        const bool is_exit = (int32_t)(0xffffffffu - (uint32_t)(b & 0xffffffffull)) == i;
        const int32_t p = L.enode[i];
        L.ehk[i] = is_exit ? 1 : 0;
        if (is_exit) atomicAdd(&L.nhk[p], 1);
        L.eapos[i] = -1;
        if (is_exit && L.nkeep[p] && L.nsf[p] < n_frm && L.efp[i] + 1 <= n_frm && ascr[i] <= 0) atomicAdd(&L.acnt[L.efp[i]], 1);
    }
    DG_BAR();
    (void)wl_scan<false>(L.acnt, L.abase, F1, 0);
    (void)wl_scan<false>(L.nhk, L.hkbase, NN, 0);
    DG_BAR();
    for (int32_t p = tid; p < NN; p += DG_T) L.nhk[p] = 0;
    DG_BAR();
    for (int32_t i = tid; i < E; i += DG_T) {
        if (!L.ehk[i]) continue;
        const int32_t p = L.enode[i];
        L.hkent[L.hkbase[p] + atomicAdd(&L.nhk[p], 1)] = i;        /* a node's exits (any order) */
        if (L.nkeep[p] && L.nsf[p] < n_frm && L.efp[i] + 1 <= n_frm && ascr[i] <= 0) L.aent[L.abase[L.efp[i]] + atomicAdd(&L.afill[L.efp[i]], 1)] = i;
    }
    DG_BAR();
    for (int32_t e = tid; e < F1; e += DG_T) {
        int32_t *seg = L.aent + L.abase[e];
        const int32_t n = L.acnt[e];
        dg_sort_desc(seg, L.enode, n);
        for (int32_t a = 0; a < n; a++) L.eapos[seg[a]] = a;
        L.lcnt[e] = (e + 1 <= n_frm) ? n * L.kcnt[e + 1] : 0;     /* real links: A_e x kept nodes of frame e + 1 */
    }
    DG_BAR();
    const int32_t LR = wl_scan<false>(L.lcnt, L.loff, F1, 0);
    if (tid == 0) { L.loff[F1] = LR; L.io[DG_IO_NNODE] = NK; L.io[DG_IO_NLINK] = LR; }
    DG_BAR();
    /* -maxedge is NOT enforced here: vithist_dag_build ignores dag_link's return (vithist.c:1161-1170); link_cap is this pass's memory */
    if (LR > G.link_cap) { if (tid == 0) L.io[DG_IO_STATUS] = DG_E_CAP; return; }
    /* real link (exit i of node x, node y of frame efp[i] + 1): id = loff[e] + eapos[i] * kcnt[e + 1] + nkpos[y] */
#define DG_RLINK(i_, y_) (L.loff[L.efp[i_]] + L.eapos[i_] * L.kcnt[L.efp[i_] + 1] + L.nkpos[y_])

    /* ---- P5: dag_bypass_filler_nodes, fillers latest first; a level = the fillers that start in one frame ---- */
    const double pen_wip = (double)G.wip;
    for (int32_t t = n_frm - 1; t >= 1; --t) {
        /* tasks: (kept filler d of frame t, exit a of A_{t-1}) */
        const int32_t nk = L.kcnt[t], na = L.acnt[t - 1];
        if (nk == 0 || na == 0) continue;           /* (uniform) */
        for (int32_t task = tid; task < nk * na; task += DG_T) {
            const int32_t d = L.knode[L.kbase[t] + task / na];
            if (!L.nfil[d]) continue;
            const int32_t ia = L.aent[L.abase[t - 1] + task % na], x = L.enode[ia];
            const int32_t a_val = (int32_t)((double)ascr[ia] + ((double)(G.fillpen[G.basewid[L.nwid[d]]] - G.wip) * G.lwf + pen_wip));
            /* one (predecessor, filler, successor link) event of dag_update_link: the pair's best score; of equals the
             * first in the reference's order = the later filler (larger nidx), and for one filler its bypass links
             * before its real links */
            auto update = [&](int32_t s, int32_t sa, unsigned long long kind) {
                const int32_t cand = add32(a_val, sa);
                if (cand > 0) { s_err = DG_E_POSEDGE; return; }
                bool fresh;
                const int32_t slot = dg_insert(L.bkey, G.bmask, (((unsigned long long)(uint32_t)x << 32) | (unsigned long long)(uint32_t)s) + 1ull, &fresh);
                if (slot < 0) { s_err = DG_E_CAP; return; }
                if (fresh) {
                    L.bnextp[slot] = atomicExch(&L.phead[x], slot);
                    L.bnexts[slot] = atomicExch(&L.shead[s], slot);
                    atomicAdd(&L.bpcnt[s], 1);
                }
                atomicMax(&L.bbest[slot], ((unsigned long long)((uint32_t)cand ^ 0x80000000u) << 32) | ((unsigned long long)(uint32_t)d << 1) | kind);
                atomicMax(&L.bdstar[slot], d);
            };
            /* succ(d): its bypass links (made at later levels), then its real links to non-filler nodes */
            for (int32_t bl = L.phead[d]; bl >= 0; bl = L.bnextp[bl])
                update((int32_t)((L.bkey[bl] - 1ull) & 0xffffffffull), (int32_t)((uint32_t)(WL_ALOAD(&L.bbest[bl]) >> 32) ^ 0x80000000u), 1ull);
            for (int32_t h = 0; h < L.nhk[d]; h++) {
                const int32_t ie = L.hkent[L.hkbase[d] + h], e2 = L.efp[ie];
                if (L.eapos[ie] < 0) continue;
                for (int32_t yy = 0; yy < L.kcnt[e2 + 1]; yy++) {
                    const int32_t s = L.knode[L.kbase[e2 + 1] + yy];
                    if (!L.nfil[s]) update(s, ascr[ie], 0ull);
                }
            }
        }
        DG_BAR();
    }
    if (s_err) { if (tid == 0) L.io[DG_IO_STATUS] = s_err; return; }
    /* a node's bypass predecessors, in list order: the most recently made first = by (latest filler, nidx of the
     * predecessor), both ascending */
    (void)wl_scan<false>(L.bpcnt, L.bpbase, NN, 0);
    DG_BAR();
    for (int32_t s = tid; s < NN; s += DG_T) {
        int32_t n = 0;
        int32_t *seg = L.bplist + L.bpbase[s];
        for (int32_t b = L.shead[s]; b >= 0; b = L.bnexts[b]) {
            const int32_t ds = L.bdstar[b], px = (int32_t)((L.bkey[b] - 1ull) >> 32);
            int32_t q = n - 1;
            while (q >= 0) {
                const int32_t ob = seg[q], od = L.bdstar[ob], op = (int32_t)((L.bkey[ob] - 1ull) >> 32);
                if (od > ds || (od == ds && op > px)) { seg[q + 1] = seg[q]; q--; } else break;
            }
            seg[q + 1] = b;
            n++;
        }
    }
    if (tid == 0) { int32_t nb = 0; for (int32_t s = 0; s < NN; s++) nb += L.bpcnt[s]; L.io[DG_IO_NBYPASS] = nb; }
    DG_BAR();
    /* dag_bypass_filler_nodes gives up when the links exceed -maxedge (dag.c:1034, :1382) and the reference then searches a
     * PARTLY bypassed lattice (srch_time_switch_tree.c:1408-1412) whose shape depends on where it stopped: not reproduced --
     * the utterance is reported as failed by this pass (status DG_E_CAP), never the batch */
    if (LR + L.io[DG_IO_NBYPASS] > G.maxedge) { if (tid == 0) L.io[DG_IO_STATUS] = DG_E_CAP; return; }

    /* ---- which links the recursion of dag_search evaluates: those that lead to the end node over non-filler nodes ---- */
    if (tid == 0) L.reach[endn] = 1;
    DG_BAR();
    for (int32_t t = n_frm - 1; t >= 0; --t) {
        for (int32_t k = tid; k < L.kcnt[t]; k += DG_T) {
            const int32_t d = L.knode[L.kbase[t] + k];
            if (L.nfil[d] || L.reach[d]) continue;
            int32_t r = 0;
            for (int32_t b = L.phead[d]; b >= 0 && !r; b = L.bnextp[b]) r = L.reach[(int32_t)((L.bkey[b] - 1ull) & 0xffffffffull)];
            for (int32_t h = 0; h < L.nhk[d] && !r; h++) {
                const int32_t ie = L.hkent[L.hkbase[d] + h], e2 = L.efp[ie];
                if (L.eapos[ie] < 0) continue;
                for (int32_t yy = 0; yy < L.kcnt[e2 + 1] && !r; yy++) { const int32_t s = L.knode[L.kbase[e2 + 1] + yy]; r = !L.nfil[s] && L.reach[s]; }
            }
            L.reach[d] = r;
        }
        DG_BAR();
    }
    /* ---- P6: dag_bestpath: a level = the non-filler nodes that start in one frame; a thread per link out of them ---- */
    for (int32_t t = 0; t < n_frm; t++) {
        const int32_t nk = L.kcnt[t];
        if (nk == 0) continue;
        /* the level's links: per kept non-filler node d with reach, its real links and its bypass links */
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        for (int32_t k = tid; k < nk; k += DG_T) {
            const int32_t d = L.knode[L.kbase[t] + k];
            if (L.nfil[d]) continue;
            for (int32_t h = 0; h < L.nhk[d]; h++) {
                const int32_t ie = L.hkent[L.hkbase[d] + h], e2 = L.efp[ie];
                if (L.eapos[ie] < 0) continue;
                for (int32_t yy = 0; yy < L.kcnt[e2 + 1]; yy++) {
                    const int32_t s = L.knode[L.kbase[e2 + 1] + yy];
                    if (L.nfil[s] || !L.reach[s]) continue;
                    const int32_t q = atomicAdd(&s_cnt, 1);
                    if (q < G.task_cap) { L.task[2 * q] = DG_RLINK(ie, s); L.task[2 * q + 1] = ie; }
                }
            }
            for (int32_t b = L.phead[d]; b >= 0; b = L.bnextp[b]) {
                const int32_t s = (int32_t)((L.bkey[b] - 1ull) & 0xffffffffull);
                if (!L.reach[s]) continue;
                const int32_t q = atomicAdd(&s_cnt, 1);
                if (q < G.task_cap) { L.task[2 * q] = -2 - b; L.task[2 * q + 1] = d; }
            }
        }
        DG_BAR();
        const int32_t nt = s_cnt;
        if (nt > G.task_cap) { if (tid == 0) L.io[DG_IO_STATUS] = DG_E_CAP; return; }
        for (int32_t q = tid; q < nt; q += DG_T) {
            const int32_t code = L.task[2 * q];
            int32_t d, s, l_ascr;
            if (code >= 0) { const int32_t ie = L.task[2 * q + 1]; d = L.enode[ie]; s = L.knode[L.kbase[L.efp[ie] + 1] + (code - L.loff[L.efp[ie]]) % L.kcnt[L.efp[ie] + 1]]; l_ascr = ascr[ie]; }
            else { const int32_t b = -2 - code; d = L.task[2 * q + 1]; s = (int32_t)((L.bkey[b] - 1ull) & 0xffffffffull); l_ascr = (int32_t)((uint32_t)(L.bbest[b] >> 32) ^ 0x80000000u); }
            const int32_t lw_d = dg_lmid(G, L.nwid[d]), lw_s = dg_lmid(G, L.nwid[s]), bw_s = G.basewid[L.nwid[s]];
            int32_t best = INT_MIN, hist = -3, lbest = 0, ops = 0;
            if (d == root) {
                /* the "stop" link of the root (dag.c:924-926): no predecessor, the bigram */
                const int32_t sc0 = add32(0, l_ascr);
                if (sc0 > best) {
                    const int32_t ls = (int32_t)(G.lwf * (double)wl_bg_score(lm, lw_d, lw_s, bw_s));
                    ops++;
                    const int32_t sc = add32(sc0, ls);
                    if (sc > best) { best = sc; hist = -1; lbest = ls; }
                }
            }
            else {
                /* pred(d): bypass links (most recent first), then the real links (A_{t-1} order) */
                const int32_t nb = L.bpcnt[d], na = t >= 1 ? L.acnt[t - 1] : 0;
                for (int32_t c = 0; c < nb + na; c++) {
                    int32_t pp, cp, ccode;
                    if (c < nb) {
                        const int32_t b = L.bplist[L.bpbase[d] + c];
                        pp = (int32_t)((L.bkey[b] - 1ull) >> 32);
                        if (L.nfil[pp]) continue;
                        cp = L.bpscr[b]; ccode = -2 - b;
                    }
                    else {
                        const int32_t ia = L.aent[L.abase[t - 1] + (c - nb)];
                        pp = L.enode[ia];
                        if (L.nfil[pp]) continue;
                        ccode = DG_RLINK(ia, d);
                        cp = L.lpscr[ccode];
                    }
                    const int32_t sc0 = add32(cp, l_ascr);
                    if (sc0 > best) {
                        const int32_t ls = (int32_t)(G.lwf * (double)wl_tg_score(lm, dg_lmid(G, L.nwid[pp]), lw_d, lw_s, bw_s));
                        ops++;
                        const int32_t sc = add32(sc0, ls);
                        if (sc > best) { best = sc; hist = ccode; lbest = ls; }
                    }
                }
            }
            if (code >= 0) { L.lpscr[code] = best; L.lhist[code] = hist; L.llscr[code] = lbest; }
            else { const int32_t b = -2 - code; L.bpscr[b] = best; L.bhist[b] = hist; L.blscr[b] = lbest; }
            if (ops) atomicAdd(&s_lmop, ops);
        }
        DG_BAR();
    }
    /* ---- dag_search's choice among the links into the end node (list order, the first of equals) ---- */
    if (tid == 0) { s_max = 0ull; s_any = 0; }
    __syncthreads();
    {
        const int32_t t = L.nsf[endn], nb = L.bpcnt[endn], na = t >= 1 ? L.acnt[t - 1] : 0;
        for (int32_t c = tid; c < nb + na; c += DG_T) {
            int32_t pp, cp;
            if (c < nb) { const int32_t b = L.bplist[L.bpbase[endn] + c]; pp = (int32_t)((L.bkey[b] - 1ull) >> 32); cp = L.bpscr[b]; }
            else { const int32_t ia = L.aent[L.abase[t - 1] + (c - nb)]; pp = L.enode[ia]; cp = L.lpscr[DG_RLINK(ia, endn)]; }
            if (L.nfil[pp]) continue;
            s_any = 1;
            atomicMax(&s_max, wl_pack(cp, (uint32_t)c));
        }
    }
    __syncthreads();
    if (tid != 0) return;
    L.io[DG_IO_LMOP] = s_lmop;
    {
        int32_t maxlmop = G.maxlmop;
        if (G.maxlpf > 0 && (long long)G.maxlpf * n_frm < (long long)maxlmop) maxlmop = G.maxlpf * n_frm;
        /* dag_search takes a link only on `l->pscr > bestscore` from (int32)0x80000000 (dag.c:918-928): links without a path do not count */
        const int32_t top = s_any ? (int32_t)((uint32_t)(s_max >> 32) ^ 0x80000000u) : INT_MIN;
        if (!s_any || top == INT_MIN || s_lmop > maxlmop) { L.io[DG_IO_STATUS] = DG_E_NOPATH; return; }      /* "Bestpath search failed" */
    }
    /* ---- dag_backtrace (one thread): from the end node back to the root, bypassed fillers restored ---- */
    {
        const int32_t t = L.nsf[endn], nb = L.bpcnt[endn];
        const int32_t c = (int32_t)(0xffffffffu - (uint32_t)(s_max & 0xffffffffull));
        int32_t code = c < nb ? -2 - L.bplist[L.bpbase[endn] + c] : DG_RLINK(L.aent[L.abase[t - 1] + (c - nb)], endn);
        int32_t fa = 0, n = 0, prev = -1;
        const int32_t cap = G.hyp_cap;
        for (int32_t h = 0; h < L.nhk[endn]; h++) { const int32_t ie = L.hkent[L.hkbase[endn] + h]; if (ef[ie] == n_frm) fa = ascr[ie]; }
        int32_t *ow = L.out, *osf = L.out + cap, *oef = L.out + 2 * cap, *oas = L.out + 3 * cap, *ols = L.out + 4 * cap;
#define DG_EMIT(w_, s_, e_, a_, l_) do { if (n >= cap) { L.io[DG_IO_STATUS] = DG_E_CAP; return; } ow[n] = (w_); osf[n] = (s_); oef[n] = (e_); oas[n] = (a_); ols[n] = (l_); n++; } while (0)
        DG_EMIT(L.nwid[endn], L.nsf[endn], n_frm - 1, fa, 0);
        prev = 0;
        L.io[DG_IO_SCORE] = add32(code >= 0 ? L.lpscr[code] : L.bpscr[-2 - code], fa);
        for (int32_t guard = 0; guard < G.hyp_cap; guard++) {
            const int32_t hist = code >= 0 ? L.lhist[code] : (code == -1 ? -3 : L.bhist[-2 - code]);
            if (prev >= 0) ols[prev] = code == -1 ? 0 : (code >= 0 ? L.llscr[code] : L.blscr[-2 - code]);
            if (code == -1) break;                          /* the root's stop link */
            if (code >= 0) {
                /* real link: the word of its source node, leaving at the link's frame */
                int32_t e2 = 0;
                { int32_t lo = 0, hi = n_frm; while (lo < hi) { const int32_t mid = (lo + hi + 1) >> 1; if (L.loff[mid] <= code) lo = mid; else hi = mid - 1; } e2 = lo;
                  while (L.lcnt[e2] == 0 || code >= L.loff[e2] + L.lcnt[e2]) e2++; }
                const int32_t ia = L.aent[L.abase[e2] + (code - L.loff[e2]) / L.kcnt[e2 + 1]], x = L.enode[ia];
                DG_EMIT(L.nwid[x], L.nsf[x], e2, ascr[ia], 0);
                prev = n - 1;
            }
            else {
                /* bypass link p -> s: p, then the fillers it stands for (dag.c:622-657); emitted in reverse here */
                int32_t b = -2 - code, p = (int32_t)((L.bkey[b] - 1ull) >> 32), first = n;
                const int32_t s = (int32_t)((L.bkey[b] - 1ull) & 0xffffffffull);
                int32_t src = p, is_first = 1;
                for (;;) {
                    const unsigned long long bb = L.bbest[b];
                    const int32_t d = (int32_t)((bb & 0xffffffffull) >> 1), kind = (int32_t)(bb & 1ull);
                    /* src -> d is a real link: src's exit at frame nsf[d] - 1 */
                    int32_t ie = -1;
                    for (int32_t h = 0; h < L.nhk[src]; h++) { const int32_t i2 = L.hkent[L.hkbase[src] + h]; if (L.efp[i2] == L.nsf[d] - 1) ie = i2; }
                    if (ie < 0) { L.io[DG_IO_STATUS] = DG_E_TABLE; return; }
                    DG_EMIT(L.nwid[src], L.nsf[src], L.nsf[d] - 1, ascr[ie], is_first ? 0 : (int32_t)(G.lwf * (double)G.fillpen[G.basewid[L.nwid[src]]]));
                    is_first = 0;
                    src = d;
                    if (kind) {
                        b = dg_find(L.bkey, G.bmask, (((unsigned long long)(uint32_t)d << 32) | (unsigned long long)(uint32_t)s) + 1ull);
                        if (b < 0) { L.io[DG_IO_STATUS] = DG_E_TABLE; return; }
                        continue;
                    }
                    /* the last filler: its real link to s */
                    ie = -1;
                    for (int32_t h = 0; h < L.nhk[d]; h++) { const int32_t i2 = L.hkent[L.hkbase[d] + h]; if (L.efp[i2] == L.nsf[s] - 1) ie = i2; }
                    if (ie < 0) { L.io[DG_IO_STATUS] = DG_E_TABLE; return; }
                    DG_EMIT(L.nwid[d], L.nsf[d], L.nsf[s] - 1, ascr[ie], (int32_t)(G.lwf * (double)G.fillpen[G.basewid[L.nwid[d]]]));
                    break;
                }
                /* the chain was emitted head first; the hypothesis is built back to front: reverse the chain in place */
                for (int32_t a = first, z = n - 1; a < z; a++, z--) {
                    int32_t x;
#define DG_SW(arr) x = arr[a]; arr[a] = arr[z]; arr[z] = x
                    DG_SW(ow); DG_SW(osf); DG_SW(oef); DG_SW(oas); DG_SW(ols);
#undef DG_SW
                }
                prev = n - 1;           /* the chain's head (p): its lscr comes from the next link */
            }
            code = hist;
            if (code == -3) break;          /* a link no path reaches: the reference's loop ends on its NULL history */
        }
        L.io[DG_IO_NWORDS] = n;
    }
#undef DG_EMIT
#undef DG_RLINK
}


/* ------------------------------------------------------------------ */
/* host side                                                           */
/* ------------------------------------------------------------------ */
__global__ void
k_dag_reset(DagShared G, const DagLane *__restrict__ lanes, int32_t use_active, const int32_t *__restrict__ lane_ids)
{
    const DagLane &L = lanes[lane_ids ? lane_ids[blockIdx.y] : (int32_t)blockIdx.y];
    if (use_active && !L.io[DG_IO_ACTIVE]) return;
    const int32_t stride = gridDim.x * blockDim.x;
    /* (use_active: the caller filled io[NENT]; else vithist_utt_end runs first in the pass and appends at most two entries) */
    const int32_t hmask = dg_eff_mask(G.h1mask, use_active ? L.io[DG_IO_NENT] : L.tab.st[0] + 2);
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i <= hmask; i += stride) {
        L.h1key[i] = 0ull; L.h2key[i] = 0ull; L.h2best[i] = 0ull; L.h1first[i] = INT_MAX; L.h1last[i] = -1; L.h1node[i] = -1;
    }
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i <= G.bmask; i += stride) { L.bkey[i] = 0ull; L.bbest[i] = 0ull; L.bdstar[i] = -1; }
}

struct s3a_dagpass_s {
    s3a_lm3g_t *lm;
    DagShared G;
    int32_t n_lanes, max_frames, device;
    std::vector<DagLane> lane;
    std::vector<int32_t *> arena, own_tab;      /* device allocations per lane */
    DagLane *d_lanes;
    int32_t *d_cfg;                              /* basewid | lwid | fillpen */
    uint8_t *d_fill;
    int32_t *h_io, *h_out;                       /* pinned: [n_lanes][DG_IO_N], [n_lanes][5 * hyp_cap] */
    int32_t n_run;
    bool lanes_dirty;
};

extern "C" void
s3a_dagpass_free(s3a_dagpass_t *dp)
{
    if (!dp) return;
    (void)hipSetDevice(dp->device);
    (void)hipDeviceSynchronize();
    for (auto p : dp->arena) if (p) (void)hipFree(p);
    for (auto p : dp->own_tab) if (p) (void)hipFree(p);
    if (dp->d_lanes) (void)hipFree(dp->d_lanes);
    if (dp->d_cfg) (void)hipFree(dp->d_cfg);
    if (dp->d_fill) (void)hipFree(dp->d_fill);
    if (dp->h_io) (void)hipHostFree(dp->h_io);
    if (dp->h_out) (void)hipHostFree(dp->h_out);
    delete dp;
}

extern "C" s3a_dagpass_t *
s3a_dagpass_init(s3a_lm3g_t *lm, const s3a_dag_cfg_t *cfg, int32_t n_lanes, int32_t max_entries, int32_t max_frames,
                 int32_t link_cap, int32_t pair_cap)
{
    if (!lm || !cfg || n_lanes <= 0 || max_entries <= 0 || max_frames <= 0 || !cfg->basewid || !cfg->is_filler || !cfg->lwid || !cfg->fillpen
        || cfg->n_word <= 0) {
        s3a_set_error("s3a_dagpass_init: bad arguments");
        return NULL;
    }
    if (!lm->d.ug_prob) { s3a_set_error("s3a_dagpass_init: the LM handle has no device arrays (s3a_lm3g_init_host)"); return NULL; }
    s3a_dagpass_t *dp = new s3a_dagpass_s();
    dp->lm = lm; dp->n_lanes = n_lanes; dp->max_frames = max_frames; dp->d_lanes = NULL; dp->d_cfg = NULL; dp->d_fill = NULL;
    dp->h_io = dp->h_out = NULL; dp->n_run = 0; dp->lanes_dirty = true; dp->device = 0;
    (void)hipGetDevice(&dp->device);
    DagShared &G = dp->G;
    memset(&G, 0, sizeof G);
    const int32_t E = max_entries + 4;
    G.n_word = cfg->n_word; G.F = max_frames + 3; G.E_cap = E;
    int32_t h = 1024; while (h < 2 * E) h <<= 1;
    G.h1mask = h - 1;
    h = 1024; while (h < 2 * (pair_cap > 0 ? pair_cap : (1 << 16))) h <<= 1;
    G.bmask = h - 1;
    /* the default follows -maxedge (the reference's 2 000 000) up to a memory budget of 2 M links (36 B each) per lane */
    G.link_cap = link_cap > 0 ? link_cap : (cfg->maxedge > 0 && cfg->maxedge < (1 << 21) ? cfg->maxedge : (1 << 21));
    G.task_cap = G.link_cap / 2 + (G.bmask + 1) / 2;
    G.hyp_cap = max_frames + 4 > 4096 ? max_frames + 4 : 4096;
    G.min_endfr = cfg->min_endfr; G.maxedge = cfg->maxedge; G.maxlmop = cfg->maxlmop; G.maxlpf = cfg->maxlpf;
    G.startwid = cfg->startwid; G.finishwid = cfg->finishwid; G.silwid = cfg->silwid; G.start_lwid = cfg->start_lwid;
    G.finish_lwid = cfg->finish_lwid; G.wip = cfg->wip; G.lwf = cfg->lwf;
    {
        const size_t nw = (size_t)cfg->n_word;
        if (hipMalloc((void **)&dp->d_cfg, nw * 12) != hipSuccess || hipMalloc((void **)&dp->d_fill, nw) != hipSuccess
            || hipMemcpy(dp->d_cfg, cfg->basewid, nw * 4, hipMemcpyHostToDevice) != hipSuccess
            || hipMemcpy(dp->d_cfg + nw, cfg->lwid, nw * 4, hipMemcpyHostToDevice) != hipSuccess
            || hipMemcpy(dp->d_cfg + 2 * nw, cfg->fillpen, nw * 4, hipMemcpyHostToDevice) != hipSuccess
            || hipMemcpy(dp->d_fill, cfg->is_filler, nw, hipMemcpyHostToDevice) != hipSuccess) {
            s3a_set_error("s3a_dagpass_init: out of device memory");
            s3a_dagpass_free(dp);
            return NULL;
        }
        G.basewid = dp->d_cfg; G.lwid = dp->d_cfg + nw; G.fillpen = dp->d_cfg + 2 * nw; G.is_filler = dp->d_fill;
    }
    dp->lane.resize(n_lanes);
    dp->arena.assign(n_lanes, NULL); dp->own_tab.assign(n_lanes, NULL);
    /* one arena of 32-bit words per lane, carved (64-bit arrays first: alignment) */
    const size_t H1 = (size_t)G.h1mask + 1, BH = (size_t)G.bmask + 1, Fw = (size_t)G.F + 2, Nw = (size_t)E + 2;
    const size_t words = 2 * (3 * H1 + 2 * BH) + 3 * H1 + 9 * (size_t)E + 10 * Fw + 15 * Nw + 3 * (size_t)G.link_cap + 7 * BH
        + 2 * (size_t)G.task_cap + DG_IO_N + 2 * (size_t)G.hyp_cap + 5 * (size_t)G.hyp_cap + 64;
    for (int32_t z = 0; z < n_lanes; z++) {
        int32_t *a = NULL;
        if (hipMalloc((void **)&a, words * 4) != hipSuccess) {
            s3a_set_error("s3a_dagpass_init: out of device memory (%zu MB per lane)", words * 4 >> 20);
            s3a_dagpass_free(dp);
            return NULL;
        }
        dp->arena[z] = a;
        DagLane &L = dp->lane[z];
        memset((void *)&L, 0, sizeof L);
        unsigned long long *q = (unsigned long long *)a;
        L.h1key = q; q += H1; L.h2key = q; q += H1; L.h2best = q; q += H1; L.bkey = q; q += BH; L.bbest = q; q += BH;
        int32_t *w = (int32_t *)q;
#define CARVE(field, n) do { L.field = w; w += (n); } while (0)
        CARVE(h1first, H1); CARVE(h1last, H1); CARVE(h1node, H1);
        CARVE(sfp, E); CARVE(efp, E); CARVE(eslot, E); CARVE(enode, E); CARVE(eapos, E); CARVE(ehk, E); CARVE(knode, E); CARVE(hkent, E); CARVE(aent, E);
        CARVE(ncnt, Fw); CARVE(nbase, Fw); CARVE(nfill, Fw); CARVE(kcnt, Fw); CARVE(kbase, Fw); CARVE(acnt, Fw); CARVE(abase, Fw); CARVE(afill, Fw);
        CARVE(lcnt, Fw); CARVE(loff, Fw);
        CARVE(nfirst, Nw); CARVE(nwid, Nw); CARVE(nsf, Nw); CARVE(nfef, Nw); CARVE(nlef, Nw); CARVE(nkeep, Nw); CARVE(nfil, Nw); CARVE(nhk, Nw);
        CARVE(hkbase, Nw); CARVE(nkpos, Nw); CARVE(phead, Nw); CARVE(shead, Nw); CARVE(reach, Nw); CARVE(bpcnt, Nw); CARVE(bpbase, Nw);
        CARVE(lpscr, G.link_cap); CARVE(lhist, G.link_cap); CARVE(llscr, G.link_cap);
        CARVE(bdstar, BH); CARVE(bnextp, BH); CARVE(bnexts, BH); CARVE(bpscr, BH); CARVE(bhist, BH); CARVE(blscr, BH); CARVE(bplist, BH);
        CARVE(task, 2 * (size_t)G.task_cap);
        CARVE(io, DG_IO_N); CARVE(hyp_wid, G.hyp_cap); CARVE(hyp_sf, G.hyp_cap); CARVE(out, 5 * (size_t)G.hyp_cap);
#undef CARVE
        if (hipMemset(L.io, 0, DG_IO_N * 4) != hipSuccess) { s3a_dagpass_free(dp); return NULL; }
    }
    if (hipMalloc((void **)&dp->d_lanes, sizeof(DagLane) * n_lanes) != hipSuccess
        || hipHostMalloc((void **)&dp->h_io, (size_t)n_lanes * DG_IO_N * 4) != hipSuccess
        || hipHostMalloc((void **)&dp->h_out, (size_t)n_lanes * 5 * G.hyp_cap * 4) != hipSuccess) {
        s3a_set_error("s3a_dagpass_init: allocation failed");
        s3a_dagpass_free(dp);
        return NULL;
    }
    return dp;
}

int32_t
s3a_dagpass_bind(s3a_dagpass_t *dp, int32_t lane, const DagTab &tab)
{
    if (!dp || lane < 0 || lane >= dp->n_lanes) return S3A_EINVAL;
    dp->lane[lane].tab = tab;
    dp->lanes_dirty = true;
    return S3A_OK;
}

int32_t
s3a_dagpass_enqueue(s3a_dagpass_t *dp, int32_t n, hipStream_t st, int32_t do_utt_end)
{
    if (!dp || n <= 0 || n > dp->n_lanes) return S3A_EINVAL;
    if (dp->lanes_dirty) {
        HIPCHK(hipMemcpyAsync(dp->d_lanes, dp->lane.data(), sizeof(DagLane) * dp->n_lanes, hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));       /* (the source vector may change after this call) */
        dp->lanes_dirty = false;
    }
    hipLaunchKernelGGL(k_dag_reset, dim3(64, n), dim3(256), 0, st, dp->G, dp->d_lanes, do_utt_end ? 0 : 1, (const int32_t *)NULL);
    hipLaunchKernelGGL(k_dag_pass, dim3(n), dim3(DG_T), 0, st, dp->G, dp->d_lanes, dp->lm->d, do_utt_end, do_utt_end ? 0 : 1, (const int32_t *)NULL);
    HIPCHK(hipGetLastError());
    dp->n_run = n;
    return S3A_OK;
}

/* a queue with lane refill (s3a_uttdec_decode_queue): the lane descriptors go up once before the first frame ... */
int32_t
s3a_dagpass_prepare(s3a_dagpass_t *dp, hipStream_t st)
{
    if (!dp) return S3A_EINVAL;
    if (dp->lanes_dirty) {
        HIPCHK(hipMemcpyAsync(dp->d_lanes, dp->lane.data(), sizeof(DagLane) * dp->n_lanes, hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
        dp->lanes_dirty = false;
    }
    dp->n_run = 0;
    return S3A_OK;
}

/* ... and at a refill event the pass (vithist_utt_end included) runs for the n lanes listed in lane_ids_dev, behind their last
 * frame and before their tables are reused; nothing is fetched: the caller's kernel copies what it wants out of the lanes */
int32_t
s3a_dagpass_enqueue_lanes(s3a_dagpass_t *dp, const int32_t *lane_ids_dev, int32_t n, hipStream_t st)
{
    if (!dp || !lane_ids_dev || n <= 0 || n > dp->n_lanes || dp->lanes_dirty) return S3A_EINVAL;
    hipLaunchKernelGGL(k_dag_reset, dim3(64, n), dim3(256), 0, st, dp->G, dp->d_lanes, 0, lane_ids_dev);
    hipLaunchKernelGGL(k_dag_pass, dim3(n), dim3(DG_T), 0, st, dp->G, dp->d_lanes, dp->lm->d, 1, 0, lane_ids_dev);
    HIPCHK(hipGetLastError());
    return S3A_OK;
}

const DagLane *s3a_dagpass_dev_lanes(const s3a_dagpass_t *dp) { return dp ? dp->d_lanes : NULL; }
int32_t s3a_dagpass_hyp_cap(const s3a_dagpass_t *dp) { return dp ? dp->G.hyp_cap : 0; }

int32_t
s3a_dagpass_fetch(s3a_dagpass_t *dp, int32_t n, hipStream_t st)
{
    if (!dp || n <= 0 || n > dp->n_lanes) return S3A_EINVAL;
    for (int32_t z = 0; z < n; z++) {
        HIPCHK(hipMemcpyAsync(dp->h_io + (size_t)z * DG_IO_N, dp->lane[z].io, DG_IO_N * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(dp->h_out + (size_t)z * 5 * dp->G.hyp_cap, dp->lane[z].out, (size_t)5 * dp->G.hyp_cap * 4, hipMemcpyDeviceToHost, st));
    }
    HIPCHK(hipStreamSynchronize(st));
    return S3A_OK;
}

extern "C" int32_t
s3a_dagpass_result(const s3a_dagpass_t *dp, int32_t lane, s3a_dag_result_t *out)
{
    if (!dp || !out || lane < 0 || lane >= dp->n_run) return S3A_EINVAL;
    const int32_t *io = dp->h_io + (size_t)lane * DG_IO_N;
    int32_t *o = dp->h_out + (size_t)lane * 5 * dp->G.hyp_cap;
    const int32_t cap = dp->G.hyp_cap;
    memset(out, 0, sizeof *out);
    out->status = io[DG_IO_STATUS]; out->n_words = io[DG_IO_STATUS] == 0 ? io[DG_IO_NWORDS] : 0; out->n_node = io[DG_IO_NNODE];
    out->n_link = io[DG_IO_NLINK]; out->n_bypass = io[DG_IO_NBYPASS]; out->lmop = io[DG_IO_LMOP]; out->score = io[DG_IO_SCORE];
    out->first_pass_score = io[DG_IO_FIRSTSCORE]; out->n_entry = io[DG_IO_NENT]; out->endid = io[DG_IO_ENDID];
    out->wid = o; out->sf = o + cap; out->ef = o + 2 * cap; out->ascr = o + 3 * cap; out->lscr = o + 4 * cap;
    return S3A_OK;
}

/* the kernel leaves the words end first (dag_backtrace prepends): utterance order for the caller */
static void
dag_reverse_out(s3a_dagpass_t *dp, int32_t n)
{
    const int32_t cap = dp->G.hyp_cap;
    for (int32_t z = 0; z < n; z++) {
        const int32_t *io = dp->h_io + (size_t)z * DG_IO_N;
        if (io[DG_IO_STATUS] != 0) continue;
        int32_t *o = dp->h_out + (size_t)z * 5 * cap;
        for (int k = 0; k < 5; k++)
            for (int32_t a = 0, b = io[DG_IO_NWORDS] - 1; a < b; a++, b--) { const int32_t x = o[k * cap + a]; o[k * cap + a] = o[k * cap + b]; o[k * cap + b] = x; }
    }
}

int32_t
s3a_dagpass_finish(s3a_dagpass_t *dp, int32_t n, hipStream_t st)
{
    const int32_t rc = s3a_dagpass_fetch(dp, n, st);
    if (rc != S3A_OK) return rc;
    dag_reverse_out(dp, n);
    return S3A_OK;
}

/*
 * The lattice the pass built for `lane` as vithist_dag_build (vithist.c:1100-1311) leaves it, in the orders of the
 * reference's lists -- what dag_write / dag_write_htk (dag.c:731-897) print and what a dag_t can be rebuilt from:
 *   nodes  in dag->list order (the LAST node made comes first: alloc_next is a head insertion, :1249-1251) = NODEID order
 *          of the Sphinx-3 file;
 *   links  grouped by source node in that order and, per source, in succlist order: dag_link (dag.c:186-238) prepends, the
 *          links of a node are made exit by exit (the hook list: latest end frame first) and per exit over the kept nodes
 *          of the next frame in sfwid order, so the list reads end frames ascending and, per end frame, the frame's nodes in
 *          reverse list order.  An exit with a positive acoustic score makes no link (dag_link refuses, :193-195).
 * The predecessor list of a node is its incoming links by source id ASCENDING in this numbering (all made in source order).
 * Available after any pass that got as far as the link count (status 0, DG_E_NOPATH, DG_E_POSEDGE, DG_E_CAP from links /
 * bypass); host side: three small copies + loops.  nodes == NULL or links == NULL: sizes only.
 */
extern "C" int32_t
s3a_dagpass_lattice(s3a_dagpass_t *dp, int32_t lane, s3a_lat_info_t *info, s3a_lat_node_t *nodes, int32_t node_cap,
                    s3a_lat_link_t *links, int32_t link_cap)
{
    if (!dp || !info || lane < 0 || lane >= dp->n_run) return S3A_EINVAL;
    return s3a_dagpass_lattice_lane(dp, lane, info, nodes, node_cap, links, link_cap);
}

/* the same for any lane whose pass has run and whose stream is synchronised (the engine's queue: a group's lanes, before the next group
 * takes them) */
int32_t
s3a_dagpass_lattice_lane(s3a_dagpass_t *dp, int32_t lane, s3a_lat_info_t *info, s3a_lat_node_t *nodes, int32_t node_cap,
                         s3a_lat_link_t *links, int32_t link_cap)
{
    if (!dp || !info || lane < 0 || lane >= dp->n_lanes) return S3A_EINVAL;
    HIPCHK(hipSetDevice(dp->device));
    const DagLane &L = dp->lane[lane];
    int32_t io[DG_IO_N], st2[2];
    HIPCHK(hipMemcpy(io, L.io, sizeof io, hipMemcpyDeviceToHost));
    memset(info, 0, sizeof *info);
    info->status = io[DG_IO_STATUS];
    const int32_t NK = io[DG_IO_NNODE], E = io[DG_IO_NENT], endid = io[DG_IO_ENDID];
    if (NK <= 0 || E <= 0 || endid < 0) { s3a_set_error("s3a_dagpass_lattice: lane %d has no lattice (pass status %d)", lane, io[DG_IO_STATUS]); return S3A_EUNSUP; }
    HIPCHK(hipMemcpy(st2, L.tab.st, 8, hipMemcpyDeviceToHost));
    const int32_t n_frm = st2[1], F1 = n_frm + 1;
    std::vector<int32_t> kcnt(F1 + 1), kbase(F1 + 1), nbase(F1 + 1), knode(NK);
    HIPCHK(hipMemcpy(kcnt.data(), L.kcnt, (size_t)F1 * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(kbase.data(), L.kbase, (size_t)(F1 + 1) * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(nbase.data(), L.nbase, (size_t)(F1 + 1) * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(knode.data(), L.knode, (size_t)NK * 4, hipMemcpyDeviceToHost));
    const int32_t NN = nbase[F1];
    if (NN <= 0 || NN > dp->G.E_cap || kbase[F1] != NK) { s3a_set_error("s3a_dagpass_lattice: inconsistent node counts"); return S3A_EUNSUP; }
    std::vector<int32_t> nfirst(NN), nsf(NN), nfef(NN), nlef(NN), nhk(NN), hkbase(NN), nkpos(NN), hkent(E), efp(E), eapos(E), enode(E),
        wid(E), ascr(E), lscr(E);
#define GET(v, src, n) HIPCHK(hipMemcpy((v).data(), (src), (size_t)(n) * 4, hipMemcpyDeviceToHost))
    GET(nfirst, L.nfirst, NN); GET(nsf, L.nsf, NN); GET(nfef, L.nfef, NN); GET(nlef, L.nlef, NN); GET(nhk, L.nhk, NN); GET(hkbase, L.hkbase, NN);
    GET(nkpos, L.nkpos, NN); GET(hkent, L.hkent, E); GET(efp, L.efp, E); GET(eapos, L.eapos, E); GET(enode, L.enode, E);
    GET(wid, L.tab.wid, E); GET(ascr, L.tab.ascr, E); GET(lscr, L.tab.lscr, E);
#undef GET
    auto file_id = [&](int32_t k) { return NK - 1 - k; };            /* k = rank among the kept nodes in (start frame, list) order */
    auto rank_of = [&](int32_t p) { return kbase[nsf[p]] + nkpos[p]; };
    /* the links, counted first */
    long long n_links = 0;
    for (int32_t k = NK - 1; k >= 0; k--) {
        const int32_t p = knode[k];
        for (int32_t h = 0; h < nhk[p]; h++) { const int32_t ie = hkent[hkbase[p] + h]; if (eapos[ie] >= 0) n_links += kcnt[efp[ie] + 1]; }
    }
    const int32_t endn = enode[endid], rootn = knode[0];
    info->n_frames = n_frm; info->n_nodes = NK; info->n_links = (int32_t)n_links;
    info->initial = file_id(rank_of(rootn)); info->final = file_id(rank_of(endn)); info->final_ascr = 0;
    for (int32_t h = 0; h < nhk[endn]; h++) { const int32_t ie = hkent[hkbase[endn] + h]; if (efp[ie] == n_frm) info->final_ascr = ascr[ie]; }
    if (!nodes || !links) return S3A_OK;
    if (node_cap < NK || link_cap < n_links) { s3a_set_error("s3a_dagpass_lattice: %d nodes / %lld links exceed the caller's arrays", NK, n_links); return S3A_ENOMEM; }
    for (int32_t k = NK - 1; k >= 0; k--) {
        const int32_t p = knode[k], i0 = nfirst[p];
        s3a_lat_node_t &n = nodes[file_id(k)];
        n.wid = wid[i0]; n.sf = nsf[p]; n.fef = nfef[p]; n.lef = nlef[p]; n.ascr = ascr[i0]; n.lscr = lscr[i0];
    }
    int32_t q = 0;
    std::vector<int32_t> ex;
    for (int32_t k = NK - 1; k >= 0; k--) {
        const int32_t p = knode[k];
        ex.clear();
        for (int32_t h = 0; h < nhk[p]; h++) { const int32_t ie = hkent[hkbase[p] + h]; if (eapos[ie] >= 0) ex.push_back(ie); }
        for (size_t a = 1; a < ex.size(); a++) {                   /* by end frame, ascending (a node has one exit per end frame) */
            const int32_t x = ex[a]; size_t b = a;
            while (b > 0 && efp[ex[b - 1]] > efp[x]) { ex[b] = ex[b - 1]; b--; }
            ex[b] = x;
        }
        for (int32_t ie : ex) {
            const int32_t f2 = efp[ie] + 1;
            for (int32_t yy = kcnt[f2] - 1; yy >= 0; yy--) {
                s3a_lat_link_t &l = links[q++];
                l.from = file_id(k); l.to = file_id(kbase[f2] + yy); l.ascr = ascr[ie]; l.lscr = lscr[ie]; l.ef = f2 - 1;
            }
        }
    }
    return S3A_OK;
}

/* host tables -> the pass's own device tables (parity tests against the oracle's restatement) */
extern "C" int32_t
s3a_dagpass_run_tables(s3a_dagpass_t *dp, int32_t n_utt, const s3a_dag_table_t *tabs)
{
    if (!dp || !tabs || n_utt <= 0 || n_utt > dp->n_lanes) return S3A_EINVAL;
    HIPCHK(hipSetDevice(dp->device));
    const int32_t E = dp->G.E_cap, F = dp->G.F + 2;
    for (int32_t z = 0; z < n_utt; z++) {
        const s3a_dag_table_t &t = tabs[z];
        if (t.n_entry <= 0 || t.n_entry > E || t.n_frm <= 0 || t.n_frm + 2 > dp->G.F || t.n_hyp < 0 || t.n_hyp > dp->G.hyp_cap || t.endid < 0 || t.endid >= t.n_entry) {
            s3a_set_error("s3a_dagpass_run_tables: table %d outside the pass's capacities", z);
            return S3A_EINVAL;
        }
        if (!dp->own_tab[z]) {
            if (hipMalloc((void **)&dp->own_tab[z], ((size_t)10 * E + F + 16) * 4) != hipSuccess) { s3a_set_error("s3a_dagpass_run_tables: out of device memory"); return S3A_ENOMEM; }
            int32_t *w = dp->own_tab[z];
            DagTab tb;
            tb.score = w; tb.pred = w + E; tb.lw0 = w + 2 * (size_t)E; tb.lw1 = w + 3 * (size_t)E; tb.wid = w + 4 * (size_t)E; tb.sf = w + 5 * (size_t)E;
            tb.ef = w + 6 * (size_t)E; tb.ascr = w + 7 * (size_t)E; tb.lscr = w + 8 * (size_t)E; tb.type = w + 9 * (size_t)E;
            tb.frame_start = w + 10 * (size_t)E; tb.st = w + 10 * (size_t)E + F; tb.cap = E;
            s3a_dagpass_bind(dp, z, tb);
        }
        const DagTab &tb = dp->lane[z].tab;
        const size_t nb = (size_t)t.n_entry * 4;
        int32_t st2[2] = { t.n_entry, t.n_frm };
        int32_t io[DG_IO_N];
        memset(io, 0, sizeof io);
        io[DG_IO_ACTIVE] = 1; io[DG_IO_NENT] = t.n_entry; io[DG_IO_ENDID] = t.endid; io[DG_IO_NHYP] = t.n_hyp;
        HIPCHK(hipMemcpy(tb.wid, t.wid, nb, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(tb.sf, t.sf, nb, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(tb.ef, t.ef, nb, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(tb.ascr, t.ascr, nb, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(tb.lscr, t.lscr, nb, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(tb.score, t.score, nb, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(tb.st, st2, 8, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(dp->lane[z].io, io, sizeof io, hipMemcpyHostToDevice));
        if (t.n_hyp) {
            HIPCHK(hipMemcpy(dp->lane[z].hyp_wid, t.hyp_wid, (size_t)t.n_hyp * 4, hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(dp->lane[z].hyp_sf, t.hyp_sf, (size_t)t.n_hyp * 4, hipMemcpyHostToDevice));
        }
    }
    int32_t rc = s3a_dagpass_enqueue(dp, n_utt, 0, 0);
    if (rc != S3A_OK) return rc;
    return s3a_dagpass_finish(dp, n_utt, 0);
}
