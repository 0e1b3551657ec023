/*
 * s3a_lextree.hip -- the per-frame lexical-tree search operations on the device.
 *
 * Replaces, for a SET of lextrees searched in lock step (sphinx3 mode 4 keeps
 * 2 x -Nlextree of them: unigram trees + filler trees),
 *   sphinx3/src/libs3decoder/libsearch/lextree.c:1093-1236  lextree_enter
 *   lextree.c:1240-1249  lextree_active_swap        lextree.c:936-961  lextree_utt_end
 *   lextree.c:1253-1310  lextree_hmm_eval  (-> hmm_vit_eval_3st_lr, libam/hmm.c:592-674)
 *   lextree.c:1365-1597  lextree_hmm_propagate_non_leaves   (composite triphones, -pheurtype 0)
 *   lextree.c:1600-1663  lextree_hmm_propagate_leaves
 *   lextree.c:910-932    lextree_ssid_active + libam/mdef.c:857-869 mdef_sseq2sen_active
 *                        + libsearch/dict2pid.c:1055-1075 dict2pid_comsseq2sen_active
 *
 * The trees are static (kbcore.c:626 hard-wires composite triphones, so no
 * cross-word nodes are grown during search) and arrive FLATTENED: all nodes of
 * all trees in one index space, CSR child lists in the reference's glist order,
 * CSR parent lists derived here.  HMM state is structure-of-arrays per node.
 *
 * ORDER IS SEMANTICS.  The reference walks the active list sequentially; the
 * position of a node in that list decides (a) which parent's history survives an
 * exact tie, (b) whether a child below the beam is cleared before or after a
 * parent re-enters it (which changes its non-entry states), and (c) the order of
 * the next active list and hence of word exits reaching vithist.  The kernels
 * reproduce the sequential result exactly, in parallel, from the node's point of
 * view: for node v at list position j with candidate parents at positions E (< j)
 * and L (> j),
 *     entered_early  = max_E(ns) > in0            -> keeps its states, never cleared
 *     else survive/clear by bestscore >= th at its own turn
 *     then L may raise in_score again (strict >, earliest position wins ties)
 * and v is appended to the next list at exactly one "turn": the first parent
 * that entered it while it was not yet in the list, or its own turn if it
 * survived.  Per-turn counts are prefix-summed and every turn writes its nodes
 * in child-list order, which yields the reference's next_active order verbatim.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <vector>
#include <algorithm>
#include <map>

#include "s3a_device.h"
#include "s3a_structs.h"
#include "s3a_vit.h"
#include "s3a_scan.h"

#define WORST S3A_WORST
#define LT_BLOCK 256

/* ------------------------------------------------------------------ */
/* helpers                                                             */
/* ------------------------------------------------------------------ */
/* ------------------------------------------------------------------ */
/* lextree_hmm_eval                                                    */
/* ------------------------------------------------------------------ */
__global__ void __launch_bounds__(LT_BLOCK)
k_lt_hmm_eval(const int32_t *__restrict__ node_base, const int32_t *__restrict__ act,
              const int32_t *__restrict__ nact, int32_t N, int32_t n_tmat,
              const int32_t *__restrict__ ssid, const int32_t *__restrict__ tmatid,
              const int32_t *__restrict__ wid, const uint8_t *__restrict__ comp,
              const int32_t *__restrict__ tp_g, const int16_t *__restrict__ sseq,
              const int16_t *__restrict__ comsseq, const int32_t *__restrict__ senscr,
              const int32_t *__restrict__ comsen, int32_t *sc, int32_t *hist, int32_t *outs,
              int32_t *outh, int32_t *bests, int32_t *best_out)
{
    extern __shared__ int32_t tp_s[];
    __shared__ int32_t red[2][LT_BLOCK / 64];
    for (int32_t i = threadIdx.x; i < n_tmat * 12; i += LT_BLOCK)
        tp_s[i] = tp_g[i];
    __syncthreads();
    const int32_t t = blockIdx.y;
    const int32_t i = blockIdx.x * LT_BLOCK + threadIdx.x;
    int32_t best = INT_MIN, wbest = INT_MIN;
    if (i < nact[t]) {
        const int32_t v = act[node_base[t] + i];
        HmmRegsT<int32_t> r;
        const int32_t ss = ssid[v];
        int32_t e0, e1, e2;
        if (comp[v]) {
            e0 = comsen[comsseq[ss * 3 + 0]]; e1 = comsen[comsseq[ss * 3 + 1]]; e2 = comsen[comsseq[ss * 3 + 2]];
        }
        else {
            e0 = senscr[sseq[ss * 3 + 0]]; e1 = senscr[sseq[ss * 3 + 1]]; e2 = senscr[sseq[ss * 3 + 2]];
        }
#pragma unroll
        for (int st = 0; st < 3; st++) { r.s[st] = sc[NSI(st, N, v)]; r.h[st] = hist[NSI(st, N, v)]; }
        r.out = outs[NSV(v)];
        r.outh = outh[NSV(v)];
        int32_t k = vit3(r, tp_s + tmatid[v] * 12, e0, e1, e2);
#pragma unroll
        for (int st = 0; st < 3; st++) { sc[NSI(st, N, v)] = r.s[st]; hist[NSI(st, N, v)] = r.h[st]; }
        outs[NSV(v)] = r.out;
        outh[NSV(v)] = r.outh;
        bests[NSV(v)] = k;
        best = k;
        if (wid[v] >= 0) wbest = k;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        best = max(best, __shfl_xor(best, o, 64));
        wbest = max(wbest, __shfl_xor(wbest, o, 64));
    }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = best; red[1][threadIdx.x >> 6] = wbest; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < LT_BLOCK / 64; w++) { best = max(best, red[0][w]); wbest = max(wbest, red[1][w]); }
        best = max(best, red[0][0]); wbest = max(wbest, red[1][0]);
        if (best != INT_MIN) atomicMax(&best_out[2 * t], best);
        if (wbest != INT_MIN) atomicMax(&best_out[2 * t + 1], wbest);
    }
}

/* ------------------------------------------------------------------ */
/* lextree_hmm_propagate_non_leaves, three phases                      */
/* ------------------------------------------------------------------ */
/* phase A: every propagating parent nominates its INACTIVE children once */
__global__ void __launch_bounds__(LT_BLOCK)
k_lt_prop_mark(const int32_t *__restrict__ node_base, const int32_t *__restrict__ act,
               const int32_t *__restrict__ nact, int32_t cf, const int32_t *__restrict__ thr,
               const int32_t *__restrict__ wid, const int32_t *__restrict__ prob,
               const int32_t *__restrict__ child_off, const int32_t *__restrict__ child,
               const int32_t *__restrict__ outs, const int32_t *__restrict__ posf,
               int32_t *candf, int32_t *cand, int32_t *ncand)
{
    const int32_t t = blockIdx.y, i = blockIdx.x * LT_BLOCK + threadIdx.x;
    const int32_t th = thr[0], pth = thr[1];
    if (i >= nact[t]) return;
    const int32_t u = act[node_base[t] + i];
    if (wid[u] >= 0 || outs[NSV(u)] < pth) return;
    for (int32_t j = child_off[u]; j < child_off[u + 1]; j++) {
        const int32_t c = child[j];
        const int32_t ns = add32(outs[NSV(u)], add32(prob[c], -prob[u]));
        if (ns >= th && posf[PPX(c)] != cf && atomicExch(&candf[c], cf) != cf)
            cand[node_base[t] + atomicAdd(&ncand[t], 1)] = c;
    }
}

/* phase B: resolve every active node and every candidate from its own point of view */
__global__ void __launch_bounds__(LT_BLOCK)
k_lt_prop_resolve(const int32_t *__restrict__ node_base, const int32_t *__restrict__ act,
                  const int32_t *__restrict__ nact, const int32_t *__restrict__ cand,
                  const int32_t *__restrict__ ncand, int32_t N, int32_t cf,
                  const int32_t *__restrict__ thr,
                  const int32_t *__restrict__ wid, const int32_t *__restrict__ prob,
                  const int32_t *__restrict__ par_off, const int32_t *__restrict__ par,
                  const int32_t *__restrict__ pos, const int32_t *__restrict__ posf,
                  int32_t *sc, int32_t *hist, int32_t *outs, int32_t *outh, int32_t *bests,
                  int32_t *frame, int32_t *turn, int32_t *selfemit, int32_t *cnt)
{
    const int32_t t = blockIdx.y, i = blockIdx.x * LT_BLOCK + threadIdx.x;
    const int32_t na = nact[t], nc = ncand[t];
    const int32_t th = thr[0], pth = thr[1];
    if (i >= na + nc) return;
    const bool is_active = i < na;
    const int32_t v = is_active ? act[node_base[t] + i] : cand[node_base[t] + (i - na)];
    const int32_t j = is_active ? i : INT_MAX;      /* own turn; candidates have none */
    const int32_t nf = cf + 1;
    const int32_t in0 = sc[NSV(v)];                      /* state 0 score */

    /* scan the parents once: maxima and earliest positions for the early and late sets */
    int32_t mE = INT_MIN, pE = INT_MAX, hE = -1, firstE = INT_MAX;
    int32_t mL = INT_MIN, pL = INT_MAX, hL = -1, firstL = INT_MAX;
    for (int32_t k = par_off[v]; k < par_off[v + 1]; k++) {
        const int32_t p = par[k];
        if (posf[PPX(p)] != cf) continue;                /* parent not in this frame's list */
        const int32_t po = outs[NSV(p)];
        if (po < pth) continue;
        const int32_t ns = add32(po, add32(prob[v], -prob[p]));
        if (ns < th) continue;
        const int32_t pp = pos[PPX(p)];
        if (pp < j) {
            if (ns > mE || (ns == mE && pp < pE)) { mE = ns; pE = pp; hE = outh[NSV(p)]; }
            if (ns > in0 && pp < firstE) firstE = pp;
        }
        else {
            if (ns > mL || (ns == mL && pp < pL)) { mL = ns; pL = pp; hL = outh[NSV(p)]; }
            if (pp < firstL) firstL = pp;
        }
    }
    int32_t cur = in0, h0 = hist[NSV(v)], my_turn = -1;
    bool in_list = false, cleared = false, entered = false;
    if (mE > in0) {                                 /* entered before its own turn */
        cur = mE; h0 = hE; entered = true; in_list = true; my_turn = firstE;
    }
    else if (is_active) {
        if (bests[NSV(v)] >= th) { in_list = true; selfemit[node_base[t] + i] = 1; atomicAdd(&cnt[node_base[t] + i], 1); }
        else { cleared = true; cur = WORST; h0 = -1; }
    }
    if (mL > cur) {                                 /* (re-)entered after its own turn */
        cur = mL; h0 = hL; entered = true;
        if (!in_list) {
            /* first late parent whose score beats what the node holds at that moment:
             * after a clear (or for an inactive node) that is WORST, so the earliest one */
            in_list = true; my_turn = firstL;
        }
    }
    if (cleared) {
        sc[NSI(1, N, v)] = WORST; sc[NSI(2, N, v)] = WORST;
        hist[NSI(1, N, v)] = -1; hist[NSI(2, N, v)] = -1;
        outs[NSV(v)] = WORST; outh[NSV(v)] = -1; bests[NSV(v)] = WORST;
    }
    if (cleared || entered) { sc[NSV(v)] = cur; hist[NSV(v)] = h0; }
    frame[NSV(v)] = in_list ? nf : (cleared ? -1 : frame[NSV(v)]);
    if (my_turn >= 0) { turn[v] = my_turn; atomicAdd(&cnt[node_base[t] + my_turn], 1); }
}

/* phase C: one workgroup per tree: scan the per-turn counts, then every turn writes
 * itself (if it survived) followed by the children it appended, in child-list order */
__global__ void __launch_bounds__(SCAN_THREADS)
k_lt_prop_emit(const int32_t *__restrict__ node_base, const int32_t *__restrict__ act,
               const int32_t *__restrict__ nact, int32_t cf,
               const int32_t *__restrict__ child_off, const int32_t *__restrict__ child,
               int32_t *turn, int32_t *selfemit, int32_t *cnt, int32_t *nxt, int32_t *nnxt,
               int32_t *pos, int32_t *posf)
{
    __shared__ int32_t total;
    const int32_t t = blockIdx.x, b = node_base[t], na = nact[t], nf = cf + 1;
    block_exclusive_scan(cnt + b, na, &total);
    __syncthreads();
    for (int32_t i = threadIdx.x; i < na; i += SCAN_THREADS) {
        const int32_t u = act[b + i];
        int32_t k = cnt[b + i];
        if (selfemit[b + i]) {
            nxt[b + k] = u; PP_SET(pos, u, k, nf); k++;
            selfemit[b + i] = 0;
        }
        for (int32_t j = child_off[u]; j < child_off[u + 1]; j++) {
            const int32_t c = child[j];
            if (turn[c] == i) {
                nxt[b + k] = c; PP_SET(pos, c, k, nf); k++;
                turn[c] = -1;
            }
        }
    }
    __syncthreads();
    for (int32_t i = threadIdx.x; i < na; i += SCAN_THREADS)
        cnt[b + i] = 0;
    if (threadIdx.x == 0) nnxt[t] = total;
}

/* ------------------------------------------------------------------ */
/* lextree_hmm_propagate_leaves: ordered compaction of word exits      */
/* ------------------------------------------------------------------ */
__global__ void __launch_bounds__(SCAN_THREADS)
k_lt_leaves(const int32_t *__restrict__ node_base, const int32_t *__restrict__ act,
            const int32_t *__restrict__ nact, int32_t N, const int32_t *__restrict__ thr,
            const int32_t *__restrict__ wid, const int32_t *__restrict__ prob,
            const int32_t *__restrict__ outs, const int32_t *__restrict__ outh,
            int32_t *flag, int32_t *exits, int32_t *nexit, int32_t n_tree)
{
    __shared__ int32_t total;
    const int32_t t = blockIdx.x, b = node_base[t], na = nact[t];
    const int32_t wth = thr[2];
    for (int32_t i = threadIdx.x; i < na; i += SCAN_THREADS) {
        const int32_t u = act[b + i];
        flag[b + i] = (wid[u] >= 0 && outs[NSV(u)] >= wth) ? 1 : 0;
    }
    __syncthreads();
    block_exclusive_scan(flag + b, na, &total);
    __syncthreads();
    for (int32_t i = threadIdx.x; i < na; i += SCAN_THREADS) {
        const int32_t u = act[b + i];
        if (wid[u] >= 0 && outs[NSV(u)] >= wth) {
            const int32_t k = b + flag[b + i];
            exits[k] = wid[u];
            exits[N + k] = add32(outs[NSV(u)], -prob[u]);
            exits[2 * N + k] = outh[NSV(u)];
            if (outh[NSV(u)] == -1) atomicExch(&nexit[n_tree + t], 1);   /* "out.history==-1, error" */
        }
    }
    __syncthreads();
    for (int32_t i = threadIdx.x; i < na; i += SCAN_THREADS)
        flag[b + i] = 0;
    if (threadIdx.x == 0) nexit[t] = total;
}

/* srch_TST_hmm_compute_lv2's threshold arithmetic (srch_time_switch_tree.c:826-905) on the
 * device, so that propagation can start without a host round trip:
 * thr = {thres, phone_thres, word_thres, besthmmscr, bestwordscr, frm_nhmm, need_histprune} */
__global__ void
k_lt_thresholds(const int32_t *__restrict__ best, const int32_t *__restrict__ nact, int32_t n_tree,
                int32_t hmmbeam, int32_t pbeam, int32_t wbeam, int32_t phone_uses_wbeam,
                int32_t maxhmmpf, int32_t *thr)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int32_t bh = INT_MIN, bw = INT_MIN, n = 0;
    for (int32_t t = 0; t < n_tree; t++) {
        bh = max(bh, best[2 * t]);
        bw = max(bw, best[2 * t + 1]);
        n += nact[t];
    }
    thr[0] = add32(bh, hmmbeam);
    thr[2] = add32(bw, wbeam);
    thr[1] = phone_uses_wbeam ? thr[2] : add32(bh, pbeam);
    thr[3] = bh; thr[4] = bw; thr[5] = n;
    thr[6] = (n > maxhmmpf + (maxhmmpf >> 1)) ? 1 : 0;
}

__global__ void
k_lt_set_thr(int32_t *thr, int32_t th, int32_t pth, int32_t wth)
{
    if (threadIdx.x == 0) { thr[0] = th; thr[1] = pth; thr[2] = wth; }
}

/* gather everything the host needs from one frame into one contiguous record:
 * [best,wbest] x T | nact x T | thr[8] | n_exit x T | err x T | extra[8] | exits (wid,score,hist) */
__global__ void __launch_bounds__(SCAN_THREADS)
k_lt_pack(const int32_t *__restrict__ node_base, int32_t n_tree, int32_t N,
          const int32_t *__restrict__ best, const int32_t *__restrict__ nact,
          const int32_t *__restrict__ thr, const int32_t *__restrict__ nexit,
          const int32_t *__restrict__ exits, const int32_t *__restrict__ extra, int32_t *pack,
          int32_t max_exits)
{
    const int32_t T = n_tree, hdr = 5 * T + 16;
    for (int32_t i = threadIdx.x; i < 2 * T; i += SCAN_THREADS) pack[i] = best[i];
    for (int32_t i = threadIdx.x; i < T; i += SCAN_THREADS) {
        pack[2 * T + i] = nact[i];
        pack[3 * T + 8 + i] = nexit[i];
        pack[4 * T + 8 + i] = nexit[T + i];
    }
    if (threadIdx.x < 8) {
        pack[3 * T + threadIdx.x] = thr[threadIdx.x];
        pack[5 * T + 8 + threadIdx.x] = extra ? extra[threadIdx.x] : 0;
    }
    int32_t off = 0;
    for (int32_t t = 0; t < T; t++) {
        const int32_t n = nexit[t], b = node_base[t];
        for (int32_t i = threadIdx.x; i < n; i += SCAN_THREADS) {
            const int32_t k = off + i;
            if (k < max_exits) {
                pack[hdr + 3 * k] = exits[b + i];
                pack[hdr + 3 * k + 1] = exits[N + b + i];
                pack[hdr + 3 * k + 2] = exits[2 * N + b + i];
            }
        }
        off += n;
    }
}

/* ------------------------------------------------------------------ */
/* lextree_enter: all calls of one frame into one tree                 */
/* ------------------------------------------------------------------ */
/*
 * The reference issues the calls sequentially (srch_utt_word_trans: one per
 * word-final CI phone), each walking its left context's root list in order;
 * root nodes are shared between contexts.  Entries e = (call c, list index i)
 * are laid out in call order (ent[e] = {node, call}).  Resolution is per node:
 *   pass 1  key[v]   = max over entries of (score, earliest call)      (64-bit atomicMax)
 *           first[v] = earliest call whose score beats the node's ORIGINAL in-score
 *   pass 2  one workgroup: flag the entry that first enters a node not yet in the
 *           next list, scan, append in entry order (= the reference's append order)
 *   pass 3  the winning entry writes score + history, the flagged one marks frame = nf
 */
__global__ void
k_lt_enter_pass0(const int32_t *__restrict__ ent, int32_t n_ent, unsigned long long *key,
                 int32_t *first)
{
    const int32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_ent) return;
    const int32_t v = ent[2 * e];
    key[v] = 0ull;
    first[v] = INT_MAX;
}

__global__ void
k_lt_enter_pass1(const int32_t *__restrict__ ent, int32_t n_ent, const int32_t *__restrict__ calls,
                 const int32_t *__restrict__ prob, const int32_t *__restrict__ sc, int32_t thresh,
                 unsigned long long *key, int32_t *first)
{
    const int32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_ent) return;
    const int32_t v = ent[2 * e], c = ent[2 * e + 1];
    const int32_t scr = add32(calls[2 * c], prob[v]);
    if (scr < thresh || !(sc[NSV(v)] < scr)) return;
    /* larger score wins; among equal scores the earlier call wins (strict '<' in the reference) */
    atomicMax(&key[v], ((unsigned long long)((uint32_t)scr ^ 0x80000000u) << 32) | (uint32_t)(0x7fffffff - c));
    atomicMin(&first[v], c);
}

__global__ void __launch_bounds__(SCAN_THREADS)
k_lt_enter_pass2(const int32_t *__restrict__ ent, int32_t n_ent, const int32_t *__restrict__ calls,
                 const int32_t *__restrict__ prob, const int32_t *__restrict__ sc,
                 const int32_t *__restrict__ frame, const int32_t *__restrict__ first,
                 int32_t thresh, int32_t nf, int32_t b, int32_t t, int32_t *flag, int32_t *nxt,
                 int32_t *nnxt, int32_t *pos, int32_t *posf)
{
    __shared__ int32_t total;
    for (int32_t e = threadIdx.x; e < n_ent; e += SCAN_THREADS) {
        const int32_t v = ent[2 * e], c = ent[2 * e + 1];
        const int32_t scr = add32(calls[2 * c], prob[v]);
        flag[e] = (scr >= thresh && sc[NSV(v)] < scr && first[v] == c && frame[NSV(v)] != nf) ? 1 : 0;
    }
    __syncthreads();
    block_exclusive_scan(flag, n_ent, &total);
    __syncthreads();
    const int32_t n0 = nnxt[t];
    for (int32_t e = threadIdx.x; e < n_ent; e += SCAN_THREADS) {
        const int32_t v = ent[2 * e], c = ent[2 * e + 1];
        const int32_t scr = add32(calls[2 * c], prob[v]);
        if (scr >= thresh && sc[NSV(v)] < scr && first[v] == c && frame[NSV(v)] != nf) {
            const int32_t k = n0 + flag[e];
            nxt[b + k] = v; PP_SET(pos, v, k, nf);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) nnxt[t] = n0 + total;
}

__global__ void
k_lt_enter_pass3(const int32_t *__restrict__ ent, int32_t n_ent, const int32_t *__restrict__ calls,
                 const int32_t *__restrict__ prob, int32_t thresh, int32_t nf,
                 const unsigned long long *__restrict__ key, const int32_t *__restrict__ first,
                 int32_t *sc, int32_t *hist, int32_t *frame)
{
    const int32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_ent) return;
    const int32_t v = ent[2 * e], c = ent[2 * e + 1];
    const unsigned long long k = key[v];
    if (k == 0ull) return;                              /* nobody entered this node */
    const int32_t win_c = 0x7fffffff - (int32_t)(uint32_t)(k & 0xffffffffu);
    const int32_t win_s = (int32_t)((uint32_t)(k >> 32) ^ 0x80000000u);
    /* every entry of v sees the same key; only the winner's thread writes.  The
     * in-score read in passes 1-2 is the ORIGINAL one, so writing here is safe */
    if (c == win_c) { sc[NSV(v)] = win_s; hist[NSV(v)] = calls[2 * c + 1]; }
    if (c == first[v]) frame[NSV(v)] = nf;
    (void)prob; (void)thresh;
}

/* ------------------------------------------------------------------ */
/* active senones                                                      */
/* ------------------------------------------------------------------ */
__global__ void __launch_bounds__(LT_BLOCK)
k_lt_sen_active(const int32_t *__restrict__ node_base, const int32_t *__restrict__ act,
                const int32_t *__restrict__ nact, const int32_t *__restrict__ ssid,
                const uint8_t *__restrict__ comp, const int16_t *__restrict__ sseq,
                const int16_t *__restrict__ comsseq, const int32_t *__restrict__ comstate_off,
                const int16_t *__restrict__ comstate, uint8_t *sen_active)
{
    const int32_t t = blockIdx.y, i = blockIdx.x * LT_BLOCK + threadIdx.x;
    if (i >= nact[t]) return;
    const int32_t v = act[node_base[t] + i], ss = ssid[v];
    if (comp[v]) {
        for (int st = 0; st < 3; st++) {
            const int32_t cs = comsseq[ss * 3 + st];
            for (int32_t j = comstate_off[cs]; j < comstate_off[cs + 1]; j++)
                sen_active[comstate[j]] = 1;
        }
    }
    else {
        for (int st = 0; st < 3; st++)
            sen_active[sseq[ss * 3 + st]] = 1;
    }
}

/* lextree_utt_end: hmm_clear on everything still active */
__global__ void __launch_bounds__(LT_BLOCK)
k_lt_utt_end(const int32_t *__restrict__ node_base, const int32_t *__restrict__ act,
             const int32_t *__restrict__ nact, int32_t N, int32_t *sc, int32_t *hist,
             int32_t *outs, int32_t *outh, int32_t *bests, int32_t *frame)
{
    const int32_t t = blockIdx.y, i = blockIdx.x * LT_BLOCK + threadIdx.x;
    if (i >= nact[t]) return;
    const int32_t v = act[node_base[t] + i];
    for (int st = 0; st < 3; st++) { sc[NSI(st, N, v)] = WORST; hist[NSI(st, N, v)] = -1; }
    outs[NSV(v)] = WORST; outh[NSV(v)] = -1; bests[NSV(v)] = WORST; frame[NSV(v)] = -1;
}

/* every node record: inactive HMM (hmm_clear) */
__global__ void __launch_bounds__(LT_BLOCK)
k_lt_reset_nodes(int32_t *sc, int32_t N, int32_t ne)
{
    const int32_t v = blockIdx.x * LT_BLOCK + threadIdx.x;
    if (v >= N) return;
    int32_t *r = sc + NSV(v);
    for (int32_t st = 0; st < ne; st++) { r[st] = WORST; r[NS_HIST(ne) + st] = -1; }
    r[NS_OUTS(ne)] = WORST; r[NS_OUTH(ne)] = -1; r[NS_BESTS(ne)] = WORST; r[NS_FRAME(ne)] = -1;
}

__global__ void
k_fill_i32(int32_t *p, int32_t v, int32_t n)
{
    int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

/* ------------------------------------------------------------------ */
/* host side                                                           */
/* ------------------------------------------------------------------ */
#define DMALLOC(ptr, bytes) HIPCHK(hipMalloc((void **)&(ptr), (bytes) ? (bytes) : 4))

static int32_t
fill(s3a_lexsearch_t *ls, int32_t *p, int32_t v, int32_t n)
{
    if (n <= 0) return S3A_OK;
    hipLaunchKernelGGL(k_fill_i32, dim3((n + 255) / 256), dim3(256), 0, ls->stream, p, v, n);
    HIPCHK(hipGetLastError());
    return S3A_OK;
}

/* every second word (one half of the (pos, posf) pairs) */
__global__ void
k_fill2_i32(int32_t *p, int32_t v, int32_t n)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[PPX(i)] = v;
}
static int32_t
fill2(s3a_lexsearch_t *ls, int32_t *p, int32_t v, int32_t n)
{
    if (n <= 0) return S3A_OK;
    hipLaunchKernelGGL(k_fill2_i32, dim3((n + 255) / 256), dim3(256), 0, ls->stream, p, v, n);
    HIPCHK(hipGetLastError());
    return S3A_OK;
}

/* the per-decoder half: HMM state, active lists, per-frame scratch, staging buffers */
static int32_t
alloc_state(s3a_lexsearch_t *ls)
{
    const int32_t N = ls->N, n_tree = ls->n_tree;
    DMALLOC(ls->d_sc, (size_t)NST * (N > 0 ? N : 1) * 4);      /* the node records (s3a_structs.h); the rest point into them */
    const int32_t ne = ls->n_emit;
    ls->d_hist = ls->d_sc + NS_HIST(ne); ls->d_outs = ls->d_sc + NS_OUTS(ne); ls->d_outh = ls->d_sc + NS_OUTH(ne);
    ls->d_bests = ls->d_sc + NS_BESTS(ne); ls->d_frame = ls->d_sc + NS_FRAME(ne);
    DMALLOC(ls->d_pos, (size_t)N * 8); ls->d_posf = ls->d_pos + 1;     /* (pairs: PPX, s3a_structs.h) */
    DMALLOC(ls->d_act[0], (size_t)N * 4); DMALLOC(ls->d_act[1], (size_t)N * 4);
    DMALLOC(ls->d_nact[0], (size_t)n_tree * 4); DMALLOC(ls->d_nact[1], (size_t)n_tree * 4);
    DMALLOC(ls->d_cand, (size_t)N * 4); DMALLOC(ls->d_ncand, (size_t)n_tree * 4);
    DMALLOC(ls->d_candf, (size_t)N * 4);
    DMALLOC(ls->d_turn, (size_t)N * 4); DMALLOC(ls->d_selfemit, (size_t)N * 4); DMALLOC(ls->d_cnt, (size_t)N * 4);
    DMALLOC(ls->d_best, (size_t)n_tree * 2 * 4);
    DMALLOC(ls->d_exit, (size_t)3 * N * 4); DMALLOC(ls->d_nexit, (size_t)2 * n_tree * 4);
    DMALLOC(ls->d_poswid, (size_t)N * 4); DMALLOC(ls->d_posout, (size_t)N * 4);
    {
        int32_t maxn = 1;
        for (int32_t t = 0; t < n_tree; t++) maxn = maxn > ls->node_base[t + 1] - ls->node_base[t] ? maxn : ls->node_base[t + 1] - ls->node_base[t];
        ls->scan_chunks = (maxn + SCAN_THREADS - 1) / SCAN_THREADS;
        ls->scan_epoch = 0;
        DMALLOC(ls->d_scan_agg, (size_t)n_tree * ls->scan_chunks * 8); DMALLOC(ls->d_scan_pre, (size_t)n_tree * ls->scan_chunks * 8);
        DMALLOC(ls->d_scan_flag, (size_t)n_tree * ls->scan_chunks * 4);
        HIPCHK(hipMemset(ls->d_scan_flag, 0, (size_t)n_tree * ls->scan_chunks * 4));
    }
    DMALLOC(ls->d_calls, (size_t)2 * 4096 * 4);
    DMALLOC(ls->d_ent, (size_t)2 * ls->ent_cap * 4); DMALLOC(ls->d_eflag, (size_t)ls->ent_cap * 4);
    DMALLOC(ls->d_first, (size_t)N * 4); DMALLOC(ls->d_key, (size_t)N * 8);
    HIPCHK(hipHostMalloc((void **)&ls->h_pin, (size_t)(8 * n_tree + 16) * 4));
    DMALLOC(ls->d_thr, 8 * 4);
    DMALLOC(ls->d_done, 4 * 4);
    HIPCHK(hipMemset(ls->d_done, 0, 16));
    DMALLOC(ls->d_hbin, 1024 * 4);
    DMALLOC(ls->d_pstamp, (size_t)(ls->n_pset > 0 ? ls->n_pset : 1) * 4);
    DMALLOC(ls->d_ctot, 4096 * 4); DMALLOC(ls->d_n0, (size_t)n_tree * 4);
    HIPCHK(hipMemset(ls->d_hbin, 0, 1024 * 4));
    ls->hist_bound = ls->last_nnxt = ls->row_bound = 1 << 30;
    HIPCHK(hipMemset(ls->d_key, 0, (size_t)N * 8));
    DMALLOC(ls->d_pack, (size_t)(6 * n_tree + 16 + 3 * ls->pack_max_exits) * 4);
    /* (coherent: k_dec_scan writes the frame record here directly; the host reads it after the event behind that kernel) */
    HIPCHK(hipHostMalloc((void **)&ls->h_pack, (size_t)(6 * n_tree + 16 + 3 * ls->pack_max_exits) * 4, hipHostMallocCoherent));
    HIPCHK(hipHostMalloc((void **)&ls->h_ring, (size_t)8 * (2 * 4096 + 2 * ls->ent_cap) * 4));
    HIPCHK(hipEventCreateWithFlags(&ls->ev_pack, hipEventDisableTiming));
    return S3A_OK;
}

static int32_t
lexsearch_build(s3a_lexsearch_t *ls, int32_t n_tree, const int32_t *n_node,
                const int32_t *const *ssid, const int32_t *const *tmatid,
                const uint8_t *const *composite, const int32_t *const *wid,
                const int32_t *const *prob, const int32_t *const *child_off,
                const int32_t *const *child, const int32_t *n_lc, const int16_t *const *lc,
                const int32_t *const *lcroot_off, const int32_t *const *lcroot,
                const int32_t *n_root, const int32_t *const *root, const s3a_tmat_t *tmat,
                const int16_t *sseq, int32_t n_sseq, const int16_t *comsseq, int32_t n_comsseq,
                int32_t n_comstate, const int32_t *comstate_off, const int16_t *comstate)
{
    int32_t N = 0, t;
    ls->n_tree = n_tree;
    ls->node_base.resize(n_tree + 1);
    for (t = 0; t < n_tree; t++) { ls->node_base[t] = N; N += n_node[t]; }
    ls->node_base[n_tree] = N;
    ls->N = N;
    ls->n_tmat = tmat->n_tmat;

    std::vector<int32_t> h_ssid(N), h_tm(N), h_wid(N), h_prob(N), h_coff(N + 1), h_child, h_poff(N + 1, 0), h_par;
    std::vector<uint8_t> h_comp(N);
    std::vector<int32_t> h_roots;
    ls->n_lc.assign(n_lc, n_lc + n_tree);
    ls->lc.resize(n_tree); ls->lcroot_off.resize(n_tree); ls->rootbuf_base.resize(n_tree);
    for (t = 0; t < n_tree; t++) {
        const int32_t b = ls->node_base[t];
        for (int32_t v = 0; v < n_node[t]; v++) {
            h_ssid[b + v] = ssid[t][v]; h_tm[b + v] = tmatid[t][v]; h_wid[b + v] = wid[t][v];
            h_prob[b + v] = prob[t][v]; h_comp[b + v] = composite[t][v];
            if (tmatid[t][v] < 0 || tmatid[t][v] >= tmat->n_tmat
                || ssid[t][v] < 0 || ssid[t][v] >= (composite[t][v] ? n_comsseq : n_sseq)) {
                s3a_set_error("lexsearch: node %d of tree %d has ssid/tmatid out of range", v, t);
                return S3A_EINVAL;
            }
            h_coff[b + v] = (int32_t)h_child.size();
            for (int32_t j = child_off[t][v]; j < child_off[t][v + 1]; j++) {
                int32_t c = child[t][j];
                if (c < 0 || c >= n_node[t]) { s3a_set_error("lexsearch: bad child index"); return S3A_EINVAL; }
                h_child.push_back(b + c);
                h_poff[b + c + 1]++;
            }
        }
        /* root lists: per left context, or the single root list when n_lc == 0 */
        ls->n_root.push_back(n_root[t]);
        ls->rootbuf_base[t] = (int32_t)h_roots.size();
        if (n_lc[t] > 0) {
            ls->lc[t].assign(lc[t], lc[t] + n_lc[t]);
            ls->lcroot_off[t].assign(lcroot_off[t], lcroot_off[t] + n_lc[t] + 1);
            for (int32_t j = 0; j < lcroot_off[t][n_lc[t]]; j++) h_roots.push_back(b + lcroot[t][j]);
        }
        else {
            ls->lcroot_off[t] = { 0, n_root[t] };
            for (int32_t j = 0; j < n_root[t]; j++) h_roots.push_back(b + root[t][j]);
        }
    }
    h_coff[N] = (int32_t)h_child.size();
    for (int32_t v = 0; v < N; v++) h_poff[v + 1] += h_poff[v];
    h_par.resize(h_child.size());
    {
        std::vector<int32_t> fillp(h_poff.begin(), h_poff.end() - 1);
        for (int32_t u = 0; u < N; u++)
            for (int32_t j = h_coff[u]; j < h_coff[u + 1]; j++)
                h_par[fillp[h_child[j]]++] = u;
    }

#define UP(dst, vec) do { DMALLOC(dst, (vec).size() * sizeof((vec)[0])); \
        if ((vec).size()) HIPCHK(hipMemcpy(dst, (vec).data(), (vec).size() * sizeof((vec)[0]), hipMemcpyHostToDevice)); } while (0)
    UP(ls->d_node_base, ls->node_base);
    UP(ls->d_ssid, h_ssid); UP(ls->d_tmatid, h_tm); UP(ls->d_wid, h_wid); UP(ls->d_prob, h_prob);
    UP(ls->d_comp, h_comp); UP(ls->d_child_off, h_coff); UP(ls->d_child, h_child);
    UP(ls->d_par_off, h_poff); UP(ls->d_par, h_par); UP(ls->d_rootlist, h_roots);
    {
        std::vector<int32_t> h_tree_of(N);
        for (t = 0; t < n_tree; t++)
            for (int32_t v = ls->node_base[t]; v < ls->node_base[t + 1]; v++) h_tree_of[v] = t;
        UP(ls->d_tree_of, h_tree_of);
    }
    {
        /* parent SETS: nodes with the same parent list share an id (the ~10 k first-level nodes of a
         * tree all hang under one of a few thousand groups of left-context root variants); an active
         * node stamps the ids its children carry, so "does v have an active parent" is one look-up */
        std::map<std::vector<int32_t>, int32_t> ids;
        std::vector<int32_t> h_ps(N, -1), h_psoff(N + 1, 0), h_psof;
        for (int32_t v = 0; v < N; v++) {
            if (h_poff[v + 1] == h_poff[v]) continue;
            std::vector<int32_t> key(h_par.begin() + h_poff[v], h_par.begin() + h_poff[v + 1]);
            auto it = ids.find(key);
            if (it == ids.end()) it = ids.emplace(std::move(key), (int32_t)ids.size()).first;
            h_ps[v] = it->second;
        }
        for (int32_t u = 0; u < N; u++) {
            std::vector<int32_t> mine;
            for (int32_t j = h_coff[u]; j < h_coff[u + 1]; j++) mine.push_back(h_ps[h_child[j]]);
            std::sort(mine.begin(), mine.end());
            mine.erase(std::unique(mine.begin(), mine.end()), mine.end());
            h_psoff[u] = (int32_t)h_psof.size();
            h_psof.insert(h_psof.end(), mine.begin(), mine.end());
        }
        h_psoff[N] = (int32_t)h_psof.size();
        ls->n_pset = (int32_t)ids.size();
        UP(ls->d_ps, h_ps); UP(ls->d_psof_off, h_psoff); UP(ls->d_psof, h_psof);
        /* ... and the members of every set, ascending (the whole-utterance engine resolves the not active nodes set by set) */
        std::vector<int32_t> h_pmoff(ids.size() + 1, 0), h_pmem(N);
        for (int32_t v = 0; v < N; v++) if (h_ps[v] >= 0) h_pmoff[h_ps[v] + 1]++;
        for (size_t q = 0; q < ids.size(); q++) h_pmoff[q + 1] += h_pmoff[q];
        {
            std::vector<int32_t> fillm(h_pmoff.begin(), h_pmoff.end() - 1);
            for (int32_t v = 0; v < N; v++) if (h_ps[v] >= 0) h_pmem[fillm[h_ps[v]]++] = v;
        }
        h_pmem.resize(h_pmoff[ids.size()] > 0 ? h_pmoff[ids.size()] : 1);
        UP(ls->d_psmem_off, h_pmoff); UP(ls->d_psmem, h_pmem);
    }
    ls->h_rootlist = h_roots;
#undef UP
    {
        /* transition matrices: rows of ne x (ne + 1) words at a 16-byte aligned stride (int4 loads in the HMM kernel) */
        const int32_t ne = ls->n_emit, tpw = NS_TPW(ne), tpr = ne * (ne + 1);
        std::vector<int32_t> h_tp((size_t)tmat->n_tmat * tpw, 0);
        for (int32_t m = 0; m < tmat->n_tmat; m++)
            for (int32_t k = 0; k < tpr; k++) h_tp[(size_t)m * tpw + k] = tmat->tp[(size_t)m * tpr + k];
        DMALLOC(ls->d_tp, h_tp.size() * 4);
        HIPCHK(hipMemcpy(ls->d_tp, h_tp.data(), h_tp.size() * 4, hipMemcpyHostToDevice));
        DMALLOC(ls->d_sseq, (size_t)n_sseq * ne * 2);
        HIPCHK(hipMemcpy(ls->d_sseq, sseq, (size_t)n_sseq * ne * 2, hipMemcpyHostToDevice));
        DMALLOC(ls->d_comsseq, (size_t)n_comsseq * ne * 2);
        if (n_comsseq) HIPCHK(hipMemcpy(ls->d_comsseq, comsseq, (size_t)n_comsseq * ne * 2, hipMemcpyHostToDevice));
        DMALLOC(ls->d_comstate_off, (size_t)(n_comstate + 1) * 4);
        HIPCHK(hipMemcpy(ls->d_comstate_off, comstate_off, (size_t)(n_comstate + 1) * 4, hipMemcpyHostToDevice));
        DMALLOC(ls->d_comstate, (size_t)comstate_off[n_comstate] * 2);
        if (comstate_off[n_comstate])
            HIPCHK(hipMemcpy(ls->d_comstate, comstate, (size_t)comstate_off[n_comstate] * 2, hipMemcpyHostToDevice));
    }
    ls->ent_cap = (int32_t)h_roots.size() > 0 ? (int32_t)h_roots.size() : 1;   /* every list entered once */
    {
        std::vector<int32_t> uniq(h_roots);
        std::sort(uniq.begin(), uniq.end());
        uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
        ls->n_rootnodes = (int32_t)uniq.size();
        DMALLOC(ls->d_rootnodes, uniq.size() * 4);
        if (!uniq.empty()) HIPCHK(hipMemcpy(ls->d_rootnodes, uniq.data(), uniq.size() * 4, hipMemcpyHostToDevice));
    }
    {
        /* every leaf can exit in the same frame (wide beams: thousands do) */
        int32_t n_leaf = 0;
        for (int32_t v = 0; v < N; v++) n_leaf += h_wid[v] >= 0 ? 1 : 0;
        ls->pack_max_exits = n_leaf > 2048 ? n_leaf : 2048;
    }
    return alloc_state(ls);
}

extern "C" s3a_lexsearch_t *
s3a_lexsearch_init(int32_t n_tree, const int32_t *n_node, const int32_t *const *ssid,
                   const int32_t *const *tmatid, const uint8_t *const *composite,
                   const int32_t *const *wid, const int32_t *const *prob,
                   const int32_t *const *child_off, const int32_t *const *child,
                   const int32_t *n_lc, const int16_t *const *lc,
                   const int32_t *const *lcroot_off, const int32_t *const *lcroot,
                   const int32_t *n_root, const int32_t *const *root,
                   const s3a_tmat_t *tmat, const int16_t *sseq, int32_t n_sseq,
                   const int16_t *comsseq, int32_t n_comsseq, int32_t n_comstate,
                   const int32_t *comstate_off, const int16_t *comstate, void *stream)
{
    s3a_lexsearch_t *ls;
    static const int32_t zero_off[1] = { 0 };
    if (n_tree <= 0 || !n_node || !tmat || !sseq || n_sseq <= 0) {
        s3a_set_error("s3a_lexsearch_init: bad arguments");
        return NULL;
    }
    if (tmat->n_state != 3 && tmat->n_state != 5) {
        s3a_set_error("s3a_lexsearch: %d-state HMM topologies are not supported (3 or 5 emitting states: hmm_vit_eval_3st_lr / _5st_lr)", tmat->n_state);
        return NULL;
    }
    if (n_comstate <= 0 || !comstate_off) { n_comstate = 0; comstate_off = zero_off; }
    ls = new s3a_lexsearch_s();        /* value-initialised: every pointer / counter starts at zero */
    ls->n_emit = tmat->n_state;
    ls->opt_calls_by_copy = s3a_variants()->calls_by_copy != 0; ls->opt_scan_chained = s3a_variants()->scan_chained != 0;
    ls->cur = 0;
    if (stream) { ls->stream = (hipStream_t)stream; ls->own_stream = 0; }
    else {
        if (hipStreamCreateWithFlags(&ls->stream, hipStreamNonBlocking) != hipSuccess) {
            s3a_set_error("s3a_lexsearch_init: no HIP device (no CPU fallback)");
            delete ls;
            return NULL;
        }
        ls->own_stream = 1;
    }
    if (lexsearch_build(ls, n_tree, n_node, ssid, tmatid, composite, wid, prob, child_off, child,
                        n_lc, lc, lcroot_off, lcroot, n_root, root, tmat, sseq, n_sseq, comsseq,
                        n_comsseq, n_comstate, comstate_off, comstate) != S3A_OK
        || s3a_lexsearch_reset(ls) != S3A_OK) {
        s3a_lexsearch_free(ls);
        return NULL;
    }
    return ls;
}

/* A second decoder over the SAME lextrees: shares proto's static device arrays (topology, ssid/tmat/
 * word ids, probabilities, parent sets, root lists, senone sequences, transition matrices) and gets
 * its own state.  proto must outlive its clones.  (20 k-word lextrees are ~25 MB of static arrays:
 * B decoders reading B copies of them evict each other from the caches; one copy is read B times.) */
extern "C" s3a_lexsearch_t *
s3a_lexsearch_clone(const s3a_lexsearch_t *proto, void *stream)
{
    if (!proto) return NULL;
    s3a_lexsearch_t *ls = new s3a_lexsearch_s(*proto);         /* host fields + static device pointers */
    ls->is_clone = 1;
    ls->opt_calls_by_copy = s3a_variants()->calls_by_copy != 0; ls->opt_scan_chained = s3a_variants()->scan_chained != 0;
    /* everything alloc_state sets must not alias proto's */
    ls->d_sc = ls->d_hist = ls->d_outs = ls->d_outh = ls->d_bests = ls->d_frame = ls->d_pos = ls->d_posf = NULL;
    ls->d_act[0] = ls->d_act[1] = ls->d_nact[0] = ls->d_nact[1] = ls->d_cand = ls->d_ncand = ls->d_candf = NULL;
    ls->d_turn = ls->d_selfemit = ls->d_cnt = ls->d_best = ls->d_exit = ls->d_nexit = ls->d_calls = NULL;
    ls->d_ent = ls->d_eflag = ls->d_first = ls->d_thr = ls->d_done = ls->d_hbin = ls->d_pstamp = NULL;
    ls->d_ctot = ls->d_n0 = ls->d_pack = NULL; ls->d_key = NULL; ls->d_poswid = ls->d_posout = NULL; ls->d_scan_agg = ls->d_scan_pre = NULL; ls->d_scan_flag = NULL;
    ls->h_pin = ls->h_pack = ls->h_ring = NULL; ls->ev_pack = NULL; ls->ring_slot = 0; ls->cur = 0;
    if (stream) { ls->stream = (hipStream_t)stream; ls->own_stream = 0; }
    else if (hipStreamCreateWithFlags(&ls->stream, hipStreamNonBlocking) != hipSuccess) {
        s3a_set_error("s3a_lexsearch_clone: stream creation failed");
        ls->own_stream = 0;
        s3a_lexsearch_free(ls);
        return NULL;
    }
    else ls->own_stream = 1;
    if (alloc_state(ls) != S3A_OK || s3a_lexsearch_reset(ls) != S3A_OK) {
        s3a_lexsearch_free(ls);
        return NULL;
    }
    return ls;
}

extern "C" void
s3a_lexsearch_free(s3a_lexsearch_t *ls)
{
    if (!ls) return;
    int32_t **statics[] = { &ls->d_node_base, &ls->d_ssid, &ls->d_tmatid, &ls->d_wid, &ls->d_prob,
        &ls->d_child_off, &ls->d_child, &ls->d_par_off, &ls->d_par, &ls->d_rootlist, &ls->d_tp,
        &ls->d_comstate_off, &ls->d_tree_of, &ls->d_rootnodes, &ls->d_ps, &ls->d_psof_off, &ls->d_psof, &ls->d_psmem_off, &ls->d_psmem };
    int32_t **state[] = { &ls->d_sc,            /* (d_hist, d_outs, d_outh, d_bests, d_frame point into d_sc's records) */
        &ls->d_pos, &ls->d_act[0], &ls->d_act[1], &ls->d_nact[0],
        &ls->d_nact[1], &ls->d_cand, &ls->d_ncand, &ls->d_candf, &ls->d_turn, &ls->d_selfemit,
        &ls->d_cnt, &ls->d_best, &ls->d_exit, &ls->d_nexit, &ls->d_calls, &ls->d_ent, &ls->d_eflag,
        &ls->d_first, &ls->d_thr, &ls->d_pack, &ls->d_done, &ls->d_hbin, &ls->d_ctot, &ls->d_n0, &ls->d_pstamp,
        &ls->d_poswid, &ls->d_posout, &ls->d_scan_flag };
    for (auto p : state) (void)hipFree(*p);
    (void)hipFree(ls->d_scan_agg); (void)hipFree(ls->d_scan_pre);
    (void)hipFree(ls->d_key);
    if (!ls->is_clone) {                /* a clone borrows its prototype's static arrays */
        for (auto p : statics) (void)hipFree(*p);
        (void)hipFree(ls->d_comp); (void)hipFree(ls->d_sseq); (void)hipFree(ls->d_comsseq);
        (void)hipFree(ls->d_comstate);
    }
    if (ls->h_pin) (void)hipHostFree(ls->h_pin);
    if (ls->h_pack) (void)hipHostFree(ls->h_pack);
    if (ls->h_ring) (void)hipHostFree(ls->h_ring);
    if (ls->ev_pack) (void)hipEventDestroy(ls->ev_pack);
    if (ls->own_stream && ls->stream) (void)hipStreamDestroy(ls->stream);
    delete ls;
}

/* all HMMs inactive, both lists empty (state after lextree_build / lextree_utt_end) */
extern "C" int32_t
s3a_lexsearch_reset(s3a_lexsearch_t *ls)
{
    int32_t rc, N = ls->N;
    hipLaunchKernelGGL(k_lt_reset_nodes, dim3((unsigned)((N + LT_BLOCK - 1) / LT_BLOCK)), dim3(LT_BLOCK), 0, ls->stream, ls->d_sc, N, ls->n_emit);
    if ( (rc = fill2(ls, ls->d_pos, -1, N)) || (rc = fill2(ls, ls->d_posf, INT_MIN, N))
        || (rc = fill(ls, ls->d_candf, INT_MIN, N)) || (rc = fill(ls, ls->d_pstamp, INT_MIN, ls->n_pset > 0 ? ls->n_pset : 1))
        || (rc = fill(ls, ls->d_turn, -1, N))
        || (rc = fill(ls, ls->d_selfemit, 0, N)) || (rc = fill(ls, ls->d_cnt, 0, N))
        || (rc = fill(ls, ls->d_nact[0], 0, ls->n_tree)) || (rc = fill(ls, ls->d_nact[1], 0, ls->n_tree))
        || (rc = fill(ls, ls->d_nexit, 0, 2 * ls->n_tree))
        || (rc = fill(ls, ls->d_best, INT_MIN, 2 * ls->n_tree)) || (rc = fill(ls, ls->d_first, INT_MAX, N))
        || (rc = fill(ls, ls->d_done, 0, 4)) || (rc = fill(ls, ls->d_hbin, 0, 1024)))
        return rc;
    ls->hist_bound = ls->last_nnxt = ls->row_bound = 1 << 30;
    HIPCHK(hipMemsetAsync(ls->d_key, 0, (size_t)N * 8, ls->stream));
    ls->cur = 0;
    return S3A_OK;
}

extern "C" int32_t
s3a_lexsearch_n_node(const s3a_lexsearch_t *ls, int32_t tree)
{
    return (tree >= 0 && tree < ls->n_tree) ? ls->node_base[tree + 1] - ls->node_base[tree] : S3A_EINVAL;
}

extern "C" int32_t
s3a_lexsearch_hmm_eval(s3a_lexsearch_t *ls, const int32_t *senscr_dev, const int32_t *comsen_dev,
                       int32_t frm, int32_t *best, int32_t *wbest, int32_t *n_active)
{
    LS_NEED_3ST(ls, "s3a_lexsearch_hmm_eval");
    int32_t maxn = 0, t, rc;
    (void)frm;
    if (!ls || !senscr_dev || !best || !wbest || !n_active) return S3A_EINVAL;
    for (t = 0; t < ls->n_tree; t++)
        maxn = max(maxn, ls->node_base[t + 1] - ls->node_base[t]);
    if ((rc = fill(ls, ls->d_best, INT_MIN, 2 * ls->n_tree)) != S3A_OK) return rc;
    hipLaunchKernelGGL(k_lt_hmm_eval, dim3((maxn + LT_BLOCK - 1) / LT_BLOCK, ls->n_tree),
                       dim3(LT_BLOCK), (size_t)ls->n_tmat * 12 * 4, ls->stream, ls->d_node_base,
                       ls->d_act[ls->cur], ls->d_nact[ls->cur], ls->N, ls->n_tmat, ls->d_ssid,
                       ls->d_tmatid, ls->d_wid, ls->d_comp, ls->d_tp, ls->d_sseq, ls->d_comsseq,
                       senscr_dev, comsen_dev ? comsen_dev : senscr_dev, ls->d_sc, ls->d_hist,
                       ls->d_outs, ls->d_outh, ls->d_bests, ls->d_best);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(ls->h_pin, ls->d_best, (size_t)2 * ls->n_tree * 4, hipMemcpyDeviceToHost, ls->stream));
    HIPCHK(hipMemcpyAsync(ls->h_pin + 2 * ls->n_tree, ls->d_nact[ls->cur], (size_t)ls->n_tree * 4,
                          hipMemcpyDeviceToHost, ls->stream));
    HIPCHK(hipStreamSynchronize(ls->stream));
    for (t = 0; t < ls->n_tree; t++) {
        best[t] = ls->h_pin[2 * t];             /* MAX_NEG_INT32 when the tree has no active HMM */
        wbest[t] = ls->h_pin[2 * t + 1];
        n_active[t] = ls->h_pin[2 * ls->n_tree + t];
    }
    return S3A_OK;
}

static int32_t
launch_propagate(s3a_lexsearch_t *ls, int32_t cf)
{
    int32_t maxn = 0, t, rc;
    const int cur = ls->cur, nxt = cur ^ 1;
    for (t = 0; t < ls->n_tree; t++)
        maxn = max(maxn, ls->node_base[t + 1] - ls->node_base[t]);
    if ((rc = fill(ls, ls->d_ncand, 0, ls->n_tree)) != S3A_OK) return rc;
    dim3 grid((maxn + LT_BLOCK - 1) / LT_BLOCK, ls->n_tree);
    hipLaunchKernelGGL(k_lt_prop_mark, grid, dim3(LT_BLOCK), 0, ls->stream, ls->d_node_base,
                       ls->d_act[cur], ls->d_nact[cur], cf, ls->d_thr, ls->d_wid, ls->d_prob,
                       ls->d_child_off, ls->d_child, ls->d_outs, ls->d_posf, ls->d_candf, ls->d_cand,
                       ls->d_ncand);
    hipLaunchKernelGGL(k_lt_prop_resolve, grid, dim3(LT_BLOCK), 0, ls->stream, ls->d_node_base,
                       ls->d_act[cur], ls->d_nact[cur], ls->d_cand, ls->d_ncand, ls->N, cf, ls->d_thr,
                       ls->d_wid, ls->d_prob, ls->d_par_off, ls->d_par, ls->d_pos, ls->d_posf,
                       ls->d_sc, ls->d_hist, ls->d_outs, ls->d_outh, ls->d_bests, ls->d_frame,
                       ls->d_turn, ls->d_selfemit, ls->d_cnt);
    hipLaunchKernelGGL(k_lt_prop_emit, dim3(ls->n_tree), dim3(SCAN_THREADS), 0, ls->stream,
                       ls->d_node_base, ls->d_act[cur], ls->d_nact[cur], cf, ls->d_child_off,
                       ls->d_child, ls->d_turn, ls->d_selfemit, ls->d_cnt, ls->d_act[nxt],
                       ls->d_nact[nxt], ls->d_pos, ls->d_posf);
    HIPCHK(hipGetLastError());
    return S3A_OK;
}

extern "C" int32_t
s3a_lexsearch_propagate_non_leaves(s3a_lexsearch_t *ls, int32_t cf, int32_t th, int32_t pth,
                                   int32_t wth)
{
    LS_NEED_3ST(ls, "s3a_lexsearch_propagate_non_leaves");
    if (!ls) return S3A_EINVAL;
    hipLaunchKernelGGL(k_lt_set_thr, dim3(1), dim3(64), 0, ls->stream, ls->d_thr, th, pth, wth);
    return launch_propagate(ls, cf);
}

extern "C" int32_t
s3a_lexsearch_propagate_leaves(s3a_lexsearch_t *ls, int32_t wth, int32_t *n_exit,
                               int32_t *exit_wid, int32_t *exit_score, int32_t *exit_hist,
                               int32_t max_per_tree)
{
    LS_NEED_3ST(ls, "s3a_lexsearch_propagate_leaves");
    int32_t t;
    const int cur = ls->cur;
    if (!ls || !n_exit || !exit_wid || !exit_score || !exit_hist) return S3A_EINVAL;
    hipLaunchKernelGGL(k_fill_i32, dim3(1), dim3(64), 0, ls->stream, ls->d_thr + 2, wth, 1);
    hipLaunchKernelGGL(k_lt_leaves, dim3(ls->n_tree), dim3(SCAN_THREADS), 0, ls->stream,
                       ls->d_node_base, ls->d_act[cur], ls->d_nact[cur], ls->N, ls->d_thr, ls->d_wid,
                       ls->d_prob, ls->d_outs, ls->d_outh, ls->d_cnt, ls->d_exit, ls->d_nexit,
                       ls->n_tree);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(ls->h_pin, ls->d_nexit, (size_t)2 * ls->n_tree * 4, hipMemcpyDeviceToHost, ls->stream));
    HIPCHK(hipStreamSynchronize(ls->stream));
    for (t = 0; t < ls->n_tree; t++) {
        int32_t n = ls->h_pin[t], b = ls->node_base[t];
        if (ls->h_pin[ls->n_tree + t]) {
            (void)fill(ls, ls->d_nexit, 0, 2 * ls->n_tree);
            s3a_set_error("out.history==-1 at a word exit of tree %d (LEXTREE_OPERATION_FAILURE)", t);
            return S3A_EINVAL;
        }
        n_exit[t] = n;
        if (n > max_per_tree) {
            s3a_set_error("s3a_lexsearch_propagate_leaves: %d word exits exceed the caller's buffer", n);
            return S3A_EINVAL;
        }
        if (n > 0) {
            HIPCHK(hipMemcpyAsync(exit_wid + (size_t)t * max_per_tree, ls->d_exit + b, (size_t)n * 4, hipMemcpyDeviceToHost, ls->stream));
            HIPCHK(hipMemcpyAsync(exit_score + (size_t)t * max_per_tree, ls->d_exit + ls->N + b, (size_t)n * 4, hipMemcpyDeviceToHost, ls->stream));
            HIPCHK(hipMemcpyAsync(exit_hist + (size_t)t * max_per_tree, ls->d_exit + 2 * (size_t)ls->N + b, (size_t)n * 4, hipMemcpyDeviceToHost, ls->stream));
        }
    }
    HIPCHK(hipStreamSynchronize(ls->stream));
    return S3A_OK;
}

extern "C" int32_t
s3a_lexsearch_enter(s3a_lexsearch_t *ls, int32_t tree, int32_t n_calls, const int32_t *lc,
                    const int32_t *inscore, const int32_t *inhist, int32_t cf, int32_t thresh)
{
    LS_NEED_3ST(ls, "s3a_lexsearch_enter");
    if (!ls || tree < 0 || tree >= ls->n_tree || n_calls < 0 || n_calls > 4096) return S3A_EINVAL;
    if (n_calls == 0) return S3A_OK;
    ls->hist_bound = ls->last_nnxt = ls->row_bound = 1 << 30;   /* step-by-step use: the fused frame can no longer bound the list */
    const int nxt = ls->cur ^ 1;
    std::vector<int32_t> calls((size_t)2 * n_calls), ent;
    /* host copy of the tree's root lists is implicit: entries are (node, call) pairs
     * built from the device-resident list through its host-side offsets */
    std::vector<int32_t> &roff = ls->lcroot_off[tree];
    int32_t n_ent = 0;
    std::vector<int32_t> coff((size_t)n_calls), clen((size_t)n_calls);
    for (int32_t c = 0; c < n_calls; c++) {
        int32_t k = 0;
        if (ls->n_lc[tree] > 0) {
            for (k = 0; k < ls->n_lc[tree] && ls->lc[tree][k] != lc[c]; k++);
            if (k >= ls->n_lc[tree]) {
                s3a_set_error("s3a_lexsearch_enter: left context %d is not a root context of tree %d", lc[c], tree);
                return S3A_EINVAL;
            }
        }
        coff[c] = ls->rootbuf_base[tree] + roff[k];
        clen[c] = roff[k + 1] - roff[k];
        calls[2 * c] = inscore[c];
        calls[2 * c + 1] = inhist[c];
        n_ent += clen[c];
    }
    if (n_ent == 0) return S3A_OK;
    if (n_ent > ls->ent_cap) {
        s3a_set_error("s3a_lexsearch_enter: a left context was entered twice in one batch of calls");
        return S3A_EINVAL;
    }
    ent.resize((size_t)2 * n_ent);
    for (int32_t c = 0, e = 0; c < n_calls; c++)
        for (int32_t i = 0; i < clen[c]; i++, e++) {
            ent[2 * e] = ls->h_rootlist[coff[c] + i];
            ent[2 * e + 1] = c;
        }
    /* stage through a pinned ring (8 slots): a slot is reused only after 4 more frames, and
     * every frame ends in a stream synchronisation, so the copies below are truly async */
    {
        const size_t slot_words = (size_t)2 * 4096 + (size_t)2 * ls->ent_cap;
        int32_t *slot = ls->h_ring + (size_t)(ls->ring_slot++ & 7) * slot_words;
        if ((size_t)2 * n_ent > (size_t)2 * ls->ent_cap) { s3a_set_error("enter staging overflow"); return S3A_EINVAL; }
        memcpy(slot, calls.data(), calls.size() * 4);
        memcpy(slot + 2 * 4096, ent.data(), ent.size() * 4);
        HIPCHK(hipMemcpyAsync(ls->d_calls, slot, calls.size() * 4, hipMemcpyHostToDevice, ls->stream));
        HIPCHK(hipMemcpyAsync(ls->d_ent, slot + 2 * 4096, ent.size() * 4, hipMemcpyHostToDevice, ls->stream));
    }
    dim3 g((n_ent + 255) / 256), blk(256);
    hipLaunchKernelGGL(k_lt_enter_pass0, g, blk, 0, ls->stream, ls->d_ent, n_ent, ls->d_key, ls->d_first);
    hipLaunchKernelGGL(k_lt_enter_pass1, g, blk, 0, ls->stream, ls->d_ent, n_ent, ls->d_calls,
                       ls->d_prob, ls->d_sc, thresh, ls->d_key, ls->d_first);
    hipLaunchKernelGGL(k_lt_enter_pass2, dim3(1), dim3(SCAN_THREADS), 0, ls->stream, ls->d_ent,
                       n_ent, ls->d_calls, ls->d_prob, ls->d_sc, ls->d_frame, ls->d_first, thresh,
                       cf + 1, ls->node_base[tree], tree, ls->d_eflag, ls->d_act[nxt],
                       ls->d_nact[nxt], ls->d_pos, ls->d_posf);
    hipLaunchKernelGGL(k_lt_enter_pass3, g, blk, 0, ls->stream, ls->d_ent, n_ent, ls->d_calls,
                       ls->d_prob, thresh, cf + 1, ls->d_key, ls->d_first, ls->d_sc, ls->d_hist,
                       ls->d_frame);
    HIPCHK(hipGetLastError());
    return S3A_OK;
}

/*
 * One whole search frame without intermediate host round trips: lextree_hmm_eval on every
 * tree, the beam thresholds of srch_TST_hmm_compute_lv2, lextree_hmm_propagate_non_leaves,
 * lextree_hmm_propagate_leaves, then ONE packed read-back + synchronisation.
 */
extern "C" int32_t
s3a_lexsearch_frame_search(s3a_lexsearch_t *ls, const int32_t *senscr_dev,
                           const int32_t *comsen_dev, int32_t frm, int32_t hmmbeam, int32_t pbeam,
                           int32_t wbeam, int32_t phone_uses_wbeam, int32_t maxhmmpf,
                           const int32_t *extra_dev, s3a_frame_result_t *res, int32_t *n_exit,
                           int32_t *exit_wid, int32_t *exit_score, int32_t *exit_hist,
                           int32_t max_exits)
{
    LS_NEED_3ST(ls, "s3a_lexsearch_frame_search");
    int32_t maxn = 0, t, rc, total = 0;
    if (!ls || !senscr_dev || !res || !n_exit || !exit_wid || !exit_score || !exit_hist) return S3A_EINVAL;
    const int32_t T = ls->n_tree, hdr = 5 * T + 16;
    for (t = 0; t < T; t++)
        maxn = max(maxn, ls->node_base[t + 1] - ls->node_base[t]);
    if ((rc = fill(ls, ls->d_best, INT_MIN, 2 * T)) != S3A_OK) return rc;
    hipLaunchKernelGGL(k_lt_hmm_eval, dim3((maxn + LT_BLOCK - 1) / LT_BLOCK, T), dim3(LT_BLOCK),
                       (size_t)ls->n_tmat * 12 * 4, ls->stream, ls->d_node_base, ls->d_act[ls->cur],
                       ls->d_nact[ls->cur], ls->N, ls->n_tmat, ls->d_ssid, ls->d_tmatid, ls->d_wid,
                       ls->d_comp, ls->d_tp, ls->d_sseq, ls->d_comsseq, senscr_dev,
                       comsen_dev ? comsen_dev : senscr_dev, ls->d_sc, ls->d_hist, ls->d_outs,
                       ls->d_outh, ls->d_bests, ls->d_best);
    hipLaunchKernelGGL(k_lt_thresholds, dim3(1), dim3(64), 0, ls->stream, ls->d_best,
                       ls->d_nact[ls->cur], T, hmmbeam, pbeam, wbeam, phone_uses_wbeam, maxhmmpf,
                       ls->d_thr);
    if ((rc = launch_propagate(ls, frm)) != S3A_OK) return rc;
    hipLaunchKernelGGL(k_lt_leaves, dim3(T), dim3(SCAN_THREADS), 0, ls->stream, ls->d_node_base,
                       ls->d_act[ls->cur], ls->d_nact[ls->cur], ls->N, ls->d_thr, ls->d_wid, ls->d_prob,
                       ls->d_outs, ls->d_outh, ls->d_cnt, ls->d_exit, ls->d_nexit, T);
    hipLaunchKernelGGL(k_lt_pack, dim3(1), dim3(SCAN_THREADS), 0, ls->stream, ls->d_node_base, T,
                       ls->N, ls->d_best, ls->d_nact[ls->cur], ls->d_thr, ls->d_nexit, ls->d_exit,
                       extra_dev, ls->d_pack, ls->pack_max_exits);
    HIPCHK(hipGetLastError());
    /* header + the first 256 exits in one copy; the (rare) rest in a second one */
    const int32_t first = 256;
    HIPCHK(hipMemcpyAsync(ls->h_pack, ls->d_pack, (size_t)(hdr + 3 * first) * 4, hipMemcpyDeviceToHost, ls->stream));
    HIPCHK(hipStreamSynchronize(ls->stream));
    const int32_t *p = ls->h_pack;
    res->best_hmm = p[3 * T + 3]; res->best_word = p[3 * T + 4]; res->n_hmm = p[3 * T + 5];
    res->thres = p[3 * T + 0]; res->phone_thres = p[3 * T + 1]; res->word_thres = p[3 * T + 2];
    res->need_histprune = p[3 * T + 6];
    for (int i = 0; i < 8; i++) res->extra[i] = p[5 * T + 8 + i];
    for (t = 0; t < T; t++) {
        if (p[4 * T + 8 + t]) {
            (void)fill(ls, ls->d_nexit, 0, 2 * T);
            s3a_set_error("out.history==-1 at a word exit of tree %d (LEXTREE_OPERATION_FAILURE)", t);
            return S3A_EINVAL;
        }
        n_exit[t] = p[3 * T + 8 + t];
        total += n_exit[t];
    }
    res->n_exit_total = total;
    if (total > max_exits || total > ls->pack_max_exits) {
        s3a_set_error("s3a_lexsearch_frame_search: %d word exits in one frame exceed the buffers", total);
        return S3A_EINVAL;
    }
    if (total > first) {
        HIPCHK(hipMemcpyAsync(ls->h_pack + hdr + 3 * first, ls->d_pack + hdr + 3 * first,
                              (size_t)3 * (total - first) * 4, hipMemcpyDeviceToHost, ls->stream));
        HIPCHK(hipStreamSynchronize(ls->stream));
    }
    for (int32_t k = 0; k < total; k++) {
        exit_wid[k] = p[hdr + 3 * k];
        exit_score[k] = p[hdr + 3 * k + 1];
        exit_hist[k] = p[hdr + 3 * k + 2];
    }
    return S3A_OK;
}

extern "C" int32_t
s3a_lexsearch_active_swap(s3a_lexsearch_t *ls)
{
    if (!ls) return S3A_EINVAL;
    ls->cur ^= 1;
    return fill(ls, ls->d_nact[ls->cur ^ 1], 0, ls->n_tree);
}

extern "C" int32_t
s3a_lexsearch_sen_active(s3a_lexsearch_t *ls, uint8_t *sen_active_dev, int32_t n_sen)
{
    LS_NEED_3ST(ls, "s3a_lexsearch_sen_active");
    int32_t maxn = 0, t;
    if (!ls || !sen_active_dev || n_sen <= 0) return S3A_EINVAL;
    for (t = 0; t < ls->n_tree; t++)
        maxn = max(maxn, ls->node_base[t + 1] - ls->node_base[t]);
    HIPCHK(hipMemsetAsync(sen_active_dev, 0, (size_t)n_sen, ls->stream));
    hipLaunchKernelGGL(k_lt_sen_active, dim3((maxn + LT_BLOCK - 1) / LT_BLOCK, ls->n_tree),
                       dim3(LT_BLOCK), 0, ls->stream, ls->d_node_base, ls->d_act[ls->cur],
                       ls->d_nact[ls->cur], ls->d_ssid, ls->d_comp, ls->d_sseq, ls->d_comsseq,
                       ls->d_comstate_off, ls->d_comstate, sen_active_dev);
    HIPCHK(hipGetLastError());
    return S3A_OK;
}

extern "C" int32_t
s3a_lexsearch_utt_end(s3a_lexsearch_t *ls)
{
    int32_t maxn = 0, t, rc;
    if (!ls) return S3A_EINVAL;
    for (t = 0; t < ls->n_tree; t++)
        maxn = max(maxn, ls->node_base[t + 1] - ls->node_base[t]);
    /* both lists: a driver that stops after the last frame's search (the batched engine drops the
     * final, pointless transition) leaves the survivors in the NEXT list; an already swapped list
     * only names nodes that are clear or about to be cleared anyway */
    for (int w = 0; w < 2; w++)
        hipLaunchKernelGGL(k_lt_utt_end, dim3((maxn + LT_BLOCK - 1) / LT_BLOCK, ls->n_tree),
                           dim3(LT_BLOCK), 0, ls->stream, ls->d_node_base, ls->d_act[ls->cur ^ w],
                           ls->d_nact[ls->cur ^ w], ls->N, ls->d_sc, ls->d_hist, ls->d_outs, ls->d_outh,
                           ls->d_bests, ls->d_frame);
    HIPCHK(hipGetLastError());
    /* frame-tagged scratch must not leak into the next utterance (frames restart at 0) */
    if ((rc = fill(ls, ls->d_nact[0], 0, ls->n_tree)) || (rc = fill(ls, ls->d_nact[1], 0, ls->n_tree))
        || (rc = fill2(ls, ls->d_posf, INT_MIN, ls->N)) || (rc = fill(ls, ls->d_candf, INT_MIN, ls->N))
        || (rc = fill(ls, ls->d_pstamp, INT_MIN, ls->n_pset > 0 ? ls->n_pset : 1)))
        return rc;
    HIPCHK(hipStreamSynchronize(ls->stream));
    return S3A_OK;
}

/* read back one tree's active list and the HMM state of its nodes (tests) */
extern "C" int32_t
s3a_lexsearch_get_active(const s3a_lexsearch_t *ls, int32_t tree, int32_t which,
                         int32_t *n_active, int32_t *nodes, int32_t max_nodes)
{
    int32_t n;
    const int buf = which ? (ls->cur ^ 1) : ls->cur;
    if (!ls || tree < 0 || tree >= ls->n_tree || !n_active) return S3A_EINVAL;
    HIPCHK(hipStreamSynchronize(ls->stream));
    HIPCHK(hipMemcpy(&n, ls->d_nact[buf] + tree, 4, hipMemcpyDeviceToHost));
    *n_active = n;
    if (nodes && n > 0) {
        if (n > max_nodes) return S3A_EINVAL;
        std::vector<int32_t> tmp(n);
        HIPCHK(hipMemcpy(tmp.data(), ls->d_act[buf] + ls->node_base[tree], (size_t)n * 4, hipMemcpyDeviceToHost));
        for (int32_t i = 0; i < n; i++) nodes[i] = tmp[i] - ls->node_base[tree];
    }
    return S3A_OK;
}

extern "C" int32_t
s3a_lexsearch_get_hmm(const s3a_lexsearch_t *ls, int32_t tree, int32_t *score, int32_t *hist,
                      int32_t *out_score, int32_t *out_hist, int32_t *bestscore, int32_t *frame)
{
    if (!ls || tree < 0 || tree >= ls->n_tree) return S3A_EINVAL;
    const int32_t b = ls->node_base[tree], n = ls->node_base[tree + 1] - b, N = ls->N;
    HIPCHK(hipStreamSynchronize(ls->stream));
    {   /* the tree's node records, taken apart on the host */
        std::vector<int32_t> rec((size_t)NST * (n > 0 ? n : 1));
        (void)N;
        HIPCHK(hipMemcpy(rec.data(), ls->d_sc + NSV(b), (size_t)NST * n * 4, hipMemcpyDeviceToHost));
        for (int32_t i = 0; i < n; i++) {
            const int32_t *r = rec.data() + (size_t)i * NST;
            const int32_t ne = ls->n_emit;
            for (int st = 0; st < ne; st++) {
                if (score) score[(size_t)st * n + i] = r[st];
                if (hist) hist[(size_t)st * n + i] = r[NS_HIST(ne) + st];
            }
            if (out_score) out_score[i] = r[NS_OUTS(ne)];
            if (out_hist) out_hist[i] = r[NS_OUTH(ne)];
            if (bestscore) bestscore[i] = r[NS_BESTS(ne)];
            if (frame) frame[i] = r[NS_FRAME(ne)];
        }
    }
    return S3A_OK;
}
