/*
 * s3a_wordlevel.h -- the WORD LEVEL of sphinx3's mode-4 search on the device:
 * what closes a search frame once the lextree kernels have compacted the frame's word exits.
 *
 * Replaces, for all word exits of a frame at once,
 *   vithist_rescore         sphinx3/src/libs3decoder/libsearch/vithist.c:492-574   (-> lm_tg_score, liblm/lm.c:1661-1833)
 *   vithist_enter           vithist.c:396-489  (comp_rc == -1: composite triphones, kbcore.c:626)
 *   vithist_prune + _gc     vithist.c:580-718  (heap order: sphinxbase util/heap.c:113-200)
 *   srch_utt_word_trans     libsearch/srch_time_switch_tree.c:1086-1179
 *   vithist_frame_windup    vithist.c:748-763
 *
 * The reference walks the exits one after another and, per exit, the history entries of the frame
 * its predecessor ended in; what each (exit, predecessor) CANDIDATE sees depends on the ones before
 * it in three ways, all reproduced here without walking:
 *   1. a word candidate enters iff score - wbeam >= the best score entered SO FAR  (vithist.c:560)
 *      = an exclusive prefix maximum over the candidates in walk order (a rejected candidate lies
 *      below the running best, so the maximum over ALL earlier candidates is the same number);
 *   2. candidates with the same LM state (lwid[0], lwid[1]) share ONE entry: it sits where the FIRST
 *      entered candidate of the state put it and holds the best one, the earliest on ties
 *      (strict <, vithist.c:449-455) = a hash insert with atomicMin(first) / atomicMax(score, ~seq);
 *   3. pruning pops the frame's entries from a heap, best first: ranks by score; and when two
 *      entries above the threshold TIE the pop order is the heap's (not FIFO, not LIFO): then -- and
 *      only then -- the heap's order among equal values is worked out in parallel (wl_heap_nrl).
 * History entries of finished frames are immutable and all valid (vithist_frame_gc), so the table
 * is structure-of-arrays in HBM and a candidate reads three words of its predecessor.
 *
 * One workgroup of WL_THREADS per decoder lane runs the phases below with workgroup barriers in
 * between (the frame waits for this one workgroup, so the phases are arranged for few DEPENDENT
 * round trips: the frame record, the exits and what the candidates of an exit share sit in LDS, the
 * two LM look-ups of a candidate run side by side 8-ary, the pruning ranks in LDS); a wide-beam
 * frame (10^5..10^6 candidates) runs the candidate phases chip-wide as separate launches instead.
 */
#ifndef S3A_WORDLEVEL_H
#define S3A_WORDLEVEL_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>
#include "s3a_vit.h"

/* 512 threads (round 4; 1024 before): with four engines on the chip a 1024-thread workgroup of 53 KB LDS waits for half a CU to
 * be free at once -- ku_emit_word 262 -> 231 us per launch in the bench, +1.9 % frames/s -- and the whole GPU suite is green at
 * 512 (profiles/r4_wl_threads_experiment.txt).  256 would shorten the launch further (+2.5 %) but the all-against-all ranking
 * below takes a thread per entry, up to WL_RANK_MAX = 384 (static_assert). */
#ifndef WL_THREADS
#define WL_THREADS 512
#endif
#define WL_WAVES (WL_THREADS / 64)
#define WL_MAXCALL 96       /* lextree_enter calls per frame: #CI phones + 1 */
#define WL_MAXT 16          /* lextrees per decoder (2 x -Nlextree) */
#ifndef WL_LDS_EX
#define WL_LDS_EX 1024
#endif
/* WL_LDS_EX:  frames with at most this many word exits keep them (and their candidate offsets) in LDS */
#define WL_RANK_MAX 384     /* entries above the pruning threshold ranked all against all; beyond: selection */
static_assert(WL_THREADS >= WL_RANK_MAX && WL_THREADS >= 256 && WL_THREADS % 64 == 0 && WL_THREADS <= 1024,
              "the word level ranks up to WL_RANK_MAX entries with a thread each and keeps 256-entry tables a thread per entry "
              "(this is what a 256-thread build got wrong on RM1)");
#define WL_BIG_G 256        /* workgroups per lane of the wide-beam launches */

/* error bits (UCtx.err / s3a_utt_result_t.err) */
#define WL_E_OPEN_EXIT 1    /* out.history == -1 at a word exit (LEXTREE_OPERATION_FAILURE / E_FATAL vithist.c:505) */
#define WL_E_EXITS 2        /* more word exits in a frame than the buffers hold */
#define WL_E_CAND 4         /* more (exit, predecessor) candidates than cand_cap */
#define WL_E_TABLE 8        /* history table full */
#define WL_E_LC 16          /* a word-final phone that is no left context of the unigram tree (assert lextree.c:1111) */
#define WL_E_NOLM 32        /* a word exit without an LM word id */
#define WL_E_SCAN 64        /* k_dec_scan's chained scan timed out */
#define WL_E_CALLS 128      /* more lextree_enter calls than WL_MAXCALL / entries than ent_cap */

struct WLm {                /* lm_t flattened (see include/cmusphinx_amd.h: s3a_lm3g_init) */
    int32_t n_ug, n_bg, n_tg;
    const int32_t *ug_prob, *ug_bowt, *ug_firstbg, *bg_wid, *bg_prob, *bg_bowt, *bg_firsttg, *tg_wid, *tg_prob,
        *inclass;
};

struct WDict {              /* per dictionary word */
    int32_t n_word, n_ci;
    const int32_t *lwid, *fillpen, *last_ci;
    const uint8_t *is_filler;
};

struct WPar {
    int32_t wbeam, bghist, maxwpf, maxhist, wordend, n_lextree, epl, T, hmmbeam;
    int32_t tree_type[WL_MAXT];
    const int32_t *lcmap;   /* [T][n_ci + 1][2]: root-list offset / length of (tree, left context); context n_ci = none */
};

/* the pending lextree_enter calls + where the lane is (device resident, one per lane) */
struct UCtx {
    const float *feat;      /* this utterance's features [nfr][D4 * 4] */
    int32_t active, cf, nfr, cur, n_lextrans, err, thresh, n_calls, n_ent, n_groups, scan_epoch, n_tie_frames;
    int32_t max_cand, max_new;
    int32_t f0;             /* the engine's frame counter at this utterance's first frame (lane refill: s3a_uttdec_decode_queue) */
    int32_t utt;            /* the utterance's place in the queue (-1: a plain decode) */
    int32_t hist_wg;        /* workgroups of ku_hist_count that have finished a frame under the histogram beam (the last one sorts) */
    int32_t groups[8];
    int32_t calls[4 * WL_MAXCALL];
    long long tacc[16];     /* time spent per word-level phase (100 MHz ticks; tools/wl_phases) */
    long long kdbg[4];      /* ku_frames, the launch that began at engine frame 512: clock at entry / exit, hardware id, XCC id (diagnostics) */
    long long kacc[16];     /* ku_frames: time per step of the frame (100 MHz ticks, workgroup 0 of the lane's cluster; s3a_uttdec_frame_ticks) */
};

struct WLane {              /* one lane's history table + per-frame scratch (device pointers) */
    int32_t *score, *pred, *lw0, *lw1, *wid, *sf, *ef, *ascr, *lscr, *type;
    int32_t *lmc;           /* [5][cap] LM context of every entry (wl_lm_context) */
    int32_t cap;
    int32_t *frame_start, *bestscore, *bestvh;      /* [max_frames + 2] */
    int32_t *st;            /* [0] n_entry  [1] n_frm */
    int32_t *ex_off;        /* [ex_cap + 1] first candidate of every exit */
    int32_t *ex_info;       /* [3][ex_cap] per exit: first predecessor (-1: filler) | LM word id (filler: its penalty) | score[history] */
    int32_t ex_cap;
    int32_t *cand_score, *cand_slot, *cand_pref, *cand_e, *cand_i;     /* [cand_cap] (cand_i: the predecessor entry) */
    int32_t cand_cap;
    unsigned long long *hkey, *hbest;               /* [hmask + 1] */
    uint32_t *hfirst;
    int32_t *hlead_rank;
    int32_t hmask;
    int32_t *sg;            /* staging [11][new_cap]: wid sf ascr lscr score pred type lw0 lw1 slot | valid */
    int32_t new_cap;
    int32_t *srt;           /* [6][new_cap]: above-threshold list, sorted order, flags, scans */
    int32_t *wfirst;        /* [n_word]: first sorted position of a word (INT_MAX when idle) */
    int32_t *wbest;         /* [n_word]: best score of a word's entries in the frame (INT_MIN when idle) */
    int32_t *part, *part2, *tb;     /* [WL_BIG_G] per-chunk partials of the wide-beam launches; [WL_MAXT + 1] */
    int32_t *heap;          /* [18][new_cap] scratch of wl_heap_nrl */
    int32_t *nrl;           /* [new_cap] pop order of a frame's staged entries among equal scores */
    int32_t *fstat;         /* [max_frames][8] per-frame statistics for the host */
};

/* ------------------------------------------------------------------ */
/* lm_tg_score as a pure function (lm.c:983-995, 1241-1312, 1661-1833) */
/* ------------------------------------------------------------------ */
__device__ __forceinline__ int32_t
wl_find(const int32_t *__restrict__ v, int32_t n, int32_t w)
{
    int32_t lo = 0, hi = n;
    while (lo < hi) {
        const int32_t mid = (lo + hi) >> 1;
        if (v[mid] < w) lo = mid + 1; else hi = mid;
    }
    return (lo < n && v[lo] == w) ? lo : -1;
}

/* Two searches at once, 8-ary: a round is 7 + 7 INDEPENDENT loads, so a look-up in a run of 1000 words is four
 * dependent round trips instead of ten -- what matters when ONE workgroup per lane walks the word level and the
 * frame waits for it (the wide-beam launches keep the bisection: they are bound by traffic, not by the chain).
 * a[0..na) / b[0..nb) ascending without duplicates (the successor lists of lm_3g_dmp.c); na, nb <= 0: empty.
 * ra / rb = the position of wa / wb, or -1. */
__device__ __forceinline__ void
wl_find2(const int32_t *__restrict__ a, int32_t na, int32_t wa, const int32_t *__restrict__ b, int32_t nb, int32_t wb,
         int32_t &ra, int32_t &rb)
{
    int32_t alo = 0, ahi = max(na, 0), blo = 0, bhi = max(nb, 0);     /* the first element >= w lies in [lo, hi] */
    while (ahi - alo > 8 || bhi - blo > 8) {
        const int32_t sa = ahi - alo > 8 ? (ahi - alo) >> 3 : 0, sb = bhi - blo > 8 ? (bhi - blo) >> 3 : 0;
        int32_t xa[7], xb[7];
#pragma unroll
        for (int u = 0; u < 7; u++) { xa[u] = sa ? a[alo + (u + 1) * sa] : INT_MAX; xb[u] = sb ? b[blo + (u + 1) * sb] : INT_MAX; }
        int32_t ca = 0, cb = 0;
#pragma unroll
        for (int u = 0; u < 7; u++) { ca += xa[u] < wa ? 1 : 0; cb += xb[u] < wb ? 1 : 0; }
        if (sa) { const int32_t o = alo; if (ca < 7) ahi = o + (ca + 1) * sa; if (ca > 0) alo = o + ca * sa + 1; }
        if (sb) { const int32_t o = blo; if (cb < 7) bhi = o + (cb + 1) * sb; if (cb > 0) blo = o + cb * sb + 1; }
    }
    int32_t ya[9], yb[9];
#pragma unroll
    for (int u = 0; u < 9; u++) {
        ya[u] = (alo + u <= ahi && alo + u < na) ? a[alo + u] : INT_MAX;
        yb[u] = (blo + u <= bhi && blo + u < nb) ? b[blo + u] : INT_MAX;
    }
    int32_t ca = 0, cb = 0;
    bool fa = false, fb = false;
#pragma unroll
    for (int u = 0; u < 9; u++) {
        ca += ya[u] < wa ? 1 : 0; cb += yb[u] < wb ? 1 : 0;
        fa |= ya[u] == wa && wa != INT_MAX; fb |= yb[u] == wb && wb != INT_MAX;
    }
    ra = fa ? alo + ca : -1;
    rb = fb ? blo + cb : -1;
}

__device__ __forceinline__ int32_t
wl_find8(const int32_t *__restrict__ v, int32_t n, int32_t w)
{
    int32_t r, r2;
    wl_find2(v, n, w, v, 0, 0, r, r2);
    return r;
}

__device__ __forceinline__ int32_t
wl_bg_score(const WLm &lm, int32_t lw1, int32_t lw2, int32_t wid)
{
    int32_t s;
    if (lm.n_bg == 0 || lw1 < 0)
        s = lm.ug_prob[lw2];
    else {
        const int32_t b0 = lm.ug_firstbg[lw1], n = lm.ug_firstbg[lw1 + 1] - b0;
        const int32_t i = n > 0 ? wl_find(lm.bg_wid + b0, n, lw2) : -1;
        s = i >= 0 ? lm.bg_prob[b0 + i] : add32(lm.ug_bowt[lw1], lm.ug_prob[lw2]);
    }
    if (lm.inclass) s = add32(s, lm.inclass[wid]);
    return s;
}

__device__ __forceinline__ int32_t
wl_tg_score(const WLm &lm, int32_t lw1, int32_t lw2, int32_t lw3, int32_t wid)
{
    if (lm.n_tg == 0 || lw1 < 0)
        return wl_bg_score(lm, lw2, lw3, wid);
    const int32_t b0 = lm.ug_firstbg[lw1], nb = lm.ug_firstbg[lw1 + 1] - b0;
    int32_t b = nb > 0 ? wl_find(lm.bg_wid + b0, nb, lw2) : -1;
    int32_t bowt = 0;
    if (b >= 0) {
        b += b0;
        bowt = lm.bg_bowt[b];
        const int32_t t0 = lm.bg_firsttg[b], nt = lm.bg_firsttg[b + 1] - t0;
        const int32_t i = nt > 0 ? wl_find(lm.tg_wid + t0, nt, lw3) : -1;
        if (i >= 0) {
            int32_t s = lm.tg_prob[t0 + i];
            if (lm.inclass) s = add32(s, lm.inclass[wid]);
            return s;
        }
    }
    return add32(bowt, wl_bg_score(lm, lw2, lw3, wid));
}

/* ------------------------------------------------------------------ */
/* workgroup primitives                                                */
/* ------------------------------------------------------------------ */
/* exclusive prefix (sum or max) of src[0..n) into dst[0..n), WL_THREADS wide; returns the total to all */
template <bool MAXOP>
__device__ __forceinline__ int32_t
wl_scan(const int32_t *src, int32_t *dst, int32_t n, int32_t init)
{
    __shared__ int32_t ws[WL_WAVES];
    __shared__ int32_t carry_s;
    const int32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __syncthreads();
    if (tid == 0) carry_s = init;
    __syncthreads();
    for (int32_t base = 0; base < n; base += WL_THREADS) {
        const int32_t i = base + tid;
        const int32_t x = i < n ? src[i] : (MAXOP ? INT_MIN : 0);
        int32_t incl = x;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int32_t y = __shfl_up(incl, o, 64);
            if (lane >= o) incl = MAXOP ? max(incl, y) : incl + y;
        }
        if (lane == 63) ws[wave] = incl;
        __syncthreads();
        int32_t pre = carry_s;          /* everything before this wave */
        for (int32_t w = 0; w < wave; w++) pre = MAXOP ? max(pre, ws[w]) : pre + ws[w];
        int32_t excl = __shfl_up(incl, 1, 64);
        excl = lane == 0 ? pre : (MAXOP ? max(pre, excl) : pre + excl);
        if (i < n) dst[i] = excl;
        __syncthreads();
        if (tid == WL_THREADS - 1) carry_s = MAXOP ? max(excl, x) : excl + x;
        __syncthreads();
    }
    return carry_s;
}

__device__ __forceinline__ unsigned long long
wl_key(int32_t lw0, int32_t lw1)
{
    return (1ull << 62) | ((unsigned long long)(uint32_t)(lw0 + 1) << 31) | (unsigned long long)(uint32_t)(lw1 + 1);
}

__device__ __forceinline__ unsigned long long
wl_pack(int32_t score, uint32_t seq)
{
    return ((unsigned long long)((uint32_t)score ^ 0x80000000u) << 32) | (unsigned long long)(0xffffffffu - seq);
}

#define WL_ALOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

/* the LM context of a history entry with LM state (lw0, lw1), worked out ONCE when the entry is made:
 * the trigram run and back-off weight of the bigram (lw1, lw0) (load_tg, lm.c:1363-1525: a missing bigram
 * or no lw1 = an empty run, weight 0) and the bigram run of lw0 (lm_bg_score's search, lm.c:1262-1280).
 * A candidate then costs two short bisections instead of four. */
__device__ __forceinline__ void
wl_lm_context(const WLm &lm, int32_t lw0, int32_t lw1, int32_t *c5)
{
    int32_t t0 = 0, nt = 0, bowt = 0, b0 = 0, nb = -1;      /* nb < 0: unigram only, no back-off weight */
    if (lm.n_tg > 0 && lw1 >= 0) {
        const int32_t q0 = lm.ug_firstbg[lw1], qn = lm.ug_firstbg[lw1 + 1] - q0;
        const int32_t b = qn > 0 ? wl_find8(lm.bg_wid + q0, qn, lw0) : -1;
        if (b >= 0) { bowt = lm.bg_bowt[q0 + b]; t0 = lm.bg_firsttg[q0 + b]; nt = lm.bg_firsttg[q0 + b + 1] - t0; }
    }
    if (lm.n_bg > 0 && lw0 >= 0) { b0 = lm.ug_firstbg[lw0]; nb = lm.ug_firstbg[lw0 + 1] - b0; }
    c5[0] = t0; c5[1] = nt; c5[2] = bowt; c5[3] = b0; c5[4] = nb;
}

/* lm_tg_score(lw1, lw0, lw3) for the predecessor entry i with precomputed context.  KARY: the trigram look-up and the
 * bigram look-up it falls back to run side by side (wl_find2), everything else is loaded up front. */
template <bool KARY>
__device__ __forceinline__ int32_t
wl_tg_score_ctx(const WLm &lm, const WLane &L, int32_t i, int32_t lw0, int32_t lw3, int32_t wid)
{
    const int32_t t0 = L.lmc[i], nt = L.lmc[L.cap + i];
    int32_t s;
    if (KARY) {
        const int32_t bowt = L.lmc[2 * (size_t)L.cap + i], b0 = L.lmc[3 * (size_t)L.cap + i], nb = L.lmc[4 * (size_t)L.cap + i];
        const int32_t ug = lm.ug_prob[lw3], ubo = lw0 >= 0 ? lm.ug_bowt[lw0] : 0;
        int32_t k, j;
        wl_find2(lm.tg_wid + t0, nt, lw3, lm.bg_wid + b0, nb, lw3, k, j);
        if (k >= 0) s = lm.tg_prob[t0 + k];
        else {
            if (nb < 0) s = ug;
            else s = j >= 0 ? lm.bg_prob[b0 + j] : add32(ubo, ug);
            s = add32(bowt, s);
        }
    }
    else {
        const int32_t k = nt > 0 ? wl_find(lm.tg_wid + t0, nt, lw3) : -1;
        if (k >= 0) s = lm.tg_prob[t0 + k];
        else {
            const int32_t bowt = L.lmc[2 * (size_t)L.cap + i], b0 = L.lmc[3 * (size_t)L.cap + i], nb = L.lmc[4 * (size_t)L.cap + i];
            if (nb < 0) s = lm.ug_prob[lw3];
            else {
                const int32_t j = nb > 0 ? wl_find(lm.bg_wid + b0, nb, lw3) : -1;
                s = j >= 0 ? lm.bg_prob[b0 + j] : add32(lm.ug_bowt[lw0], lm.ug_prob[lw3]);
            }
            s = add32(bowt, s);
        }
    }
    if (lm.inclass) s = add32(s, lm.inclass[wid]);
    return s;
}

/*
 * One frame of the word level for one lane, in phases.  `pack` = the frame record (d_dec_pack_frame):
 * [best,wbest] x T | nact x T | thr[8] | n_exit x T | err x T | misc[8] | n_next x T | exits (wid, score,
 * history) in tree then list order.  Candidates are numbered in the reference's walk order: exit by exit,
 * within a word exit the history entries of the predecessor's frame in table order.
 *
 * The candidate phases take a candidate RANGE [c_lo, c_hi): the usual frame (a few hundred to a few thousand
 * candidates) is ONE workgroup running every phase over [0, n_cand) with workgroup barriers in between
 * (d_wordlevel_frame); a wide-beam frame (10^5 .. 10^6 candidates) is a sequence of launches in which
 * workgroup g of G owns chunk g and the two order-dependent quantities travel as per-chunk partials
 * (L.part): the running best before a candidate = max(best of the chunks in front, prefix inside the chunk),
 * an entry's place in the table = entries founded in the chunks in front + rank inside the chunk.
 * L.st: [0] n_entry [1] n_frm | per frame: [4] nx [5] n_cand [6] the frame's best [7] n_new [8] stop.
 */
struct WlFr {               /* what every phase needs of the frame */
    const int32_t *ex;      /* exits: (wid, score, history) x nx */
    const int32_t *off;     /* [nx + 1] first candidate of every exit */
    const int32_t *tb;      /* [T + 1] first exit of every tree */
    const int32_t *xa, *xb, *xc;    /* per exit (P1): first predecessor entry (-1: a filler word) | LM word id (filler: its
                             * penalty) | path score of the exit's history entry */
    int32_t nx, n_cand, cf;
};

/* P1: the exits' candidate counts -> off[], n_cand; what every candidate of an exit shares -> xa / xb / xc.
 * tb / err_out are shared-memory words of the caller. */
__device__ __forceinline__ int32_t
wl_p1(const WLane &L, const int32_t *pack, const WDict &dict, int32_t T, int32_t *tb, int32_t *off, const int32_t *ex,
      int32_t *xa, int32_t *xb, int32_t *xc, int32_t *err_out)
{
    const int32_t tid = threadIdx.x;
    const int32_t *fstart = L.frame_start;
    if (tid == 0) {
        int32_t n = 0, e = 0;
        for (int32_t t = 0; t < T; t++) {
            tb[t] = n; n += pack[3 * T + 8 + t];
            if (pack[4 * T + 8 + t] == 1) e |= WL_E_OPEN_EXIT;
            if (pack[4 * T + 8 + t] == 2) e |= WL_E_SCAN;
        }
        tb[T] = n;
        if (n > L.ex_cap) e |= WL_E_EXITS;
        *err_out = e;
    }
    __syncthreads();
    const int32_t nx = tb[T];
    if (*err_out) return -1;
    for (int32_t e = tid; e < nx; e += WL_THREADS) {
        const int32_t w = ex[3 * e], h = ex[3 * e + 2];
        const bool filler = dict.is_filler[w] != 0;
        const int32_t aux = filler ? dict.fillpen[w] : dict.lwid[w], sh = L.score[h];
        int32_t c = 1, pb = 0;
        if (!filler && h != 0) { const int32_t f = L.ef[h]; pb = fstart[f]; c = fstart[f + 1] - pb; }
        off[e] = c;
        xa[e] = filler ? -1 : pb; xb[e] = aux; xc[e] = sh;
    }
    const int32_t n_cand = wl_scan<false>(off, off, nx, 0);
    if (tid == 0) off[nx] = n_cand;
    __syncthreads();
    return n_cand;
}

/* P2: the candidates' path scores, their exits and predecessors; the exclusive prefix maximum INSIDE the range ->
 * cand_pref; returns the range's maximum.  *bad (shared) is set when a word exit has no LM word. */
template <bool KARY>
__device__ __forceinline__ int32_t
wl_p2(const WLane &L, const WLm &lm, const WDict &dict, const WlFr &fr, int32_t c_lo, int32_t c_hi, int32_t *bad)
{
    for (int32_t c = c_lo + (int32_t)threadIdx.x; c < c_hi; c += WL_THREADS) {
        int32_t lo = 0, hi = fr.nx;             /* the exit of candidate c: last e with off[e] <= c */
        while (hi - lo > 1) { const int32_t mid = (lo + hi) >> 1; if (fr.off[mid] <= c) lo = mid; else hi = mid; }
        const int32_t e = lo, w = fr.ex[3 * e], scr = fr.ex[3 * e + 1], pb = fr.xa[e], aux = fr.xb[e];
        L.cand_e[c] = e;
        int32_t sc, i;
        if (pb < 0) { sc = add32(scr, aux); i = fr.ex[3 * e + 2]; }         /* a filler word: its penalty, no LM state change */
        else if (aux < 0) { *bad = 1; sc = INT_MIN; i = 0; }
        else {
            i = pb + (c - fr.off[e]);
            sc = add32(add32(L.score[i], add32(scr, -fr.xc[e])), wl_tg_score_ctx<KARY>(lm, L, i, L.lw0[i], aux, w));
        }
        L.cand_score[c] = sc;
        L.cand_i[c] = i;
    }
    __syncthreads();
    return wl_scan<true>(L.cand_score + c_lo, L.cand_pref + c_lo, c_hi - c_lo, INT_MIN);
}

/* P3: which candidates enter (vithist.c:560; `before` = the best score of everything in front of the range), into
 * which LM state */
__device__ __forceinline__ void
wl_p3(const WLane &L, const WDict &dict, const WPar &par, const WlFr &fr, int32_t c_lo, int32_t c_hi, int32_t before)
{
    for (int32_t c = c_lo + (int32_t)threadIdx.x; c < c_hi; c += WL_THREADS) {
        const int32_t e = L.cand_e[c], i = L.cand_i[c], sc = L.cand_score[c];
        const bool filler = fr.xa[e] < 0;
        int32_t slot = -1;
        if (filler || add32(sc, -par.wbeam) >= max(before, L.cand_pref[c])) {
            unsigned long long key;
            if (filler) key = wl_key(L.lw0[i], L.lw1[i]);
            else key = wl_key(fr.xb[e], L.lw0[i]);
            uint32_t hh = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 40) & (uint32_t)L.hmask;
            for (;;) {
                const unsigned long long old = atomicCAS(&L.hkey[hh], 0ull, key);
                if (old == 0ull || old == key) break;
                hh = (hh + 1) & (uint32_t)L.hmask;
            }
            slot = (int32_t)hh;
            atomicMin(&L.hfirst[hh], (uint32_t)c);
            atomicMax(&L.hbest[hh], wl_pack(sc, (uint32_t)c));
        }
        L.cand_slot[c] = slot;
    }
}

/* P4a: the first entered candidate of each LM state founds the entry: its rank INSIDE the range -> cand_pref;
 * returns the number of entries founded in the range */
__device__ __forceinline__ int32_t
wl_p4a(const WLane &L, int32_t c_lo, int32_t c_hi)
{
    for (int32_t c = c_lo + (int32_t)threadIdx.x; c < c_hi; c += WL_THREADS) {
        const int32_t slot = L.cand_slot[c];
        L.cand_pref[c] = (slot >= 0 && WL_ALOAD(&L.hfirst[slot]) == (uint32_t)c) ? 1 : 0;
    }
    return wl_scan<false>(L.cand_pref + c_lo, L.cand_pref + c_lo, c_hi - c_lo, 0);
}

/* P4b: the founders publish the entry's place (base = entries founded in front of the range) */
__device__ __forceinline__ void
wl_p4b(const WLane &L, int32_t c_lo, int32_t c_hi, int32_t base)
{
    for (int32_t c = c_lo + (int32_t)threadIdx.x; c < c_hi; c += WL_THREADS) {
        const int32_t slot = L.cand_slot[c];
        if (slot >= 0 && WL_ALOAD(&L.hfirst[slot]) == (uint32_t)c) L.hlead_rank[slot] = base + L.cand_pref[c];
    }
}

/* P5: the best candidate of each LM state writes the entry (staged in founding order) */
__device__ __forceinline__ void
wl_p5(const WLane &L, const WDict &dict, const WPar &par, const WlFr &fr, int32_t c_lo, int32_t c_hi)
{
    int32_t *sg_wid = L.sg, *sg_sf = L.sg + L.new_cap, *sg_ascr = L.sg + 2 * L.new_cap, *sg_lscr = L.sg + 3 * L.new_cap,
        *sg_score = L.sg + 4 * L.new_cap, *sg_pred = L.sg + 5 * L.new_cap, *sg_type = L.sg + 6 * L.new_cap,
        *sg_lw0 = L.sg + 7 * L.new_cap, *sg_lw1 = L.sg + 8 * L.new_cap, *sg_slot = L.sg + 9 * L.new_cap;
    for (int32_t c = c_lo + (int32_t)threadIdx.x; c < c_hi; c += WL_THREADS) {
        const int32_t slot = L.cand_slot[c];
        if (slot < 0) continue;
        const int32_t sc = L.cand_score[c];
        if (WL_ALOAD(&L.hbest[slot]) != wl_pack(sc, (uint32_t)c)) continue;
        const int32_t e = L.cand_e[c], w = fr.ex[3 * e], scr = fr.ex[3 * e + 1], h = fr.ex[3 * e + 2], i = L.cand_i[c];
        const int32_t k = WL_ALOAD(&L.hlead_rank[slot]), ascr = add32(scr, -fr.xc[e]);
        int32_t ty = 0;
        for (int32_t t = 0; t < par.T; t++) if (e >= fr.tb[t]) ty = par.tree_type[t];
        sg_wid[k] = w; sg_sf[k] = L.ef[h] + 1; sg_ascr[k] = ascr; sg_score[k] = sc; sg_type[k] = ty; sg_slot[k] = slot;
        if (fr.xa[e] < 0) { sg_lscr[k] = fr.xb[e]; sg_pred[k] = h; sg_lw0[k] = L.lw0[h]; sg_lw1[k] = L.lw1[h]; }
        else { sg_lscr[k] = add32(sc, -add32(L.score[i], ascr)); sg_pred[k] = i; sg_lw0[k] = fr.xb[e]; sg_lw1[k] = L.lw0[i]; }
    }
}

/* ordered append to a list whose length lives in shared memory: one atomic per wave.  Every lane of the wave calls. */
__device__ __forceinline__ int32_t
wl_append(int32_t *counter, bool pred)
{
    const unsigned long long m = __ballot(pred);
    if (!m) return -1;
    const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
    int32_t base = 0;
    if (lane == leader) base = atomicAdd(counter, __popcll(m));
    base = __shfl(base, leader, 64);
    return pred ? base + __popcll(m & ((1ull << lane) - 1ull)) : -1;
}

/* the K-th largest (K >= 1) of vals[0..n), and how many are greater / equal: radix select on (max - value), eight
 * bits at a time from the highest bit the values differ in (scores of one frame share their high bits: a fixed
 * 32-bit radix would send every key to one histogram bin for two passes) */
__device__ __forceinline__ void
wl_select(const int32_t *vals, int32_t n, int32_t K, int32_t &V, int32_t &n_gt, int32_t &n_eq)
{
    __shared__ int32_t sh_hist[256];
    __shared__ uint32_t sh_prefix;
    __shared__ int32_t sh_k, sh_cnt[2], sh_mm[2];
    const int32_t tid = threadIdx.x;
    __syncthreads();
    if (tid == 0) { sh_prefix = 0u; sh_k = K; sh_cnt[0] = 0; sh_cnt[1] = 0; sh_mm[0] = INT_MIN; sh_mm[1] = INT_MAX; }
    __syncthreads();
    {
        int32_t mx = INT_MIN, mn = INT_MAX;
        for (int32_t i = tid; i < n; i += WL_THREADS) { const int32_t v = vals[i]; mx = max(mx, v); mn = min(mn, v); }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { mx = max(mx, __shfl_xor(mx, o, 64)); mn = min(mn, __shfl_xor(mn, o, 64)); }
        if ((tid & 63) == 0) { atomicMax(&sh_mm[0], mx); atomicMin(&sh_mm[1], mn); }
    }
    __syncthreads();
    const int32_t vmax = sh_mm[0];
    const uint32_t range = (uint32_t)vmax - (uint32_t)sh_mm[1];
    int bits = 32 - __clz((int)(range | 1u));               /* keys u = vmax - v lie in [0, 2^bits) */
    bits = (bits + 7) & ~7;
    uint32_t mask = 0u;
    /* the K-th SMALLEST key */
    for (int shift = bits - 8; shift >= 0; shift -= 8) {
        if (tid < 256) sh_hist[tid] = 0;
        __syncthreads();
        const uint32_t prefix = sh_prefix;
        for (int32_t i = tid; i < n; i += WL_THREADS) {
            const uint32_t u = (uint32_t)vmax - (uint32_t)vals[i];
            if ((u & mask) == prefix) atomicAdd(&sh_hist[(u >> shift) & 255u], 1);
        }
        __syncthreads();
        if (tid == 0) {
            int32_t acc = 0, k = sh_k, b;
            for (b = 0; b < 255; b++) { if (acc + sh_hist[b] >= k) break; acc += sh_hist[b]; }
            sh_k = k - acc;
            sh_prefix = prefix | ((uint32_t)b << shift);
        }
        mask |= 255u << shift;
        __syncthreads();
    }
    const uint32_t ku = sh_prefix;
    int32_t g = 0, e = 0;
    for (int32_t i = tid; i < n; i += WL_THREADS) {
        const uint32_t u = (uint32_t)vmax - (uint32_t)vals[i];
        g += u < ku ? 1 : 0; e += u == ku ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { g += __shfl_xor(g, o, 64); e += __shfl_xor(e, o, 64); }
    if ((tid & 63) == 0) { atomicAdd(&sh_cnt[0], g); atomicAdd(&sh_cnt[1], e); }
    __syncthreads();
    V = (int32_t)((uint32_t)vmax - ku); n_gt = sh_cnt[0]; n_eq = sh_cnt[1];
    __syncthreads();
}

/*
 * The pop order of the reference's heap among EQUAL values, in parallel.  sphinxbase's heap (util/heap.c) is a
 * binary tree balanced by subtree counts, so its SHAPE depends on the number of inserts only: insert #i walks a
 * fixed root-to-leaf path (arrival j >= 1 at a node goes left when j is odd, right when even, as arrival (j-1)/2
 * there), pushing the larger of (node's value, carried value) down (strict >: on a tie the newcomer moves on).
 * A node therefore keeps the minimum of its arrival stream (the earliest on ties) and sends, at arrival j, the
 * loser of (minimum so far, arrival j) to a child: an exclusive prefix minimum per node -- one segmented scan per
 * tree level builds the heap.  Popping merges the children's pop sequences, the right child's element first on a
 * tie (subheap_pop, heap.c:159-200: `l->val < r->val` picks left): among equal values the pop order is the
 * pre-order (node, right subtree, left subtree) of the BUILT heap.  nrl[k] = that pre-order number of the node
 * that holds staged entry k after all inserts (subtree sizes = arrival counts).  One workgroup.
 * Scratch hs: [18][stride] int32.  The heap's values are -score (vithist.c:669).
 */
__device__ __forceinline__ void
wl_heap_nrl(const int32_t *sg_score, int32_t n, int32_t *hs, int32_t stride, int32_t *nrl)
{
    __shared__ unsigned long long sh_wk[WL_WAVES];
    __shared__ int32_t sh_wf[WL_WAVES];
    __shared__ unsigned long long sh_carry;
    __shared__ int32_t sh_cnt;
    const int32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int32_t *e_val[2] = { hs, hs + 3 * stride }, *e_id[2] = { hs + stride, hs + 4 * stride },
        *e_node[2] = { hs + 2 * stride, hs + 5 * stride };
    int32_t *len = hs + 6 * stride, *off = hs + 10 * stride, *nrlp = hs + 14 * stride;    /* [4 * stride] each, by node */
    const unsigned long long NONE = ~0ull;
    __syncthreads();
    for (int32_t i = tid; i < n; i += WL_THREADS) {
        e_val[0][i] = (int32_t)(0u - (uint32_t)sg_score[i]); e_id[0][i] = i; e_node[0][i] = 1;
    }
    if (tid == 0) { len[1] = n; off[1] = 0; nrlp[1] = 0; }
    __syncthreads();
    int32_t m = n, cur = 0;
    for (int32_t d = 0; m > 0; d++, cur ^= 1) {
        const int32_t P0 = 1 << d, P1 = 2 << d;
        /* the children's arrival counts and pre-order numbers */
        for (int32_t p = P0 + tid; p < P1; p += WL_THREADS) {
            const int32_t Lp = len[p];
            const int32_t ll = Lp > 0 ? Lp / 2 : 0, lr = Lp > 0 ? (Lp - 1) / 2 : 0;
            len[2 * p] = ll; len[2 * p + 1] = lr;
            nrlp[2 * p + 1] = nrlp[p] + 1; nrlp[2 * p] = nrlp[p] + 1 + lr;
        }
        __syncthreads();
        const int32_t m_next = wl_scan<false>(len + P1, off + P1, P1, 0);      /* the next level's stream offsets */
        /* exclusive segmented prefix minimum over this level's streams */
        if (tid == 0) sh_carry = NONE;
        __syncthreads();
        for (int32_t base = 0; base < m; base += WL_THREADS) {
            const int32_t i = base + tid;
            const bool valid = i < m;
            int32_t p = 0, j = 0, v = 0;
            if (valid) { p = e_node[cur][i]; j = i - off[p]; v = e_val[cur][i]; }
            const unsigned long long own = valid ? (((unsigned long long)((uint32_t)v ^ 0x80000000u) << 32) | (uint32_t)i) : NONE;
            unsigned long long k = own;
            int32_t f = (!valid || j == 0) ? 1 : 0;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned long long k2 = __shfl_up(k, o, 64);
                const int32_t f2 = __shfl_up(f, o, 64);
                if (lane >= o) { if (!f) k = k2 < k ? k2 : k; f |= f2; }
            }
            if (lane == 63) { sh_wk[wave] = k; sh_wf[wave] = f; }
            __syncthreads();
            unsigned long long acc = sh_carry;              /* the running minimum entering this wave */
            for (int32_t w = 0; w < wave; w++) acc = sh_wf[w] ? sh_wk[w] : (sh_wk[w] < acc ? sh_wk[w] : acc);
            const unsigned long long incl = f ? k : (acc < k ? acc : k);
            unsigned long long prev = __shfl_up(incl, 1, 64);
            if (lane == 0) prev = acc;
            __syncthreads();
            if (tid == WL_THREADS - 1) sh_carry = incl;
            if (valid) {
                const unsigned long long excl = j == 0 ? NONE : prev;
                const int32_t Lp = len[p];
                if (j == Lp - 1) {                          /* the node's final element */
                    const unsigned long long fin = excl < own ? excl : own;
                    nrl[e_id[cur][(int32_t)(uint32_t)(fin & 0xffffffffull)]] = nrlp[p];
                }
                if (j > 0) {
                    const int32_t ei = (int32_t)(uint32_t)(excl & 0xffffffffull);
                    const int32_t ev = e_val[cur][ei];
                    const bool push_old = ev > v;           /* root->val > val: the old minimum moves down */
                    const int32_t c = (j & 1) ? 2 * p : 2 * p + 1, q = off[c] + (j - 1) / 2;
                    e_val[cur ^ 1][q] = push_old ? ev : v;
                    e_id[cur ^ 1][q] = push_old ? e_id[cur][ei] : e_id[cur][i];
                    e_node[cur ^ 1][q] = c;
                }
            }
            __syncthreads();
        }
        m = m_next;
    }
    __syncthreads();
    (void)sh_cnt;
}

/* phase timing (thread 0; constant 100 MHz clock): ctx->tacc[i] += time since the previous stamp */
#define WL_STAMP(tp, i) do { if (threadIdx.x == 0 && (tp)) { const long long t_ = (long long)wall_clock64(); ctx->tacc[i] += t_ - *(tp); *(tp) = t_; } } while (0)

/* P6 + P7: vithist_prune, vithist_frame_gc, srch_utt_word_trans, vithist_frame_windup; arms the lane's next frame.
 * One workgroup.  M = the frame's best score, n_new = entries staged. */
__device__ __forceinline__ void
wl_finish(const WLane &L, UCtx *ctx, const int32_t *pack, const WLm &lm, const WDict &dict, const WPar &par, int32_t cf,
          int32_t nx, int32_t n_new, int32_t M, long long *tp)
{
    __shared__ int32_t s_i[16];
    __shared__ unsigned long long s_ci[256];
    __shared__ unsigned long long s_u64[2];
    __shared__ int32_t s_rk[12][WL_RANK_MAX];
    __shared__ int32_t s_cnt[2][256];
    const int32_t tid = threadIdx.x, T = par.T;
    const int32_t fs = L.st[0];
    int32_t *sg_wid = L.sg, *sg_sf = L.sg + L.new_cap, *sg_ascr = L.sg + 2 * L.new_cap, *sg_lscr = L.sg + 3 * L.new_cap,
        *sg_score = L.sg + 4 * L.new_cap, *sg_pred = L.sg + 5 * L.new_cap, *sg_type = L.sg + 6 * L.new_cap,
        *sg_lw0 = L.sg + 7 * L.new_cap, *sg_lw1 = L.sg + 8 * L.new_cap, *sg_slot = L.sg + 9 * L.new_cap,
        *sg_valid = L.sg + 10 * L.new_cap;
    const int32_t prune_beam = add32(pack[3 * T + 2], -pack[3 * T + 4]);    /* word_thres - bestwordscore */
    const int32_t th = add32(M, prune_beam);
    int32_t *a_list = L.srt, *a_sorted = L.srt + L.new_cap, *a_c = L.srt + 2 * L.new_cap, *a_scan = L.srt + 3 * L.new_cap,
        *a_first = L.srt + 4 * L.new_cap, *a_val = L.srt + 5 * L.new_cap;
    __syncthreads();
    if (tid == 0) { s_i[1] = 0; s_i[2] = 0; s_i[3] = INT_MAX; s_i[5] = 0; s_i[6] = 0; s_i[7] = INT_MIN; s_i[8] = 0; s_i[9] = 0; s_i[10] = 0; s_i[13] = ctx->n_lextrans; }
    for (int32_t k = tid; k < n_new; k += WL_THREADS) sg_valid[k] = 0;
    __syncthreads();
    for (int32_t k0 = 0; k0 < n_new; k0 += WL_THREADS) {
        const int32_t k = k0 + tid;
        const int32_t sck = k < n_new ? sg_score[k] : INT_MIN, wk = k < n_new ? sg_wid[k] : 0;
        const int32_t at = wl_append(&s_i[1], k < n_new && sck >= th);
        if (at >= 0) {
            a_list[at] = k;
            if (at < WL_RANK_MAX) { s_rk[0][at] = k; s_rk[1][at] = sck; s_rk[2][at] = wk; s_rk[3][at] = dict.is_filler[wk] ? 1 : 0; }
        }
    }
    __syncthreads();
    const int32_t n_th = s_i[1];
    bool done = false;
    int32_t *nrl = L.nrl;
    for (int attempt = 0; attempt < 2 && !done && n_th > WL_RANK_MAX && par.maxhist > 0 && par.maxwpf > 0; attempt++) {
        /* ---- many entries above the threshold: vithist_prune by SELECTION instead of sorting.  The walk of
         * vithist.c:683-713 keeps (a) the best filler entry, (b) the maxwpf distinct words whose best entries score
         * highest, (c) of those words every entry (only the best with -bghist), (d) of all these the maxhist best.
         * Each is a maximum or a K-th largest value -- unless candidates TIE exactly at one of the cuts: then the
         * heap's pop order among the tied decides, and a second attempt breaks every tie by it (wl_heap_nrl). ---- */
        const bool use_nrl = attempt == 1;
        int32_t sens = 0;
        __syncthreads();
        if (tid == 0) { s_i[5] = 0; s_i[6] = 0; s_i[7] = INT_MIN; s_i[8] = 0; s_i[9] = -1; s_i[10] = 0; s_i[11] = INT_MAX; s_i[12] = 0; }
        __syncthreads();
        if (use_nrl) {
            wl_heap_nrl(sg_score, n_new, L.heap, L.new_cap, nrl);
            if (tid == 0) ctx->n_tie_frames++;
            __syncthreads();
        }
        /* (a) the first filler */
        for (int32_t q = tid; q < n_th; q += WL_THREADS) {
            const int32_t k = a_list[q];
            if (dict.is_filler[sg_wid[k]]) atomicMax(&s_i[7], sg_score[k]);
        }
        __syncthreads();
        const int32_t fmax = s_i[7];
        for (int32_t q = tid; q < n_th; q += WL_THREADS) {
            const int32_t k = a_list[q];
            if (dict.is_filler[sg_wid[k]] && sg_score[k] == fmax) { atomicAdd(&s_i[8], 1); if (use_nrl) atomicMin(&s_i[11], nrl[k]); else s_i[9] = k; }
        }
        __syncthreads();
        if (use_nrl && s_i[8] > 0)
            for (int32_t q = tid; q < n_th; q += WL_THREADS) {
                const int32_t k = a_list[q];
                if (dict.is_filler[sg_wid[k]] && sg_score[k] == fmax && nrl[k] == s_i[11]) s_i[9] = k;
            }
        __syncthreads();
        if (!use_nrl && s_i[8] > 1) sens = 1;
        const int32_t ff = s_i[8] >= 1 ? s_i[9] : -1;
        /* (b) the words' best scores (and, second attempt, where their first entries pop among equals) */
        for (int32_t q0 = 0; q0 < n_th; q0 += WL_THREADS) {
            const int32_t q = q0 + tid;
            bool first = false;
            int32_t w = 0;
            if (q < n_th) {
                const int32_t k = a_list[q];
                w = sg_wid[k];
                if (!(dict.is_filler[w] && k != ff)) first = atomicMax(&L.wbest[w], sg_score[k]) == INT_MIN;   /* first touch: a distinct word */
            }
            const int32_t at = wl_append(&s_i[5], first);
            if (at >= 0) a_sorted[at] = w;
        }
        __syncthreads();
        const int32_t n_words = s_i[5];
        if (use_nrl) {
            for (int32_t q = tid; q < n_th; q += WL_THREADS) {
                const int32_t k = a_list[q], w = sg_wid[k];
                if (!(dict.is_filler[w] && k != ff) && sg_score[k] == WL_ALOAD(&L.wbest[w])) atomicMin(&L.wfirst[w], nrl[k]);
            }
            __syncthreads();
        }
        for (int32_t i = tid; i < n_words; i += WL_THREADS) a_val[i] = WL_ALOAD(&L.wbest[a_sorted[i]]);
        __syncthreads();
        int32_t wcut = INT_MIN, wncut = INT_MAX;        /* kept: best score > wcut, or == wcut and first entry's order <= wncut */
        if (n_words > par.maxwpf) {
            int32_t g, e;
            wl_select(a_val, n_words, par.maxwpf, wcut, g, e);
            if (g + e != par.maxwpf) {
                if (!use_nrl) sens = 1;
                else {
                    for (int32_t i0 = 0; i0 < n_words; i0 += WL_THREADS) {
                        const int32_t i = i0 + tid;
                        const bool tied = i < n_words && a_val[i] == wcut;
                        const int32_t at = wl_append(&s_i[12], tied);
                        if (at >= 0) a_c[at] = -WL_ALOAD(&L.wfirst[a_sorted[i]]);
                    }
                    __syncthreads();
                    int32_t v, g2, e2;
                    wl_select(a_c, s_i[12], par.maxwpf - g, v, g2, e2);
                    wncut = -v;
                }
            }
        }
        /* (c) the candidates of the final cut */
        for (int32_t q0 = 0; q0 < n_th; q0 += WL_THREADS) {
            const int32_t q = q0 + tid;
            int32_t c = 0, k = 0;
            if (q < n_th) {
                k = a_list[q];
                const int32_t w = sg_wid[k];
                if (!(dict.is_filler[w] && k != ff)) {
                    const int32_t wb = WL_ALOAD(&L.wbest[w]);
                    const bool kept = wb > wcut || (wb == wcut && (!use_nrl || wncut == INT_MAX || WL_ALOAD(&L.wfirst[w]) <= wncut));
                    if (kept) {
                        if (!par.bghist) c = 1;
                        else if (sg_score[k] == wb) {
                            if (use_nrl) c = nrl[k] == WL_ALOAD(&L.wfirst[w]) ? 1 : 0;
                            else { c = 1; if (atomicAdd(&L.wfirst[w], 1) != INT_MAX) s_i[10] = 1; }     /* two best entries of a word */
                        }
                    }
                }
            }
            const int32_t at = wl_append(&s_i[6], c != 0);
            if (at >= 0) a_first[at] = k;
        }
        __syncthreads();
        if (s_i[10]) sens = 1;
        const int32_t n_c = s_i[6];
        for (int32_t i = tid; i < n_words; i += WL_THREADS) { L.wbest[a_sorted[i]] = INT_MIN; L.wfirst[a_sorted[i]] = INT_MAX; }
        for (int32_t i = tid; i < n_c; i += WL_THREADS) a_val[i] = sg_score[a_first[i]];
        __syncthreads();
        /* (d) the maxhist best of them */
        int32_t hcut = INT_MIN, hncut = INT_MAX;
        if (n_c > par.maxhist) {
            int32_t g, e;
            wl_select(a_val, n_c, par.maxhist, hcut, g, e);
            if (g + e != par.maxhist) {
                if (!use_nrl) sens = 1;
                else {
                    if (tid == 0) s_i[12] = 0;
                    __syncthreads();
                    for (int32_t i0 = 0; i0 < n_c; i0 += WL_THREADS) {
                        const int32_t i = i0 + tid;
                        const bool tied = i < n_c && a_val[i] == hcut;
                        const int32_t at = wl_append(&s_i[12], tied);
                        if (at >= 0) a_c[at] = -nrl[a_first[i]];
                    }
                    __syncthreads();
                    int32_t v, g2, e2;
                    wl_select(a_c, s_i[12], par.maxhist - g, v, g2, e2);
                    hncut = -v;
                }
            }
        }
        if (!sens) {
            for (int32_t i = tid; i < n_c; i += WL_THREADS)
                if (a_val[i] > hcut || (a_val[i] == hcut && (hncut == INT_MAX || nrl[a_first[i]] <= hncut))) sg_valid[a_first[i]] = 1;
            done = true;
        }
        __syncthreads();
    }
    if (!done && n_th > WL_RANK_MAX) done = true;      /* (-maxwpf 0 / -maxhistpf 0: nothing survives) */
    if (!done) {
        /* ---- at most WL_RANK_MAX entries above the threshold: rank them by score, all against all, in LDS (s_rk: the
         * entry, its score, its word, filler?).  Two of them that TIE pop in the heap's order: wl_heap_nrl, ranks again ---- */
        /* (s_rk[0..3] were filled when the list above the threshold was made) */
        for (int pass = 0; pass < 2; pass++) {
            int32_t k = 0, w = 0, fl = 0, r = 0, tie = 0;
            if (tid < n_th) {
                k = s_rk[0][tid]; w = s_rk[2][tid]; fl = s_rk[3][tid];
                const int32_t sc = s_rk[1][tid], nk = pass ? s_rk[4][tid] : k;
                for (int32_t p2 = 0; p2 < n_th; p2++) {
                    const int32_t s2 = s_rk[1][p2], n2 = pass ? s_rk[4][p2] : s_rk[0][p2];
                    r += (s2 > sc || (s2 == sc && n2 < nk)) ? 1 : 0;
                    tie |= (s2 == sc && p2 != tid) ? 1 : 0;
                }
                if (tie) s_i[2] = 1;
            }
            __syncthreads();
            if (pass == 0 && s_i[2] && par.maxhist > 0) {
                wl_heap_nrl(sg_score, n_new, L.heap, L.new_cap, nrl);
                if (tid == 0) ctx->n_tie_frames++;
                __syncthreads();
                if (tid < n_th) s_rk[4][tid] = nrl[k];
                __syncthreads();
                continue;
            }
            if (tid < n_th) { s_rk[5][r] = k; s_rk[6][r] = w; s_rk[7][r] = fl; }       /* the pop order */
            __syncthreads();
            break;
        }
        /* the walk of vithist.c:683-713 over the sorted entries, in closed form */
        if (tid < n_th && s_rk[7][tid]) atomicMin(&s_i[3], tid);
        __syncthreads();
        const int32_t first_filler = s_i[3];
        {
            const int32_t r = tid;
            int32_t f = -1, c = 0, w = 0;
            bool elig = false;
            if (r < n_th) {
                w = s_rk[6][r];
                elig = !(s_rk[7][r] && r > first_filler);
                if (elig) {                         /* where the word was first seen */
                    f = r;
                    for (int32_t p2 = 0; p2 < r; p2++)
                        if (s_rk[6][p2] == w && !(s_rk[7][p2] && p2 > first_filler)) { f = p2; break; }
                }
                s_rk[8][r] = (elig && f == r) ? 1 : 0;
            }
            (void)wl_scan<false>(s_rk[8], s_rk[9], n_th, 0);        /* s_rk[9][r] = index of the word first seen at r */
            if (r < n_th) {
                c = (f >= 0 && s_rk[9][f] < par.maxwpf && (f == r || !par.bghist)) ? 1 : 0;
                s_rk[10][r] = c;
            }
            (void)wl_scan<false>(s_rk[10], s_rk[11], n_th, 0);      /* entries kept before r */
            if (r < n_th && c && s_rk[11][r] < par.maxhist) sg_valid[s_rk[5][r]] = 1;
        }
        __syncthreads();
    }
    WL_STAMP(tp, 6);
    /* vithist_frame_gc: the valid entries, in table order, become the frame's entries */
    const int32_t n_valid = wl_scan<false>(sg_valid, a_scan, n_new, 0);
    if (tid == 0) { s_u64[0] = 0ull; }
    if (tid < 256) s_ci[tid] = 0ull;
    __syncthreads();
    for (int32_t k = tid; k < n_new; k += WL_THREADS) {
        const int32_t slot = sg_slot[k];
        L.hkey[slot] = 0ull; L.hfirst[slot] = 0xffffffffu; L.hbest[slot] = 0ull;   /* the table is clean again */
        if (!sg_valid[k]) continue;
        const int32_t id = fs + a_scan[k];
        L.wid[id] = sg_wid[k]; L.sf[id] = sg_sf[k]; L.ef[id] = cf; L.ascr[id] = sg_ascr[k]; L.lscr[id] = sg_lscr[k];
        L.score[id] = sg_score[k]; L.pred[id] = sg_pred[k]; L.type[id] = sg_type[k]; L.lw0[id] = sg_lw0[k]; L.lw1[id] = sg_lw1[k];
        {
            int32_t c5[5];
            wl_lm_context(lm, sg_lw0[k], sg_lw1[k], c5);
#pragma unroll
            for (int q = 0; q < 5; q++) L.lmc[(size_t)q * L.cap + id] = c5[q];
        }
        const unsigned long long key = wl_pack(sg_score[k], (uint32_t)id);
        atomicMax(&s_u64[0], key);                                      /* best valid entry, the first on ties */
        atomicMax(&s_ci[dict.last_ci[sg_wid[k]]], key);                 /* ... per word-final CI phone */
    }
    __syncthreads();

    WL_STAMP(tp, 7);
    /* ---- P7: srch_utt_word_trans, vithist_frame_windup, the lane's next frame ----
     * One lextree_enter call per word-final CI phone whose best entry is within -wend_beam of the best of them, in
     * phone order; a lane per phone reads its root list (lcmap), the calls' places follow from counting. */
    const int32_t bestvh = s_u64[0] ? (int32_t)(0xffffffffu - (uint32_t)(s_u64[0] & 0xffffffffull)) : -1;
    const int32_t ktree = (s_i[13] % (par.n_lextree * par.epl)) / par.epl;       /* the unigram tree of this transition */
    if (tid == 0) { s_i[14] = INT_MIN; s_i[15] = 0; }
    __syncthreads();
    int32_t p_bs = 0, p_bv = 0, p_m0 = 0, p_m1 = 0;
    bool p_on = false;
    if (bestvh >= 0 && tid < dict.n_ci && s_ci[tid]) {
        p_on = true;
        p_bs = (int32_t)((uint32_t)(s_ci[tid] >> 32) ^ 0x80000000u);
        p_bv = (int32_t)(0xffffffffu - (uint32_t)(s_ci[tid] & 0xffffffffull));
        const int32_t *m = par.lcmap + ((size_t)ktree * (dict.n_ci + 1) + tid) * 2;
        p_m0 = m[0]; p_m1 = m[1];
        atomicMax(&s_i[14], p_bs);
    }
    __syncthreads();
    if (p_on) p_on = par.wordend == 0 || p_bs > add32(par.wordend, s_i[14]);
    if (p_on && p_m1 < 0) { atomicOr(&s_i[15], WL_E_LC); p_on = false; }
    if (tid < 256) { s_cnt[0][tid] = p_on ? 1 : 0; s_cnt[1][tid] = p_on ? p_m1 : 0; }
    __syncthreads();
    if (p_on) {
        int32_t idx = 0, ent = 0;
        for (int32_t q = 0; q < tid; q++) { idx += s_cnt[0][q]; ent += s_cnt[1][q]; }
        if (idx >= WL_MAXCALL - 1) atomicOr(&s_i[15], WL_E_CALLS);
        else { ctx->calls[4 * idx] = p_bs; ctx->calls[4 * idx + 1] = p_bv; ctx->calls[4 * idx + 2] = p_m0; ctx->calls[4 * idx + 3] = ent; }
    }
    __syncthreads();
    if (tid == 0) {
        const int32_t n_entry = fs + n_valid;
        const int32_t bh = pack[3 * T + 3];
        int32_t n_calls = 0, n_ent = 0, n_groups = 0;
        const int32_t e2 = s_i[15];
        L.bestscore[cf] = nx > 0 ? M : INT_MIN;
        L.bestvh[cf] = bestvh;
        if (bestvh >= 0) {
            for (int32_t q = 0; q < dict.n_ci && q < 256; q++) { n_calls += s_cnt[0][q]; n_ent += s_cnt[1][q]; }
            if (n_calls > WL_MAXCALL - 1) n_calls = WL_MAXCALL - 1;     /* (WL_E_CALLS is set: the utterance stops) */
            ctx->n_lextrans = s_i[13] + 1;
            if (n_calls > 0) {
                ctx->groups[0] = ktree; ctx->groups[1] = 0; ctx->groups[2] = n_ent; ctx->groups[3] = 0;
                n_groups = 1;
            }
            {   /* the filler lextree of this transition: the frame's best exit, no left context */
                const int32_t tf = par.n_lextree + ktree;
                const int32_t *m = par.lcmap + ((size_t)tf * (dict.n_ci + 1) + dict.n_ci) * 2;
                ctx->calls[4 * n_calls] = nx > 0 ? M : INT_MIN; ctx->calls[4 * n_calls + 1] = bestvh;
                ctx->calls[4 * n_calls + 2] = m[0]; ctx->calls[4 * n_calls + 3] = n_ent;
                ctx->groups[4 * n_groups] = tf; ctx->groups[4 * n_groups + 1] = n_ent; ctx->groups[4 * n_groups + 3] = n_calls;
                n_ent += m[1]; n_calls++;
                ctx->groups[4 * n_groups + 2] = n_ent;
                n_groups++;
            }
        }
        ctx->n_calls = n_calls; ctx->n_ent = n_ent; ctx->n_groups = n_groups;
        ctx->thresh = add32(bh, par.hmmbeam);           /* bm->bestscore + bm->hmm */
        L.st[0] = n_entry; L.st[1] = cf + 1;
        L.frame_start[cf + 1] = n_entry;
        L.bestscore[cf + 1] = INT_MIN; L.bestvh[cf + 1] = -1;
        /* what the host keeps of the frame: srch->ascale[], stat_t counters */
        int32_t *fsr = L.fstat + (size_t)cf * 8;
        fsr[0] = pack[5 * T + 8 + 6]; fsr[1] = pack[3 * T + 5]; fsr[2] = pack[5 * T + 8 + 1]; fsr[3] = pack[5 * T + 8 + 2];
        fsr[4] = pack[5 * T + 8 + 3]; fsr[5] = pack[5 * T + 8 + 4]; fsr[6] = pack[3 * T + 6]; fsr[7] = nx;
        if (e2) ctx->err |= e2;
        ctx->cf = cf + 1;
        ctx->cur ^= 1;                                  /* lextree_active_swap */
        /* (NOT at the utterance's last frame: the emission workgroups of this very launch test ctx->active, and one that
         * starts after this workgroup has finished -- another queue's kernels on the chip -- would skip its sweep and leave
         * the propagation scratch set for the lane's NEXT utterance; frames >= nfr are skipped by the frame test anyway) */
        if (e2) ctx->active = 0;
    }
    WL_STAMP(tp, 8);
}

/* errors that end the utterance (as the reference's E_FATAL / SRCH_FAILURE do) */
__device__ __forceinline__ void
wl_stop(UCtx *ctx, int32_t err)
{
    if (threadIdx.x == 0) { ctx->err |= err; ctx->active = 0; }
}

__device__ __forceinline__ bool
wl_check_caps(const WLane &L, UCtx *ctx, int32_t n_cand)
{
    const bool over = n_cand > L.cand_cap || 2 * (long long)n_cand > (long long)L.hmask + 1;
    if (threadIdx.x == 0 && n_cand > ctx->max_cand) ctx->max_cand = n_cand;
    if (over) wl_stop(ctx, WL_E_CAND);
    return !over;
}

__device__ __forceinline__ bool
wl_check_new(const WLane &L, UCtx *ctx, int32_t n_new)
{
    const bool over = n_new > L.new_cap || (long long)L.st[0] + n_new > L.cap;
    if (threadIdx.x == 0 && n_new > ctx->max_new) ctx->max_new = n_new;
    if (over) wl_stop(ctx, WL_E_TABLE);
    return !over;
}

/* the whole frame by ONE workgroup.  hdr = the frame record's header (the caller's LDS copy), ex_lds = the exits in LDS
 * (NULL when there are more than WL_LDS_EX: then they are read from the record in global memory, `pack`) */
__device__ __forceinline__ void
d_wordlevel_frame(const WLane &L, UCtx *ctx, const int32_t *pack, const int32_t *hdr, const int32_t *ex_lds, const WLm &lm,
                  const WDict &dict, const WPar &par, const int32_t cf, long long t_in)
{
    __shared__ int32_t s_tb[WL_MAXT + 1];
    __shared__ int32_t s_flag[2];
    __shared__ int32_t s_off[WL_LDS_EX + 1];                        /* the usual frame: candidate offsets in LDS */
    __shared__ int32_t s_xi[3][WL_LDS_EX];
    const int32_t T = par.T;
    const int32_t *ex = pack + (6 * T + 16);
    int32_t *off = L.ex_off, *xa = L.ex_info, *xb = L.ex_info + L.ex_cap, *xc = L.ex_info + 2 * (size_t)L.ex_cap;
    long long tprev = t_in, *tp = &tprev;
    if (threadIdx.x == 0) s_flag[1] = 0;
    if (ex_lds) { ex = ex_lds; off = s_off; xa = s_xi[0]; xb = s_xi[1]; xc = s_xi[2]; }
    WL_STAMP(tp, 0);
    const int32_t n_cand = wl_p1(L, hdr, dict, T, s_tb, off, ex, xa, xb, xc, &s_flag[0]);
    if (n_cand < 0) { wl_stop(ctx, s_flag[0]); return; }
    if (!wl_check_caps(L, ctx, n_cand)) return;
    WlFr fr;
    fr.ex = ex; fr.off = off; fr.tb = s_tb; fr.xa = xa; fr.xb = xb; fr.xc = xc; fr.nx = s_tb[T]; fr.n_cand = n_cand; fr.cf = cf;
    WL_STAMP(tp, 1);
    const int32_t M = wl_p2<true>(L, lm, dict, fr, 0, n_cand, &s_flag[1]);
    if (s_flag[1]) { wl_stop(ctx, WL_E_NOLM); return; }
    WL_STAMP(tp, 2);
    wl_p3(L, dict, par, fr, 0, n_cand, INT_MIN);
    __syncthreads();
    WL_STAMP(tp, 3);
    const int32_t n_new = wl_p4a(L, 0, n_cand);
    if (!wl_check_new(L, ctx, n_new)) return;
    wl_p4b(L, 0, n_cand, 0);
    __syncthreads();
    WL_STAMP(tp, 4);
    wl_p5(L, dict, par, fr, 0, n_cand);
    __syncthreads();
    WL_STAMP(tp, 5);
    wl_finish(L, ctx, hdr, lm, dict, par, cf, fr.nx, n_new, M, tp);
}

/* ---- the same frame as a sequence of launches, G workgroups per lane (wide-beam frames) ---- */
__device__ __forceinline__ void
wl_chunk(int32_t n_cand, int32_t g, int32_t G, int32_t &c_lo, int32_t &c_hi)
{
    const int32_t per = ((n_cand + G - 1) / G + 63) & ~63;
    c_lo = min(n_cand, g * per); c_hi = min(n_cand, c_lo + per);
}

/* launch A (one workgroup per lane; after d_dec_pack_frame): P1 */
__device__ __forceinline__ void
d_wl_big_begin(const WLane &L, UCtx *ctx, const int32_t *pack, const WDict &dict, const WPar &par)
{
    __shared__ int32_t s_tb[WL_MAXT + 1];
    __shared__ int32_t s_flag[2];
    const int32_t T = par.T, hdr = 6 * T + 16;
    if (threadIdx.x == 0) L.st[8] = 1;                  /* stop, until this launch got through */
    const int32_t n_cand = wl_p1(L, pack, dict, T, s_tb, L.ex_off, pack + hdr, L.ex_info, L.ex_info + L.ex_cap,
                                 L.ex_info + 2 * (size_t)L.ex_cap, &s_flag[0]);
    if (n_cand < 0) { wl_stop(ctx, s_flag[0]); return; }
    if (!wl_check_caps(L, ctx, n_cand)) return;
    if (threadIdx.x <= T) L.tb[threadIdx.x] = s_tb[threadIdx.x];
    if (threadIdx.x == 0) { L.st[4] = s_tb[T]; L.st[5] = n_cand; L.st[8] = 0; }
}

#define WL_BIG_FRAME                                                                                       \
    if (L.st[8]) return;                                                                                   \
    WlFr fr;                                                                                               \
    fr.ex = pack + 6 * par.T + 16; fr.off = L.ex_off; fr.tb = L.tb; fr.nx = L.st[4]; fr.n_cand = L.st[5]; fr.cf = cf; \
    fr.xa = L.ex_info; fr.xb = L.ex_info + L.ex_cap; fr.xc = L.ex_info + 2 * (size_t)L.ex_cap;             \
    int32_t c_lo, c_hi;                                                                                    \
    wl_chunk(fr.n_cand, g, G, c_lo, c_hi)

/* launch B: P2 for chunk g; part[g] = the chunk's best score */
__device__ __forceinline__ void
d_wl_big_p2(const WLane &L, UCtx *ctx, const int32_t *pack, const WLm &lm, const WDict &dict, const WPar &par, int32_t cf,
            int32_t g, int32_t G)
{
    __shared__ int32_t s_bad;
    WL_BIG_FRAME;
    if (threadIdx.x == 0) s_bad = 0;
    __syncthreads();
    const int32_t m = wl_p2<false>(L, lm, dict, fr, c_lo, c_hi, &s_bad);
    if (threadIdx.x == 0) { L.part[g] = m; if (s_bad) atomicOr(&ctx->err, WL_E_NOLM); }
}

/* launch C: P3 for chunk g (the best score in front of the chunk = max of the earlier chunks' bests) */
__device__ __forceinline__ void
d_wl_big_p3(const WLane &L, UCtx *ctx, const int32_t *pack, const WDict &dict, const WPar &par, int32_t cf, int32_t g, int32_t G)
{
    __shared__ int32_t s_m[2];
    WL_BIG_FRAME;
    if (ctx->err & WL_E_NOLM) return;
    if (threadIdx.x == 0) { s_m[0] = INT_MIN; s_m[1] = INT_MIN; }
    __syncthreads();
    int32_t before = INT_MIN, all = INT_MIN;
    for (int32_t q = threadIdx.x; q < G; q += WL_THREADS) { const int32_t v = L.part[q]; all = max(all, v); if (q < g) before = max(before, v); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { before = max(before, __shfl_xor(before, o, 64)); all = max(all, __shfl_xor(all, o, 64)); }
    if ((threadIdx.x & 63) == 0) { atomicMax(&s_m[0], before); atomicMax(&s_m[1], all); }
    __syncthreads();
    if (g == 0 && threadIdx.x == 0) L.st[6] = s_m[1];
    wl_p3(L, dict, par, fr, c_lo, c_hi, s_m[0]);
}

/* launch D: P4a for chunk g; part2[g] = entries founded in the chunk */
__device__ __forceinline__ void
d_wl_big_p4a(const WLane &L, UCtx *ctx, const int32_t *pack, const WPar &par, int32_t cf, int32_t g, int32_t G)
{
    WL_BIG_FRAME;
    if (ctx->err & WL_E_NOLM) return;
    const int32_t n = wl_p4a(L, c_lo, c_hi);
    if (threadIdx.x == 0) L.part2[g] = n;
}

/* launch E: P4b for chunk g */
__device__ __forceinline__ void
d_wl_big_p4b(const WLane &L, UCtx *ctx, const int32_t *pack, const WPar &par, int32_t cf, int32_t g, int32_t G)
{
    __shared__ int32_t s_n[2];
    WL_BIG_FRAME;
    if (ctx->err & WL_E_NOLM) return;
    if (threadIdx.x == 0) { s_n[0] = 0; s_n[1] = 0; }
    __syncthreads();
    int32_t before = 0, all = 0;
    for (int32_t q = threadIdx.x; q < G; q += WL_THREADS) { const int32_t v = L.part2[q]; all += v; if (q < g) before += v; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { before += __shfl_xor(before, o, 64); all += __shfl_xor(all, o, 64); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&s_n[0], before); atomicAdd(&s_n[1], all); }
    __syncthreads();
    if (g == 0 && threadIdx.x == 0) L.st[7] = s_n[1];
    if (s_n[1] > L.new_cap || (long long)L.st[0] + s_n[1] > L.cap) return;       /* (the last launch reports it) */
    wl_p4b(L, c_lo, c_hi, s_n[0]);
}

/* launch F: P5 for chunk g */
__device__ __forceinline__ void
d_wl_big_p5(const WLane &L, UCtx *ctx, const int32_t *pack, const WDict &dict, const WPar &par, int32_t cf, int32_t g, int32_t G)
{
    WL_BIG_FRAME;
    if (ctx->err & WL_E_NOLM) return;
    if (L.st[7] > L.new_cap || (long long)L.st[0] + L.st[7] > L.cap) return;
    wl_p5(L, dict, par, fr, c_lo, c_hi);
}

/* launch G (one workgroup per lane): P6 + P7 */
__device__ __forceinline__ void
d_wl_big_finish(const WLane &L, UCtx *ctx, const int32_t *pack, const WLm &lm, const WDict &dict, const WPar &par, int32_t cf)
{
    if (L.st[8]) return;
    if (ctx->err & WL_E_NOLM) { wl_stop(ctx, WL_E_NOLM); return; }
    const int32_t n_new = L.st[7];
    if (!wl_check_new(L, ctx, n_new)) return;
    wl_finish(L, ctx, pack, lm, dict, par, cf, L.st[4], n_new, L.st[6], (long long *)NULL);
}

#endif
